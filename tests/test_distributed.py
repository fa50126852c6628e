"""Multi-process (world_size 2, gloo, CPU) tests of the ensemble sharding / gather layer
(heyoka_amd/ensemble.py). The integrators themselves need a GPU, so here a stand-in integrator
object exercising the same interface checks partitioning, gathering and ordering."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from heyoka_amd import ensemble as hens


def test_shard_bounds_partition():
    for n in (1, 2, 7, 8, 1000, 1048576 + 3):
        for w in (1, 2, 3, 8):
            spans = [hens.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


class _FakeIntegrator:
    """Stands in for taylor_adaptive_batch on a machine without GPU: "propagates" by a known
    affine map so that the gathered result can be checked exactly."""

    def __init__(self, n):
        self.n = n
        self.state = None

    def propagate_until(self, t, max_steps=0):
        self.state = self.state * 2.0 + t
        # (Double-length time and step-size records which depend on the lane, so that the gather is checked per system.)
        self.dtime = (np.full(self.n, float(t)), self.state[0] * 1e-20)

    def propagate_res_arrays(self):
        oc = np.full(self.n, -4294967299, dtype=np.int64)
        ns = (np.arange(self.n) % 5 + 10).astype(np.uint64)
        return oc, self.state[1] * 1e-3, self.state[2] * 1e-2, ns


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = np.arange(3 * n_total, dtype=np.float64).reshape(3, n_total)
        ta, st, meta, rec = hens.ensemble_propagate_until_sharded(lambda n: _FakeIntegrator(n), g, 7.0)
        lo, hi = hens.shard_bounds(n_total, rank, world)
        fin = g * 2.0 + 7.0
        ok = bool(np.array_equal(st.numpy(), fin)) and ta.n == hi - lo
        ok = ok and meta.shape == (2, n_total) and bool(np.all(meta[0].numpy() == -4294967299))
        # The records of SURVEY 8(e): times (hi, lo), min / max |h| of every system, in the order of the systems.
        ok = ok and rec.shape == (4, n_total) and bool(np.all(rec[0].numpy() == 7.0))
        ok = ok and bool(np.array_equal(rec[1].numpy(), fin[0] * 1e-20)) and bool(np.array_equal(rec[2].numpy(), fin[1] * 1e-3))
        ok = ok and bool(np.array_equal(rec[3].numpy(), fin[2] * 1e-2))
        # Equal-size fast path (all_gather_into_tensor) and ragged path.
        loc = torch.full((2, 4 if n_total % world == 0 else 3 + rank), float(rank))
        gathered = hens.all_gather_states(loc)
        exp = torch.cat([torch.full((2, 4 if n_total % world == 0 else 3 + r), float(r)) for r in range(world)], dim=1)
        ok = ok and bool(torch.equal(gathered, exp))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [10, 11])
def test_sharded_ensemble_gather_world2(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def _gpu_worker(rank, world, port, n_total, t_final, q):
    """One rank of the sharded ensemble with REAL integrators; all the ranks share GPU 0 (the single-GPU stand-in for
    one-process-per-GPU: same code path as bench.py --gpus N --single-device, gloo instead of RCCL)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import heyoka_amd as hy
        from heyoka_amd import configs

        M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
        g = configs.outer_ss_state(n_total, perturb=1e-8, seed=17)
        mk = lambda n: hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), None, n, high_accuracy=True, device=0)
        ta, st, meta, rec = hens.ensemble_propagate_until_sharded(mk, g, t_final)
        lo, hi = hens.shard_bounds(n_total, rank, world)
        assert ta.batch_size == hi - lo and st.shape == (36, n_total) and meta.shape == (2, n_total)
        assert rec.shape == (4, n_total) and bool((rec[0] == t_final).all()) and bool((rec[2] > 0).all()) and bool((rec[3] >= rec[2]).all())
        q.put((rank, st.numpy().copy(), meta.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("n_total", [256, 250])
def test_sharded_ensemble_real_integrators_two_ranks_one_gpu(n_total):
    """ensemble_propagate_until_sharded() with real integrators on 2 ranks (both on GPU 0): every rank receives the
    same gathered state; it is bit-identical to a single-process run over the whole ensemble (systems are independent,
    the reference's test/ensemble_propagate.cpp:419-420 pattern) and agrees with the oracle."""
    import sys

    import heyoka_amd as hy
    from heyoka_amd import configs

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import heyoka_oracle as ho

    t_final = 12.0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, n_total, t_final, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])

    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    g = configs.outer_ss_state(n_total, perturb=1e-8, seed=17)
    ta = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), g, n_total, high_accuracy=True)
    ta.propagate_until(t_final)
    assert np.array_equal(ta.state, res[0][1])
    oc, _, _, ns = ta.propagate_res_arrays()
    assert res[0][2].dtype == np.int64
    assert np.array_equal(res[0][2][0], oc.astype(np.int64)) and np.array_equal(res[0][2][1], ns.astype(np.int64))
    # The same driver in one process with the collective on the device: final state, outcomes and step counters come
    # straight from the integrator's device arrays (the path the RCCL ranks of bench.py / a multi-GPU job take).
    import torch

    mk = lambda n: hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), None, n, high_accuracy=True, device=0)
    _, st_d, meta_d, rec_d = hens.ensemble_propagate_until_sharded(mk, g, t_final, device="cuda:0")
    assert st_d.is_cuda and meta_d.is_cuda and meta_d.dtype == torch.int64
    assert np.array_equal(st_d.cpu().numpy(), res[0][1]) and np.array_equal(meta_d.cpu().numpy(), res[0][2])

    n_o = (n_total // 8) * 8
    ref, *_ = ho.ensemble_propagate_until(ho.nbody(6, masses=M, Gconst=G), g[:, :n_o], n_o, 8, t_final, high_accuracy=True)
    ref = ref.reshape(36, n_o)
    eps = np.finfo(float).eps
    assert np.max(np.abs(res[0][1][:, :n_o] - ref) / np.maximum(1.0, np.abs(ref))) <= 1e5 * eps


@pytest.mark.gpu
def test_bench_py_two_ranks_on_one_gpu_prints_the_contract_line():
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1), with
    the two ranks sharing GPU 0 and gloo standing in for RCCL: one JSON line from rank 0, whole-job aggregate over both
    ranks, weak scaling, every rank's systems counted."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n = 8192
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--systems", str(n), "--single-device", "--backend", "gloo"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"]
    assert d["unit"] == "system-steps/s" and d["value"] > 0 and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["config"]["systems_per_gpu"] == n and "roofline" in d
    # Both shards did their steps: the whole-job aggregate is twice what rank 0 stepped per launch (the shards differ only
    # in their seeds), and the untimed gather - when the backend could do it - returned every rank's systems.
    per_step_total = d["value"] * d["ms_per_step"] * 1e-3
    assert abs(per_step_total / (2 * d["config"]["system_steps_per_launch"]) - 1) < 0.05
    if d["config"]["untimed_final_state_all_gather_error"] is None:
        assert d["config"]["gathered_systems"] == 2 * n
    # Round 6, first-contact fields: what every rank measured on its own clock and the device it bound, the size of the
    # communicator as the backend reports it, the timeout which bounds every collective, and the build id of the library.
    c = d["config"]
    assert c["rccl_ranks_seen"] == 2 and c["collective_backend"] == "gloo" and c["collective_timeout_s"] > 0
    assert len(c["per_rank_value"]) == 2 and all(v > 0 for v in c["per_rank_value"])
    assert len(c["per_rank_kernel_ms"]) == 2 and all(v > 0 for v in c["per_rank_kernel_ms"])
    assert abs(sum(c["per_rank_value"]) / d["value"] - 1) < 0.25  # (the aggregate uses the slowest rank's clock)
    assert [r["ordinal"] for r in c["per_rank_device"]] == [0, 0]  # (--single-device: both ranks on GPU 0)
    assert all(len(r["pci"].split(":")) == 3 for r in c["per_rank_device"])
    import heyoka_amd as hy

    assert c["library_build_id"] == hy.build_id() and len(c["library_build_id"]) == 16


@pytest.mark.gpu
def test_rccl_executes_on_one_rank_sharded_driver_and_bench():
    """The collective of the multi-GPU path with the REAL backend: a one-rank `nccl` process group (RCCL) on this GPU -
    all_gather / all_gather_into_tensor on the integrator's device views through ensemble_propagate_until_sharded(), and
    bench.py launched the way the driver launches it (`torch.distributed.run --nproc-per-node 1 ... --backend nccl`), whose
    exit status now fails on a failed gather."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
import heyoka_amd as hy
from heyoka_amd import configs, ensemble as hens
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
g = configs.outer_ss_state(96, perturb=1e-8, seed=17)
mk = lambda n: hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), None, n, high_accuracy=True, device=0)
ta, st, meta, rec = hens.ensemble_propagate_until_sharded(mk, g, 6.0, device="cuda:0")
assert st.is_cuda and st.shape == (36, 96) and meta.dtype == torch.int64
assert np.array_equal(st.cpu().numpy(), ta.state)
# Unequal shard sizes take the padded all_gather branch: exercise it with a ragged tensor as well.
r = hens.all_gather_states(torch.arange(10, dtype=torch.float64, device="cuda:0").reshape(2, 5))
assert r.shape == (2, 5)
dist.barrier()
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
""" % root
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0 and "RCCL_ONE_RANK_OK" in out.stdout, out.stderr[-3000:]

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--systems", "8192", "--backend", "nccl", "--no-cpu-baseline", "--no-extra-workloads"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["untimed_final_state_all_gather_error"] is None
    assert d["config"]["gathered_systems"] == 8192 and d["config"]["untimed_final_state_all_gather_ms"] is not None
    assert d["config"]["rccl_ranks_seen"] == 1 and d["config"]["collective_backend"] == "nccl"
    assert d["config"]["per_rank_device"][0]["ordinal"] == 0 and len(d["config"]["per_rank_value"]) == 1


@pytest.mark.gpu
def test_native_gather_behind_the_c_abi_with_and_without_rccl(monkeypatch):
    """hy_ensemble_gather_states(): the final states of the copies returned by ensemble_propagate_until_batch() in one
    buffer - by device-to-device copies, and through RCCL (ncclCommInitAll + grouped ncclSend / ncclRecv, forced here on
    the one device of the box: the route the copies take between the 8 GPUs of a node)."""
    import heyoka_amd as hy
    from heyoka_amd import configs

    n, n_iter = 64, 5
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    states = [configs.outer_ss_state(n, perturb=1e-8, seed=100 + i) for i in range(n_iter)]
    ta = hy.taylor_adaptive_batch(hy.model.nbody(6, masses=M, Gconst=G), states[0], n, high_accuracy=True)

    def gen(copy, i):
        copy.state = states[i]

    res = hy.ensemble_propagate_until_batch(ta, 4.0, n_iter, gen, n_devices=-2)
    expect = np.concatenate([r.state for r in res], axis=1)
    for force, want in (("0", False), ("1", True)):
        monkeypatch.setenv("HEYOKA_AMD_GATHER_RCCL", force)
        got, used = hy.ensemble_gather_states(res)
        assert used is want, (force, used)
        assert got.shape == (36, n * n_iter) and np.array_equal(got, expect)
        # Everything SURVEY 8(e) names in one packed block per integrator: states, double-length times, the records of the
        # propagation (src/ensemble_propagate.cpp:193-297 returns whole integrators).
        full = hy.ensemble_gather_results(res)
        assert full["used_rccl"] is want and np.array_equal(full["state"], expect)
        assert np.array_equal(full["time_hi"], np.concatenate([np.asarray(r.dtime[0]) for r in res]))
        assert np.array_equal(full["time_lo"], np.concatenate([np.asarray(r.dtime[1]) for r in res]))
        pr = [p_ for r in res for p_ in r.propagate_res]
        assert np.array_equal(full["outcome"], np.array([int(p_[0]) for p_ in pr], dtype=np.int64))
        assert np.array_equal(full["min_h"], np.array([p_[1] for p_ in pr])) and np.array_equal(full["max_h"], np.array([p_[2] for p_ in pr]))
        assert np.array_equal(full["n_steps"].astype(np.int64), np.array([int(p_[3]) for p_ in pr]))
        assert np.all(full["time_hi"] == 4.0) and np.all(full["n_steps"] > 0)
