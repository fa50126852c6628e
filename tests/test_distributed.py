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

    def propagate_res_arrays(self):
        oc = np.full(self.n, -4294967299, dtype=np.int64)
        ns = (np.arange(self.n) % 5 + 10).astype(np.uint64)
        return oc, np.zeros(self.n), np.zeros(self.n), ns


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
        ta, st, meta = hens.ensemble_propagate_until_sharded(lambda n: _FakeIntegrator(n), g, 7.0)
        lo, hi = hens.shard_bounds(n_total, rank, world)
        ok = bool(np.array_equal(st.numpy(), g * 2.0 + 7.0)) and ta.n == hi - lo
        ok = ok and meta.shape == (2, n_total) and bool(np.all(meta[0].numpy() == -4294967299))
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
