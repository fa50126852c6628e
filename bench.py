#!/usr/bin/env python
"""Headline benchmark: ODE systems x steps / sec (fp64) for the outer-Solar-System ensemble.

Metric / workload (BASELINE.json): `model::nbody` outer Solar System 6-body, fp64, tol = eps
(order 20), high_accuracy = true, 1 048 576 perturbed ICs per MI355X
(benchmark/outer_ss_long_term_batch.cpp of the reference with --perturb 1e-12 --seed 42), propagated
with the device-resident `propagate_until()`.

One bench "step" = one propagate_until(t + DT) over the whole ensemble (every system takes its own
adaptive Taylor steps). value = sum over systems of Taylor steps taken (h != 0, like m_ts_count in
src/taylor_adaptive_batch.cpp:1417) / wall time, aggregated over all ranks (weak scaling: every rank
integrates `--systems` independent ICs, different seeds; the only collective on the path is the
final all_gather of the states, the `ensemble_propagate_*` contract).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

# The CPU baseline runs one OpenMP worker per PHYSICAL core (two SMT siblings thrash the L2-resident tape of a batch):
# ask the OpenMP runtime to place the workers on distinct cores before anything loads it.
os.environ.setdefault("OMP_PLACES", "cores")
os.environ.setdefault("OMP_PROC_BIND", "spread")
# (With OMP_PROC_BIND the OpenMP runtime pins the calling thread - this process's main thread, which also drives the GPU -
# to ONE core for the rest of the process, and every thread created afterwards inherits that mask. The CPU baseline
# restores the mask it found, so that the legs which follow it run like the ones before it.)
AFFINITY_AT_START = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: SURVEY.md 8d's rough estimate of F_alg (flop / system-step, order 20), kept for continuity with the round-1
    # figures only: the roofline numerator is derived from the decomposition the integrator actually built
    # (heyoka_amd/roofline.py: products and additions of the reference's formulas, term by term).
    "outer_ss": 5.7e4,
    "two_body": 4.1e3,
    "nbody64": 6.9e6,
    # (Not a BASELINE.json configuration: model::nbody(6) with its DEFAULT masses - the accelerations are sum / sub /
    # negation trees which the planner flattens in the internal program to reach the kernel of the headline system.)
    "nbody6_default_masses": 5.7e4,
    # Round 6: the general-DAG paths. No SURVEY estimate: the derived count (heyoka_amd/roofline.py) is the only figure.
    # - the headline configuration with kw::compact_mode = true (benchmark/outer_ss_long_term_batch.cpp:113 exposes the
    #   flag): the on-chip kernel stays, the arithmetic is checked against the compact-mode oracle in the tests;
    # - the headline DAG FORCED onto the table stepper (kw::emitter = table): the staged variant, tape in LDS;
    # - a mixed model the cluster planners cannot shape (point masses + the oblateness of the first body: the histories of
    #   one cluster of every class exceed the register file): the staged table stepper picked AUTOMATICALLY;
    # - a mixed model on the multi-class wave-cluster stepper (pendulum chain with cubic bonds: two classes of clusters).
    "outer_ss_compact_mode": None,
    "outer_ss_forced_table": None,
    # - the same DAG on the OTHER table stepper: one system per lane, the rule functions interpreted from node tables, the
    #   tape in HBM - the north star's literal formulation and the fallback for decompositions whose tape exceeds the LDS of
    #   a CU (HEYOKA_AMD_TABLE_LDS=0 switches the staged variant off while the integrator is built);
    "outer_ss_forced_table_hbm_tape": None,
    "nbody6_j2_mixed": None,
    "sine_lattice16_mixed": None,
}
# Initial conditions of the workloads which are not perturbed copies of one orbit.
WORKLOAD_ICS = {
    "nbody64": "Plummer spheres of 64 bodies (seeded)",
    "nbody6_default_masses": "Plummer spheres of 6 bodies (seeded)",
    "outer_ss_compact_mode": "perturbed ICs (perturb 1e-12, seed 42+rank), kw::compact_mode = true",
    "outer_ss_forced_table": "perturbed ICs (perturb 1e-12, seed 42+rank), kw::emitter = table (the staged table stepper)",
    "outer_ss_forced_table_hbm_tape": "perturbed ICs (perturb 1e-12, seed 42+rank), kw::emitter = table with the staged variant switched off (one system per lane, tape in HBM)",
    "nbody6_j2_mixed": "perturbed outer-SS ICs, point masses + oblateness of the first body (heyoka_amd/mixed_models.py)",
    "sine_lattice16_mixed": "random ICs of a chain of 16 pendula with cubic bonds (heyoka_amd/mixed_models.py, seeded)",
}
# Default ensemble sizes of the BASELINE.json configurations (systems per GPU).
DEFAULT_SYSTEMS = {"outer_ss": 1048576, "two_body": 4194304, "nbody64": 65536, "nbody6_default_masses": 1048576,
                   "outer_ss_compact_mode": 1048576, "outer_ss_forced_table": 262144, "outer_ss_forced_table_hbm_tape": 262144, "nbody6_j2_mixed": 262144,
                   "sine_lattice16_mixed": 262144}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_PEAK_TFLOPS = 78.6  # vector FP64


def make_integrator(hy, configs, workload, n_systems, seed, device=0):
    if workload == "outer_ss":
        sys_ = hy.model.nbody(6, masses=configs.OUTER_SS_MASSES, Gconst=configs.OUTER_SS_G)
        st = configs.outer_ss_state(n_systems, perturb=1e-12, seed=seed)
        ta = hy.taylor_adaptive_batch(sys_, None, n_systems, high_accuracy=True, device=device)
        # Years per bench step: ~82 Taylor steps per system and call, i.e. ~0.17 s of kernel per step - the timed region
        # of the driver's `--steps 20 --warmup 5` is > 3 s (clock / thermal steady state, visible to the SMI sampler).
        dt = 60.0
    elif workload in ("outer_ss_compact_mode", "outer_ss_forced_table", "outer_ss_forced_table_hbm_tape"):
        sys_ = hy.model.nbody(6, masses=configs.OUTER_SS_MASSES, Gconst=configs.OUTER_SS_G)
        st = configs.outer_ss_state(n_systems, perturb=1e-12, seed=seed)
        if workload == "outer_ss_compact_mode":
            ta = hy.taylor_adaptive_batch(sys_, None, n_systems, high_accuracy=True, compact_mode=True, device=device)
            dt = 60.0
        elif workload == "outer_ss_forced_table_hbm_tape":
            old = os.environ.get("HEYOKA_AMD_TABLE_LDS")
            os.environ["HEYOKA_AMD_TABLE_LDS"] = "0"
            try:
                ta = hy.taylor_adaptive_batch(sys_, None, n_systems, high_accuracy=True, emitter="table", device=device)
            finally:
                if old is None:
                    del os.environ["HEYOKA_AMD_TABLE_LDS"]
                else:
                    os.environ["HEYOKA_AMD_TABLE_LDS"] = old
            dt = 10.0
        else:
            ta = hy.taylor_adaptive_batch(sys_, None, n_systems, high_accuracy=True, emitter="table", device=device)
            dt = 10.0
    elif workload == "nbody6_j2_mixed":
        from heyoka_amd import mixed_models as mm

        sys_ = mm.nbody_j2(hy, 6, configs.OUTER_SS_MASSES, configs.OUTER_SS_G, 1e-7)
        st = configs.outer_ss_state(n_systems, perturb=1e-12, seed=seed)
        ta = hy.taylor_adaptive_batch(sys_, None, n_systems, device=device)
        dt = 10.0
    elif workload == "sine_lattice16_mixed":
        from heyoka_amd import mixed_models as mm

        sys_ = mm.sine_lattice(hy, 16)
        st = mm.sine_lattice_state(16, n_systems, seed=seed)
        ta = hy.taylor_adaptive_batch(sys_, None, n_systems, device=device)
        dt = 4.0
    elif workload == "nbody64":
        sys_ = hy.model.nbody(64)
        st = configs.plummer_nbody_state(64, n_systems, seed=1234 + seed)
        ta = hy.taylor_adaptive_batch(sys_, None, n_systems, high_accuracy=False, device=device)
        dt = 0.03
    elif workload == "nbody6_default_masses":
        sys_ = hy.model.nbody(6)
        st = configs.plummer_nbody_state(6, n_systems, seed=1234 + seed)
        ta = hy.taylor_adaptive_batch(sys_, None, n_systems, high_accuracy=False, device=device)
        dt = 0.5
    else:
        sys_ = hy.model.nbody(2, masses=[1.0, 0.0])
        st = configs.two_body_state(n_systems, perturb=1e-12, seed=seed)
        ta = hy.taylor_adaptive_batch(sys_, None, n_systems, high_accuracy=False, device=device)
        dt = 50.0
    return ta, st, dt


def physical_cores(hw_threads):
    """Number of physical cores of this host (distinct (package, core) pairs of /proc/cpuinfo), capped by the OpenMP
    thread limit; falls back to the hardware threads."""
    try:
        cores = set()
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if core is not None:
                        cores.add((phys, core))
                    phys = core = None
        if core is not None:
            cores.add((phys, core))
        n = len(cores)
        # (Containers: the affinity mask may be narrower than the machine.)
        n = min(n, len(os.sched_getaffinity(0))) if n else 0
        return max(1, min(n, hw_threads)) if n else hw_threads
    except Exception:
        return hw_threads


def cgroup_cpu_quota():
    """CPU quota of this container in CPUs (cgroup v2 cpu.max / v1 cfs quota), or None if unlimited. More runnable
    threads than the quota are throttled by the kernel: the GPU boxes of this pool expose 256 hardware threads of two
    EPYC 9575F to a container whose quota is 16 CPUs - 128 workers there measured 3.6e4 system-steps/s each, i.e. the
    2.9e5 per CPU of the quota which 16 workers reach directly."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = float(f.read())
        return None if q <= 0 else q / p
    except Exception:
        return None


# (Taken at import: once the OpenMP runtime has bound the primary thread to its place, the affinity mask of this process
# no longer shows the machine.)
HW_THREADS_AT_START = len(os.sched_getaffinity(0))
CPU_QUOTA = cgroup_cpu_quota()
PHYSICAL_CORES_AT_START = physical_cores(HW_THREADS_AT_START)
if CPU_QUOTA is not None:
    # One worker per CPU the container may actually use.
    PHYSICAL_CORES_AT_START = max(1, min(PHYSICAL_CORES_AT_START, int(CPU_QUOTA + 0.5)))


def cpu_baseline(workload, dt, target_seconds=15.0):
    """The oracle (C restatement, OpenMP over SIMD-width batches like the reference's
    TBB-over-batches ensemble) timed on this host on a bounded sample of the same workload."""
    if workload not in ("outer_ss", "two_body", "nbody64", "outer_ss_compact_mode", "outer_ss_forced_table"):
        return None  # (auxiliary workloads carry no CPU baseline)
    try:
        return _cpu_baseline(workload, dt, target_seconds)
    finally:
        if AFFINITY_AT_START is not None:
            os.sched_setaffinity(0, AFFINITY_AT_START)


def _cpu_baseline(workload, dt, target_seconds):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import heyoka_oracle as ho
    from heyoka_amd import configs

    width = 8
    hw_threads = max(ho.max_threads(), HW_THREADS_AT_START)
    threads = min(PHYSICAL_CORES_AT_START, hw_threads)
    compact = workload in ("outer_ss_compact_mode", "outer_ss_forced_table")
    if workload == "outer_ss" or compact:
        osys = ho.nbody(6, masses=configs.OUTER_SS_MASSES, Gconst=configs.OUTER_SS_G)
        gen = lambda n: configs.outer_ss_state(n, perturb=1e-12, seed=42)
        ha = True
    elif workload == "nbody64":
        osys = ho.nbody(64)
        gen = lambda n: configs.plummer_nbody_state(64, n, seed=1234 + 42)
        ha = False
    else:
        osys = ho.nbody(2, masses=[1.0, 0.0])
        gen = lambda n: configs.two_body_state(n, perturb=1e-12, seed=42)
        ha = False
    # Bounded sample: a fixed set of systems propagated further and further (t += dt per call, exactly
    # like the GPU bench steps) until ~target_seconds of CPU work have been spent.
    n = width * threads * (1 if workload == "nbody64" else (32 if compact else 256))
    st = gen(n)
    # (The compact-mode legs: the oracle's interpreter with the arithmetic of the reference's compact mode - rolled loops
    # over the decomposition, running sums: the CPU analogue of src/taylor_02.cpp:1194-1260.)
    tmpl = ho.OracleIntegrator(osys, np.zeros(len(osys) * width), width, high_accuracy=ha, compact_mode=compact)
    # Outer Solar System / two-body: the jet is a compiled, fully unrolled, 8-wide vectorised function generated from
    # the oracle's decomposition (oracle/compiled_baseline.py - the CPU analogue of the reference's default-mode SIMD
    # JIT, checked bit by bit against the interpreter in tests/test_oracle_golden.py); N = 64 (18 663 u variables x
    # 20 orders) stays on the interpreter.
    how, compile_s = "oracle C interpreter, gcc -O2 -march=native -ffp-contract=off", 0.0
    if compact:
        how = "oracle C interpreter with the compact-mode arithmetic (running sums), gcc -O2 -march=native -ffp-contract=off"
    elif workload != "nbody64":
        import compiled_baseline as cb

        compile_s = cb.install(tmpl, fast=True)
        how = ("compiled straight-line C generated from the oracle's decomposition (oracle/compiled_baseline.py: default-mode "
               "operation order, gcc vector extensions, -O3 -march=native -ffp-contract=fast, built in %.1f s)" % compile_s)
    import ctypes as _ct

    thi, tlo = np.zeros(n), np.zeros(n)
    oc = np.zeros(n, dtype=np.int64)
    mn, mx = np.zeros(n), np.zeros(n)
    ns = np.zeros(n, dtype=np.int64)
    pr = np.zeros(max(tmpl.n_par, 1) * n)
    st = np.ascontiguousarray(st.reshape(-1))
    tot, el, t_cur, calls = 0, 0.0, 0.0, 0
    while el < target_seconds and calls < 10000:
        t_cur += dt
        t0 = time.perf_counter()
        tot += int(ho._lib().hy_oracle_ensemble_propagate_until(
            _ct.byref(tmpl._prog), _ct.c_int64(n), width, ho._p(st), ho._p(pr), ho._p(thi), ho._p(tlo),
            _ct.c_double(t_cur), _ct.c_int64(0), ho._p(oc), ho._p(mn), ho._p(mx), ho._p(ns), threads))
        el += time.perf_counter() - t0
        calls += 1
    dt = t_cur
    return {
        "value": tot / el,
        "unit": "system-steps/s",
        "cores": threads,
        "threads": threads,
        "hw_threads": hw_threads,
        "cgroup_cpu_quota": CPU_QUOTA,
        "kind": "port",
        "per_core": tot / el / threads,
        "sample": "%d %s systems propagated to t=%g (%d system-steps) in %.1f s; %s; batch width %d (lock-step batches like "
        "the reference's batch mode), one OpenMP worker per usable physical core (%d workers; %d hardware threads visible, cgroup "
        "CPU quota %s; OMP_PLACES=cores) over batches" % (n, workload, dt, tot, el, how, width, threads, hw_threads,
                                                          "%.1f CPUs" % CPU_QUOTA if CPU_QUOTA is not None else "none"),
    }


def kernel_sha(ta):
    import hashlib

    return hashlib.sha256(ta.hip_source.encode()).hexdigest()[:16]


def pmc_traffic(sha, n_systems):
    """HBM traffic per launch of the stepper kernel from the committed rocprofv3 PMC summaries
    (profiles/*_pmc.json, separate --pmc FETCH_SIZE / WRITE_SIZE passes): only used if the summary was
    taken on exactly this kernel (same generated source) and ensemble size."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if d.get("kernel_sha256") == sha and d.get("systems_per_gpu") == n_systems:
            # Calibrated on known-byte streams (profiles/r02_pmc_calibration.json): traffic = 2 x FETCH_SIZE + WRITE_SIZE.
            # Per system-step of the profiled launches (adaptive step counts differ from launch to launch).
            steps = d.get("bench_line_of_profiled_run", {}).get("config", {}).get("system_steps_per_launch")
            if steps:
                return d["per_launch_avg"]["traffic_bytes_fetch_x2"] / steps, os.path.basename(path)
    return None, None


def measure_traffic(workload, n):
    """HBM traffic of the stepper kernel from PMC passes taken NOW, on this box and this very library (`--measure-traffic`):
    two child runs of this script under rocprofv3 - `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate passes, counters
    only (never combined with tracing domains, as the guide's HBM section prescribes) -, the per-dispatch values of
    hy_taylor summed as 2 x FETCH_SIZE + WRITE_SIZE (the gfx950 correction calibrated in profiles/*_pmc_calibration.json)
    and divided by the system-steps of the same launches (the child prints them). Returns (bytes per system-step, source)
    or (None, reason)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    rp = shutil.which("rocprofv3")
    if rp is None:
        return None, "rocprofv3 is not on this box"
    vals, steps_per_launch = {}, None
    tmp = tempfile.mkdtemp(prefix="hy_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out_dir = os.path.join(tmp, ctr)
            cmd = [rp, "--pmc", ctr, "-d", out_dir, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), "--workload", workload,
                   "--systems", str(n), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra-workloads"]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd="/tmp", env=env)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                return None, "the %s pass failed (rc %d): %s" % (ctr, r.returncode, r.stderr[-300:])
            steps_per_launch = json.loads(line[-1])["config"]["system_steps_per_launch"]
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(out_dir) for f in fs if f.endswith(".db")]
            if not dbs:
                return None, "no rocpd database from the %s pass" % ctr
            cur = sqlite3.connect(dbs[0]).cursor()
            per = [row[0] for row in cur.execute("select value from counters_collection where kernel_name = 'hy_taylor' and "
                                                 "counter_name = ? order by dispatch_id", (ctr,))]
            if len(per) < 2:
                return None, "no hy_taylor dispatches in the %s pass" % ctr
            timed = per[1:]  # (the first launch is the warmup)
            vals[ctr] = sum(timed) / len(timed) * 1024.0  # KiB as reported -> bytes per launch
    except Exception as e:  # a measurement aid must never cost the line
        return None, "%s: %s" % (type(e).__name__, e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    traffic = 2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]
    return traffic / steps_per_launch, "same-run rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of this command (2 x FETCH + WRITE)"


def run_workload(ctx, workload, n, steps, warmup):
    """One timed leg: `warmup` untimed + `steps` timed propagate_until() calls over n systems per GPU. Returns the bench
    line of the leg on rank 0 (None elsewhere)."""
    torch, hy, configs, hens, dist = ctx["torch"], ctx["hy"], ctx["configs"], ctx["hens"], ctx["dist"]
    rank, world, distributed, dev, dev_index = ctx["rank"], ctx["world"], ctx["distributed"], ctx["dev"], ctx["dev_index"]
    out = None
    t_build = time.perf_counter()
    # One process per GPU: the integrator lives on this rank's device ordinal.
    ta, st, dt = make_integrator(hy, configs, workload, n, seed=42 + rank, device=dev_index)
    build_s = time.perf_counter() - t_build

    # Inputs resident in HBM before the timed region; kernels on torch's current stream so that
    # torch.cuda.Event (HIP events) brackets them.
    ta.set_stream(torch.cuda.current_stream().cuda_stream)
    view = torch.as_tensor(ta.device_array("state"), device=dev)
    view.copy_(torch.from_numpy(st))
    torch.cuda.synchronize()
    ta.mark_device_modified()

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # Step counters live on the device: accumulate them with a device-side reduction per call (same
    # stream, no host synchronisation inside the timed region).
    nsteps_view = torch.as_tensor(ta.device_array("n_steps"), device=dev)
    steps_acc = torch.zeros(max(steps, warmup, 1), dtype=torch.int64, device=dev)
    t_cur = 0.0
    for k in range(warmup):
        # NOTE: the warmup runs the complete step, reduction included (the first use of a torch kernel
        # costs ~20 ms of lazy code loading).
        t_cur += dt
        ta.propagate_until(t_cur)
        steps_acc[k] = nsteps_view.sum()
    barrier()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    steps_acc.zero_()
    t0 = time.perf_counter()
    for k in range(steps):
        t_cur += dt
        ev[k][0].record()
        ta.propagate_until(t_cur)
        ev[k][1].record()
        steps_acc[k] = nsteps_view.sum()
    barrier()
    elapsed = time.perf_counter() - t0
    # Shader clock right after the timed region (the FP64-dense steppers are power limited: 2.0 - 2.4 GHz depending on the
    # kernel and the box; the sustained value during a launch is in profiles/*_sq_counters.json, GRBM_GUI_ACTIVE).
    try:
        sclk_mhz = int(torch.cuda.clock_rate(dev))
    except Exception:
        sclk_mhz = None
    # The path partitions with no exchange step (independent systems, the reference's ensemble is a
    # parallel_for over copies, src/ensemble_propagate.cpp:203-219): the optional gather of the final states
    # (heyoka_amd/ensemble.py, RCCL all-gather over xGMI) is exercised here, outside of the timed region.
    gathered = None
    gather_ms = None
    gather_err = None
    if distributed:
        try:
            tg = time.perf_counter()
            gathered = hens.all_gather_states(view)
            torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - tg) * 1e3
            # Every rank must hold the final state of all the N x systems ICs, its own shard in place.
            if tuple(gathered.shape) != (view.shape[0], world * n):
                raise RuntimeError("gathered shape %s, expected %s" % (tuple(gathered.shape), (view.shape[0], world * n)))
            if not torch.equal(gathered[:, rank * n:(rank + 1) * n], view):
                raise RuntimeError("the gathered state does not contain this rank's shard at its place")
        except Exception as e:  # the optional collective must never cost the measurement
            gather_err = "%s: %s" % (type(e).__name__, e)

    # Per-launch kernel durations (HIP events on the launch stream) and step counts.
    call_ms = [a.elapsed_time(b) for a, b in ev]  # torch events around the whole call (incl. small copies)
    kern_ms = list(ta.kernel_ms_history(steps))  # HIP events recorded right around each launch
    steps_per_call_all = steps_acc[: steps].cpu().numpy().astype(np.float64)
    steps_per_call = float(steps_per_call_all.mean())

    oc, mn, mx, ns = ta.propagate_res_arrays()
    ok = bool(np.all(oc == int(hy.taylor_outcome.time_limit)))

    local_steps = float(steps_per_call_all.sum())
    local = torch.tensor([elapsed, local_steps, float(np.mean(kern_ms))], dtype=torch.float64, device=dev)
    # What every rank actually ran on and measured (a slow or misplaced rank is invisible in the one aggregate): device
    # ordinal, PCI location, own elapsed time / system-steps / mean kernel time.
    props = torch.cuda.get_device_properties(dev)
    pci = [float(getattr(props, "pci_domain_id", -1)), float(getattr(props, "pci_bus_id", -1)), float(getattr(props, "pci_device_id", -1))]
    rank_rec = torch.tensor([elapsed, local_steps, float(np.mean(kern_ms)), float(dev_index)] + pci, dtype=torch.float64, device=dev)
    if distributed:
        mx_t = local.clone()
        dist.all_reduce(mx_t, op=dist.ReduceOp.MAX)
        sm_t = local.clone()
        dist.all_reduce(sm_t, op=dist.ReduceOp.SUM)
        elapsed_max = float(mx_t[0])
        total_steps = float(sm_t[1])
        recs = [torch.zeros_like(rank_rec) for _ in range(world)]
        dist.all_gather(recs, rank_rec)
        rank_recs = [r.cpu().numpy() for r in recs]
        ranks_seen = int(dist.get_world_size())
    else:
        elapsed_max = elapsed
        total_steps = local_steps
        rank_recs = [rank_rec.cpu().numpy()]
        ranks_seen = 1

    if rank == 0:
        from heyoka_amd import codegen_check, roofline

        n_eq, n_u = ta.dim, ta.n_uvars
        f_alg, b_tape = roofline.counts_for(ta)
        f_survey = WORKLOADS[workload]
        value = total_steps / elapsed_max
        k_ms = float(np.mean(kern_ms))
        per_launch_steps = float(steps_per_call)
        achieved_gbs = b_tape * per_launch_steps / (k_ms * 1e-3) / 1e9
        achieved_tflops = f_alg * per_launch_steps / (k_ms * 1e-3) / 1e12
        traffic_per_step, traffic_src = pmc_traffic(kernel_sha(ta), n)
        if ctx.get("measure_traffic"):
            # (A same-run counter pass beats a committed summary matched by sha.)
            m_tps, m_src = measure_traffic(workload, n)
            if m_tps is not None:
                traffic_per_step, traffic_src = m_tps, m_src
            elif traffic_per_step is None:
                traffic_src = "not measured: " + str(m_src)
        traffic = traffic_per_step * per_launch_steps if traffic_per_step else None
        # Which ceiling binds. The tape model B_tape describes a stepper that streams its jets through HBM
        # (block / table modes). The cluster and register-resident steppers keep the jets on chip: their measured HBM
        # traffic is a fraction of a percent of the peak and the binding ceiling is the FP64 arithmetic rate (78.6
        # TFLOP/s vector = matrix peak for f64 on MI355X; the recurrences have no dense contraction, MFMA itself is
        # unused) - reported under the contract's "mfma" label with the algorithmic flop count F_alg.
        # Without a counter file for this exact kernel: the cluster / register-resident steppers keep their jets on chip
        # (FP64-bound), the block / table steppers stream them through HBM (tape model).
        mode_str = ta.hip_source_mode
        on_chip = mode_str.startswith("cluster") or "jets in registers" in mode_str or mode_str.startswith("unrolled")
        on_chip = on_chip and "global scratch" not in mode_str
        # (Block mode v2 keeps the low rows of the cluster histories in registers and recomputes three of five members:
        # the B_tape model counts bytes which it does not move.)
        partly_on_chip = (not on_chip) and "v2 cluster phase" in mode_str
        # (The staged table stepper keeps the tape of a system in LDS: the B_tape figure is the ALGORITHMIC traffic of the
        # one-lane-per-system formulation - SURVEY 8d's official scale for this path -, not bytes which reach HBM.)
        staged = mode_str.startswith("table") and "table mode (staged)" in mode_str
        hbm_util = (traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else (0.0 if on_chip else min(1.0, achieved_gbs / HBM_PEAK_GBS))
        compute_bound = achieved_tflops / FP64_PEAK_TFLOPS > hbm_util
        # (How the binding ceiling was decided: from counter traffic of this very kernel, or - without a matching summary
        # under profiles/ - from where the generator keeps the jets.)
        bound_basis = ("measured: HBM traffic of this kernel from %s" % traffic_src) if traffic else (
            "by construction: the jets of this stepper live in %s (no counter summary for this kernel under profiles/)"
            % ("LDS / registers, HBM only sees the state" if on_chip else "an HBM tape (B_tape model)"))
        if compute_bound:
            # NOTE: "fp64_valu" = the FP64 *vector* rate (16 lanes x FMA per clock and SIMD = 78.6 TFLOP/s at 2.4 GHz;
            # equal to the f64 matrix peak of gfx950, which is why the contract files it under the "mfma" ceiling -
            # MFMA itself is unused: the recurrences are elementwise).
            roof = {"bound": "fp64_valu", "bound_contract_class": "mfma", "achieved": achieved_tflops,
                    "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved_tflops / FP64_PEAK_TFLOPS}
        elif traffic and (partly_on_chip or achieved_gbs > HBM_PEAK_GBS):
            # The tape model counts bytes this stepper no longer moves (rows cached in registers, members recomputed; or a
            # wave-cluster stepper which keeps the histories in registers and only the jets of the state variables in global
            # scratch): a fraction above 1 is not a roofline fraction. The bytes it DOES move are the counter traffic of this very
            # kernel: achieved = measured HBM bytes per second; the tape-model figure stays as hbm_tape_model_frac.
            meas_gbs = traffic / (k_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": meas_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": meas_gbs / HBM_PEAK_GBS,
                    "achieved_basis": "HBM counter traffic of this kernel (2 x FETCH_SIZE + WRITE_SIZE per launch) / kernel time"}
        else:
            roof = {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": min(1.0, achieved_gbs / HBM_PEAK_GBS) if partly_on_chip else achieved_gbs / HBM_PEAK_GBS,
                    "achieved_basis": "algorithmic tape bytes (B_tape, SURVEY 8d) / kernel time"
                    + ("; capped at 1: part of the tape stays on chip and no counter summary of this kernel is under profiles/"
                       if partly_on_chip else "")
                    + ("; the tape of this stepper lives in LDS (one system per workgroup): the figure is the algorithmic traffic "
                       "of the one-lane-per-system formulation on the 8 TB/s scale, what binds is the LDS latency of the "
                       "dependency chain of an order" if staged else "")}
        out = {
            "metric": "ODE systems x steps/sec (fp64)",
            "value": value,
            "unit": "system-steps/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": elapsed_max / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d %s per GPU, tol=eps (order 20), high_accuracy=%s, propagate_until in increments of %g time units"
                % (workload, n, WORKLOAD_ICS.get(workload, "perturbed ICs (perturb 1e-12, mt19937 seed 42+rank)"),
                   str(bool(ta.high_accuracy)).lower(), dt),
                "systems_per_gpu": n,
                "taylor_order": ta.order,
                "n_eq": n_eq,
                "n_uvars": n_u,
                "system_steps_per_launch": per_launch_steps,
                "all_outcomes_time_limit": ok,
                "integrator_build_s": build_s,
                "hiprtc_compile_s": ta.compile_seconds,
                "kernel_sha256": kernel_sha(ta),
                # Build id of libheyoka_amd.so = hash of the sources it was built from; the import refuses a library which
                # does not match the tree (heyoka_amd/_lib.py), so this is also the hash of heyoka_amd/csrc of this run.
                "library_build_id": hy.build_id(),
                # (The kernels are compiled on THIS box at construction: which hiprtc / HIP runtime did it.)
                "toolchain": hy.version(),
                # Rank by rank (index = rank): throughput from the rank's own clock, its mean kernel time, the device it
                # bound (ordinal in its process, PCI domain:bus:device), and the size of the communicator as the collective
                # library reports it.
                "per_rank_value": [float(r[1] / r[0]) for r in rank_recs],
                "per_rank_kernel_ms": [float(r[2]) for r in rank_recs],
                "per_rank_device": [{"ordinal": int(r[3]), "pci": "%04x:%02x:%02x" % (int(r[4]) & 0xFFFF, int(r[5]) & 0xFF, int(r[6]) & 0xFF)}
                                    for r in rank_recs],
                "rccl_ranks_seen": ranks_seen,
                "collective_backend": (ctx.get("backend") if distributed else None),
                "collective_timeout_s": (ctx.get("collective_timeout") if distributed else None),
                "untimed_final_state_all_gather_ms": gather_ms,
                "gathered_systems": (int(gathered.shape[1]) if gathered is not None else None),
                "gathered_bytes_per_rank": (int(gathered.numel()) * 8 if gathered is not None else None),
                "untimed_final_state_all_gather_error": gather_err,
            },
            "roofline": {
                **roof,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "bound_basis": bound_basis,
                "sclk_mhz_after_timed_region": sclk_mhz,
                "kernel": "hy_taylor",
                "kernel_ms_avg": k_ms,
                "call_ms_avg": float(np.mean(call_ms)),
                "algorithmic_flop_per_system_step": f_alg,
                "algorithmic_bytes_per_system_step": b_tape,
                # Launch by launch: adaptive step counts make the launches differ (and, for one system per workgroup,
                # the launch cannot end before its slowest system: see steps_per_system_last_launch).
                "per_launch": [{"kernel_ms": float(k), "system_steps": float(s_)} for k, s_ in zip(kern_ms, steps_per_call_all)],
                "steps_per_system_last_launch": {"mean": float(ns.mean()), "max": int(ns.max()), "min": int(ns.min()),
                                                 "p99": float(np.percentile(ns, 99))},
                "algorithmic_counts_source": "heyoka_amd/roofline.py on the decomposition of this integrator "
                "(%d u variables, order %d)" % (n_u, ta.order),
                # Round-1 basis (SURVEY 8d estimate of F_alg), for continuity only.
                "fp64_valu_frac_survey_falg": (f_survey * per_launch_steps / (k_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if f_survey else None,
                "kernel_resources": codegen_check.kernel_resources(ta.code_object),
                "kernel_mode": ta.hip_source_mode,
                # Both views, whichever binds.
                "fp64_valu_frac": achieved_tflops / FP64_PEAK_TFLOPS,
                "hbm_tape_model_frac": achieved_gbs / HBM_PEAK_GBS,
                # (> 1 in the tape model means the jets never travel: they live in registers / LDS - not skipped work.)
                "tape_on_chip": bool(on_chip or partly_on_chip or staged),
                "tape_on_chip_note": ("all of the jets in LDS / registers" if on_chip else
                                      ("part of the tape never travels: " + mode_str[mode_str.find("v2 cluster phase"):]
                                       + " - see hbm_measured_frac for the bytes which do") if partly_on_chip else
                                      ("the tape of a system lives in LDS for the whole step (staged table stepper)" if staged else
                                       "the tape streams through HBM")),
                "hbm_measured_frac": (traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            },
        }
        out["_dt"] = dt

    # Any rank's failed gather fails the job (every rank learns it: the exit status is decided in main()).
    if distributed:
        flag = torch.tensor([1.0 if gather_err is not None else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        out = {} if out is None else out
        out["_gather_failed"] = bool(flag.item() != 0.0)
    del ta, view, nsteps_view
    torch.cuda.empty_cache()
    return out


def divergence_leg(ctx, n, t_span=60.0):
    """Heterogeneous step counts: the headline ensemble is 1e-12-perturbed copies which all take the same
    number of steps. Here: outer Solar System with 1e-2 relative perturbations and PER-LANE final times T * U(0.5, 1.5), against
    the same perturbed ensemble with the common final time T; one untimed + two timed launches each, kernel time from the HIP
    events around the launch. Reference behaviour for comparison: the lanes of a batch run in lock step and idle until the
    slowest is done (src/taylor_adaptive_batch.cpp:1378-1460)."""
    torch, hy, configs = ctx["torch"], ctx["hy"], ctx["configs"]
    sys_ = hy.model.nbody(6, masses=configs.OUTER_SS_MASSES, Gconst=configs.OUTER_SS_G)
    st = configs.outer_ss_state(n, perturb=1e-2, seed=4242)
    rng = np.random.RandomState(99)
    frac = rng.uniform(0.5, 1.5, n)
    res = {}
    for name, per_lane in (("uniform", False), ("divergent", True)):
        ta = hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=True, device=ctx["dev_index"])
        rates, spread = [], None
        t_prev = np.zeros(n)
        for k in range(3):
            t_fin = t_prev + t_span * (frac if per_lane else 1.0)
            ta.propagate_until(t_fin if per_lane else float(t_fin[0]))
            oc, _, _, ns = ta.propagate_res_arrays()
            ms = list(ta.kernel_ms_history(1))[-1]
            if k > 0:
                rates.append(float(ns.sum()) / (ms * 1e-3))
                spread = [int(ns.min()), float(ns.mean()), int(ns.max())]
            t_prev = t_fin
        res[name] = {"value": float(np.mean(rates)), "steps_per_system_min_mean_max": spread,
                     "all_outcomes_time_limit": bool(np.all(oc == int(hy.taylor_outcome.time_limit)))}
        del ta
        torch.cuda.empty_cache()
    return {
        "config": {"workload": "outer_ss_divergent: %d ICs, relative perturbation 1e-2, per-lane final times T * U(0.5, 1.5), "
                               "T = %g; compared with the same ensemble and a common final time" % (n, t_span),
                   "systems_per_gpu": n},
        "unit": "system-steps/s", "value": res["divergent"]["value"], "uniform_value": res["uniform"]["value"],
        "divergent_over_uniform": res["divergent"]["value"] / res["uniform"]["value"],
        "steps_per_system_min_mean_max": res["divergent"]["steps_per_system_min_mean_max"],
        "uniform_steps_per_system_min_mean_max": res["uniform"]["steps_per_system_min_mean_max"],
        "all_outcomes_time_limit": res["divergent"]["all_outcomes_time_limit"] and res["uniform"]["all_outcomes_time_limit"],
    }


def events_leg(ctx, n, n_steps=6):
    """Lock-step steps with event detection on an ensemble in which events actually FIRE (reference: step_e,
    src/taylor_00.cpp:592-710, detection src/detail/event_detection.cpp, event branch of step_impl()
    src/taylor_adaptive_batch.cpp:727-1030). The systems are first spread over their orbits (per-lane propagation times
    U(0, 30 yr)), then stepped with (a) two non-terminal events - Jupiter and Saturn crossing the plane y = 0, any direction:
    ~12 % + ~5 % of the systems fire per step - and (b) the same plus a TERMINAL event (Uranus crossing y = 0, callback
    continues: truncated step, state redone from the Taylor coefficients, cooldown), against the event-free step() of the
    same ensemble. The callbacks are the library's counting callbacks (a Python callback per event would be the only thing
    measured at 10^5 events per step). Reported: wall time per step (stepper + detection + bookkeeping + host side), the
    events per step, and from a second, phase-timed set of steps the split over the kernels."""
    import time as _time

    torch, hy, configs = ctx["torch"], ctx["hy"], ctx["configs"]
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    sys_ = hy.model.nbody(6, masses=M, Gconst=G)
    st0 = configs.outer_ss_state(n, perturb=1e-6, seed=4243)
    rng = np.random.RandomState(4244)
    spread = hy.taylor_adaptive_batch(sys_, st0, n, high_accuracy=True, device=ctx["dev_index"])
    spread.propagate_until(rng.uniform(0.0, 30.0, n))
    st = np.array(spread.state)
    del spread
    torch.cuda.empty_cache()
    y1, y2, y3 = hy.make_vars("y_1", "y_2", "y_3")
    res = {}
    for name in ("event_free", "non_terminal", "with_terminal"):
        c_nt, c_t = hy.native_event_counter(), hy.native_event_counter()
        kw = {}
        if name != "event_free":
            kw["nt_events"] = [hy.nt_event(y1, c_nt), hy.nt_event(y2, c_nt)]
        if name == "with_terminal":
            kw["t_events"] = [hy.t_event(y3, c_t)]
        ta = hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=True, device=ctx["dev_index"], **kw)
        # (Untimed steps first: the buffers of the Taylor coefficients - 3 GB for 1 048 576 systems - are touched for the
        # first time by whichever workgroups hold a possible event, a different set at every step.)
        for _ in range(8 if name != "event_free" else 2):
            ta.step()
        torch.cuda.synchronize()
        c0, ct0 = c_nt.value, c_t.value
        t0 = _time.perf_counter()
        for _ in range(n_steps):
            ta.step()
        torch.cuda.synchronize()
        el = (_time.perf_counter() - t0) / n_steps
        r = {"s_per_step": el, "mode": ta.hip_source_mode[-110:], "nt_events_per_step": (c_nt.value - c0) / n_steps,
             "t_events_per_step": (c_t.value - ct0) / n_steps}
        if name != "event_free":
            # Phase split: a second set of steps with a synchronisation after every phase.
            s0 = ta.event_stats
            ta.set_event_timing(True)
            for _ in range(n_steps):
                ta.step()
            ta.set_event_timing(False)
            s1 = ta.event_stats
            r["phase_ms_per_step"] = {k: (s1[k] - s0[k]) / n_steps for k in s1 if k.startswith("ms_")}
            r["tc_regeneration_launches"] = s1["tc_regeneration_launches"]
            r["systems_with_events_per_step"] = (s1["systems_with_events"] - s0["systems_with_events"]) / n_steps
            r["outcomes_ok"] = bool(np.all(np.isfinite(np.asarray(ta.time))))
            # (HIP-event durations of the stepper launches of the phase-timed steps: the kernel itself, whatever the wall
            # clock of the phase says.)
            r["stepper_kernel_ms"] = [round(x, 3) for x in ta.kernel_ms_history(n_steps)]
        res[name] = r
        del ta
        torch.cuda.empty_cache()
    ev, fr, tv = res["non_terminal"], res["event_free"]["s_per_step"], res["with_terminal"]
    return {
        "config": {"workload": "outer_ss_plane_crossings: %d ICs spread over 30 yr of their orbits, lock-step step() with Jupiter / "
                               "Saturn crossing y = 0 as non-terminal events (library-side counting callbacks), a variant with Uranus "
                               "crossing y = 0 as a terminal event (continuing), against the event-free step()" % n,
                   "systems_per_gpu": n, "stepper": ev["mode"]},
        "unit": "system-steps/s", "value": n / ev["s_per_step"], "event_free_value": n / fr, "ms_per_step": ev["s_per_step"] * 1e3,
        "event_free_ms_per_step": fr * 1e3, "with_events_over_event_free_time": ev["s_per_step"] / fr,
        "events_detected_per_step": ev["nt_events_per_step"], "fraction_of_systems_with_an_event_per_step":
        ev.get("systems_with_events_per_step", 0.0) / n, "phase_ms_per_step": ev.get("phase_ms_per_step"),
        "stepper_kernel_ms": ev.get("stepper_kernel_ms"),
        "tc_regeneration_launches": ev.get("tc_regeneration_launches"),
        "terminal_variant": {"ms_per_step": tv["s_per_step"] * 1e3, "over_event_free_time": tv["s_per_step"] / fr,
                             "terminal_events_per_step": tv["t_events_per_step"], "nt_events_per_step": tv["nt_events_per_step"],
                             "phase_ms_per_step": tv.get("phase_ms_per_step"), "stepper_kernel_ms": tv.get("stepper_kernel_ms"),
                             "tc_regeneration_launches": tv.get("tc_regeneration_launches"),
                             "stepper": tv["mode"]},
    }


def long_horizon_leg(ctx, n, t_final=1.0e4, n_snap=8, margin=2.0):
    """The reference benchmark's protocol on the headline configuration (benchmark/outer_ss_long_term_batch.cpp:229-241:
    snapshots of the state at fixed times through DENSE OUTPUT while the integration runs; :339-365: a long horizon and
    the final energy error): n outer Solar Systems to t_final years, n_snap equally spaced snapshots, the relative energy
    error of every snapshot from a compiled function evaluated on the device - no state leaves the GPU. Between two
    snapshots the ensemble runs through the device-resident propagate_until() (one launch per stretch); around a snapshot
    time t_k the device-resident propagate_grid() takes it from t_k - margin over the grid (t_k - margin, t_k, t_k + margin):
    t_k is an interior grid point, i.e. sampled by dense output inside the step which crosses it, not by a clamped step. (A
    propagate_grid() over the whole horizon writes the Taylor coefficients of every system at every step - 6 GB per step
    for 1 048 576 systems -: 230 s for the same 1e4 yr.) The reference runs 1e6 yr on a handful of systems; 1 048 576
    systems x 1e4 yr is 1e10 system-years."""
    import time as _time

    torch, hy, configs = ctx["torch"], ctx["hy"], ctx["configs"]
    M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
    sys_ = hy.model.nbody(6, masses=M, Gconst=G)
    st = configs.outer_ss_state(n, perturb=1e-12, seed=42)
    ta = hy.taylor_adaptive_batch(sys_, None, n, high_accuracy=True, device=ctx["dev_index"])
    dev = ctx["dev"]
    view = torch.as_tensor(ta.device_array("state"), device=dev)
    view.copy_(torch.from_numpy(st))
    torch.cuda.synchronize()
    ta.mark_device_modified()
    cf = hy.cfunc([hy.model.nbody_energy(6, masses=M, Gconst=G)], sys_.vars)
    e0 = torch.empty(n, dtype=torch.float64, device=dev)
    ek = torch.empty_like(e0)
    cf.eval_device(e0.data_ptr(), ta.device_array("state").ptr, n)
    out = torch.empty((3, 36, n), dtype=torch.float64, device=dev)
    nsteps_view = torch.as_tensor(ta.device_array("n_steps"), device=dev)
    torch.cuda.synchronize()
    errs, total, ok = [], 0, True
    t_snap = [t_final * (k + 1) / n_snap for k in range(n_snap)]
    t0 = _time.perf_counter()
    for tk in t_snap:
        ta.propagate_until(tk - margin)
        total += int(nsteps_view.sum())
        last = tk + margin if tk < t_final else tk + margin
        ta.propagate_grid_device(np.array([tk - margin, tk, last]), out.data_ptr())
        oc, _, _, ns = ta.propagate_res_arrays()
        total += int(ns.sum())
        ok = ok and bool(np.all(oc == int(hy.taylor_outcome.time_limit)))
        cf.eval_device(ek.data_ptr(), out[1].data_ptr(), n)
        errs.append(float(((ek - e0) / e0).abs().max()))
    ta.synchronize()
    torch.cuda.synchronize()
    wall = _time.perf_counter() - t0
    return {
        "config": {"workload": "outer_ss_long_horizon: %d ICs (perturb 1e-12) to %g yr, %d snapshots through dense output "
                               "(propagate_until between them, a device-resident propagate_grid across each), energy monitor = "
                               "compiled function on the device" % (n, t_final, n_snap),
                   "systems_per_gpu": n, "reference_protocol": "benchmark/outer_ss_long_term_batch.cpp:229-241, :339-365"},
        "unit": "system-steps/s", "value": total / wall, "wall_s": wall, "system_steps": float(total), "steps_per_system_mean": total / n,
        "system_years": n * t_final, "max_rel_energy_error_per_snapshot": errs, "max_rel_energy_error": max(errs),
        "all_outcomes_time_limit": ok,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--systems", type=int, default=0, help="systems per GPU (default: the BASELINE.json size)")
    ap.add_argument("--workload", default="outer_ss", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="skip the short two-body / N = 64 legs which follow the outer-Solar-System measurement at N = 1")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-long-horizon", action="store_true", help="skip the long-horizon leg of the extra workloads (~30 s)")
    ap.add_argument("--long-horizon-years", type=float, default=1.0e4)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL over xGMI)")
    ap.add_argument("--measure-traffic", action="store_true",
                    help="fill roofline.traffic from rocprofv3 PMC passes taken now (two child runs per workload, ~1 min each) "
                    "instead of the committed summary under profiles/ matched by the sha of the kernel")
    ap.add_argument("--collective-timeout", type=float, default=180.0,
                    help="seconds after which a collective that does not complete fails the run instead of hanging it")
    ap.add_argument("--single-device", action="store_true",
                    help="debug: all ranks share GPU 0 (use with --backend gloo to exercise the N > 1 path on one GPU)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # (A process group also for ONE rank when the launcher set up a rendezvous - `torch.distributed.run --nproc-per-node 1`:
    # RCCL's all-gather then runs on a single-GPU box as well; a plain `python bench.py` stays without a group.)
    distributed = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dev_index = 0 if args.single_device else local_rank
        torch.cuda.set_device(dev_index)
        import datetime

        # Every collective is bounded: a rank which never arrives fails the run after the timeout (the process-group
        # watchdog aborts the communicator) instead of eating the driver's whole time budget.
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        tmo = datetime.timedelta(seconds=args.collective_timeout)
        if args.backend == "nccl":
            # Binding the communicator to this rank's device up front: barrier() and the first collective do not have to
            # guess the device (and RCCL initialises eagerly, before the timed region).
            try:
                dist.init_process_group(backend=args.backend, device_id=torch.device("cuda", dev_index), timeout=tmo)
            except Exception:  # (eager initialisation not available / failed: fall back to the lazy one)
                if dist.is_initialized():
                    dist.destroy_process_group()
                dist.init_process_group(backend=args.backend, timeout=tmo)
        else:
            dist.init_process_group(backend=args.backend, timeout=tmo)
    else:
        dev_index = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)

    import heyoka_amd as hy
    from heyoka_amd import configs
    from heyoka_amd import ensemble as hens

    ctx = dict(torch=torch, hy=hy, configs=configs, hens=hens, dist=(dist if distributed else None), rank=rank, world=world,
               distributed=distributed, dev=dev, dev_index=dev_index, backend=args.backend,
               collective_timeout=args.collective_timeout, measure_traffic=bool(args.measure_traffic) and world == 1)
    n = args.systems if args.systems > 0 else DEFAULT_SYSTEMS[args.workload]
    out = run_workload(ctx, args.workload, n, args.steps, args.warmup)
    gather_failed = bool(out.pop("_gather_failed", False)) if out is not None else False
    if rank == 0:
        if world == 1 and not args.no_extra_workloads and args.workload == "outer_ss":
            # The other two single-GPU measurement points of BASELINE.json (configs 3 and 5) as short legs of the same run,
            # each with its own roofline: the two-body problem (small DAG, jets in registers: FP64-bound) and
            # model::nbody(64) (block mode, jets on a tape: HBM-bound, counter traffic when a matching summary is
            # committed).
            extra = []
            for wl, st_, wu_ in (("two_body", 4, 2), ("nbody64", 2, 1)):
                try:
                    r = run_workload(ctx, wl, DEFAULT_SYSTEMS[wl], st_, wu_)
                    leg = {k: r[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline")}
                    if not args.no_cpu_baseline:
                        leg["cpu_baseline"] = cpu_baseline(wl, r["_dt"], 5.0)
                    extra.append(leg)
                except Exception as e:  # an auxiliary leg must never cost the headline line
                    extra.append({"config": {"workload": wl}, "error": "%s: %s" % (type(e).__name__, e)})
            try:
                # A system off the headline shape: equal (default) masses. No CPU baseline of its own.
                r = run_workload(ctx, "nbody6_default_masses", DEFAULT_SYSTEMS["nbody6_default_masses"], 3, 1)
                extra.append({k: r[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline")})
            except Exception as e:
                extra.append({"config": {"workload": "nbody6_default_masses"}, "error": "%s: %s" % (type(e).__name__, e)})
            # Round 6: the general-DAG paths (see WORKLOADS), each with its own roofline.
            for wl in ("outer_ss_compact_mode", "outer_ss_forced_table", "outer_ss_forced_table_hbm_tape", "nbody6_j2_mixed",
                       "sine_lattice16_mixed"):
                try:
                    r = run_workload(ctx, wl, DEFAULT_SYSTEMS[wl], 3, 1)
                    leg = {k: r[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline")}
                    if wl == "outer_ss_forced_table" and not args.no_cpu_baseline:
                        leg["cpu_baseline"] = cpu_baseline(wl, r["_dt"], 4.0)
                    extra.append(leg)
                except Exception as e:
                    extra.append({"config": {"workload": wl}, "error": "%s: %s" % (type(e).__name__, e)})
            try:
                extra.append(divergence_leg(ctx, DEFAULT_SYSTEMS["outer_ss"]))
            except Exception as e:
                extra.append({"config": {"workload": "outer_ss_divergent"}, "error": "%s: %s" % (type(e).__name__, e)})
            try:
                extra.append(events_leg(ctx, DEFAULT_SYSTEMS["outer_ss"]))
            except Exception as e:
                extra.append({"config": {"workload": "outer_ss_plane_crossings"}, "error": "%s: %s" % (type(e).__name__, e)})
            if not args.no_long_horizon:
                try:
                    extra.append(long_horizon_leg(ctx, DEFAULT_SYSTEMS["outer_ss"], args.long_horizon_years))
                except Exception as e:
                    extra.append({"config": {"workload": "outer_ss_long_horizon"}, "error": "%s: %s" % (type(e).__name__, e)})
            out["extra_workloads"] = extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, out.pop("_dt"), args.cpu_seconds)
        out.pop("_dt", None)
        print(json.dumps(out))

    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if distributed and gather_failed:
        # The collective of the multi-GPU path is part of what a run with a process group has to prove: the line above
        # carries the error text, the exit status fails the run.
        sys.exit(3)


if __name__ == "__main__":
    main()
