#!/bin/bash
# Round 5, GPU call 15: default bench run with the main thread's CPU affinity restored after the CPU baselines (events leg),
# SQ counters of a single-step launch.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python bench.py ) > gpurun_out/r05_run15_bench_default.log 2>&1
tail -4 gpurun_out/r05_run15_bench_default.log | cut -c1-300
bash profiles/experiments/sq_single_step.sh > gpurun_out/r05_run15_sq.log 2>&1
tail -45 gpurun_out/r05_run15_sq.log
