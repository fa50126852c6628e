#!/bin/bash
# Round 5, GPU call 14: the whole GPU suite on the final tree, then the round-5 profile collection and the default bench run.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05_run14_pytest.log 2>&1
tail -8 gpurun_out/r05_run14_pytest.log
timeout 1800 bash profiles/collect.sh r05 outer_ss two_body nbody64 > gpurun_out/r05_run14_collect.log 2>&1
tail -5 gpurun_out/r05_run14_collect.log
( time timeout 900 python bench.py ) > gpurun_out/r05_run14_bench_default.log 2>&1
tail -4 gpurun_out/r05_run14_bench_default.log | cut -c1-600
