#!/bin/bash
# Round 5, GPU call 28: the default bench line with the default-masses leg.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( time timeout 900 python bench.py ) > gpurun_out/r05_run28_bench_default.log 2>&1
tail -4 gpurun_out/r05_run28_bench_default.log | cut -c1-200
