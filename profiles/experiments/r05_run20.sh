#!/bin/bash
# Round 5, GPU call 20: the default bench run on the final tree (the line the driver will produce).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
( time timeout 900 python bench.py ) > gpurun_out/r05_run20_bench_default.log 2>&1
tail -4 gpurun_out/r05_run20_bench_default.log | cut -c1-200
