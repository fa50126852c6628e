#!/bin/bash
# Round 5, GPU call 24: profile collection of the final outer-SS kernel (bookkeeping block in the slab), default bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 bash profiles/collect.sh r05 outer_ss > gpurun_out/r05_run24_collect.log 2>&1
tail -3 gpurun_out/r05_run24_collect.log | cut -c1-200
cp gpurun_out/collect_r05/r05_outer_ss_pmc.json profiles/ 2>/dev/null
( time timeout 900 python bench.py ) > gpurun_out/r05_run24_bench_default.log 2>&1
tail -4 gpurun_out/r05_run24_bench_default.log | cut -c1-200
