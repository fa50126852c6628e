#!/bin/bash
# Kernel trace of the divergent-ensemble leg (per-lane final times, 1e-2 perturbations) against the uniform one.
cd "${GRAFT_REPO_ROOT:-.}"
R=$(pwd)
mkdir -p gpurun_out/r77
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r77/kt -o kt -- python $R/profiles/experiments/divergence.py > $R/gpurun_out/r77/run.log 2>&1
cd $R
python profiles/summarize_rocprof.py gpurun_out/r77/summary "$(find gpurun_out/r77/kt -name '*.db' | head -1)" > gpurun_out/r77/summary.log 2>&1
head -14 gpurun_out/r77/summary_kernel_stats.txt; grep -E "value|over_uniform|min_mean" -A3 gpurun_out/r77/run.log | head -30
find gpurun_out/r77 -name '*.db' -delete
