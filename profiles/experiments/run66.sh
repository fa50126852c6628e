#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
R=$(pwd)
mkdir -p gpurun_out/r66
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r66/kt -o kt -- env HEYOKA_AMD_V2_BS=256 python $R/profiles/experiments/events_scale.py --systems 1048576 --skip-lane-stepper --steps 6 > $R/gpurun_out/r66/run.log 2>&1
cd $R
python profiles/summarize_rocprof.py gpurun_out/r66/summary "$(find gpurun_out/r66/kt -name '*.db' | head -1)" > gpurun_out/r66/summary.log 2>&1
head -30 gpurun_out/r66/summary_kernel_stats.txt 2>/dev/null || tail -20 gpurun_out/r66/summary.log
find gpurun_out/r66 -name '*.db' -size +8M -delete
