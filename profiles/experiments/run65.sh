#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
R=$(pwd)
mkdir -p gpurun_out/r65
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r65/kt -o kt -- python $R/profiles/experiments/events_scale.py --systems 1048576 --skip-lane-stepper --steps 6 > $R/gpurun_out/r65/run.log 2>&1
cd $R
python profiles/summarize_rocprof.py gpurun_out/r65/summary "$(find gpurun_out/r65/kt -name '*.db' | head -1)" > gpurun_out/r65/summary.log 2>&1
head -30 gpurun_out/r65/summary_kernel_stats.txt 2>/dev/null || tail -20 gpurun_out/r65/summary.log
find gpurun_out/r65 -name '*.db' -size +8M -delete
