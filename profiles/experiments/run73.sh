#!/bin/bash
# Event equations inside the stepper: tests, then the cost of a step with ONE event of each kind on 1 048 576 systems (kernel
# times from the HIP events are not available for the auxiliary kernels: rocprofv3 kernel trace).
cd "${GRAFT_REPO_ROOT:-.}"
R=$(pwd)
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "inside_the_stepper or time_dependent or compact_taylor or events" 2>&1 | tail -6
mkdir -p gpurun_out/r73
for ev in ${EVS:-linear d2}; do
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r73/kt_$ev -o kt -- python $R/profiles/experiments/events_scale.py --systems 1048576 --skip-lane-stepper --steps 6 --event $ev > $R/gpurun_out/r73/run_$ev.log 2>&1
  cd $R
  python profiles/summarize_rocprof.py gpurun_out/r73/summary_$ev "$(find gpurun_out/r73/kt_$ev -name '*.db' | head -1)" > gpurun_out/r73/summary_$ev.log 2>&1
  echo "== $ev"; tail -2 gpurun_out/r73/run_$ev.log | cut -c1-200; head -12 gpurun_out/r73/summary_${ev}_kernel_stats.txt; tail -3 gpurun_out/r73/summary_${ev}_kernel_stats.txt
done
find gpurun_out/r73 -name '*.db' -delete
