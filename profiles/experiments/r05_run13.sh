#!/bin/bash
# Round 5, GPU call 13: single-step specialisation of the one-lane-per-pair stepper (mode as a compile-time constant) against
# the general kernel in lock-step sweeps of 1 048 576 systems; raw stepper ABI test.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for sk in 0 1; do
  HEYOKA_AMD_STEP_KERNEL=$sk timeout 600 python profiles/experiments/single_step.py --kernels 5 2>&1 | tail -1 | sed "s/^/STEP_KERNEL=$sk /"
done > gpurun_out/r05_run13_single_step.log 2>&1
cat gpurun_out/r05_run13_single_step.log
timeout 900 python -m pytest tests -x -q -m gpu -k "raw_step or lockstep or propagate_grid or callback" > gpurun_out/r05_run13_tests.log 2>&1
tail -5 gpurun_out/r05_run13_tests.log
