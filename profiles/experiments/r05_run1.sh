#!/bin/bash
# Round 5, GPU call 1: parity of the v5 stepper after the instruction diet + interleaved A/B of its three items.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "(bench_length_parity and v5) or (loop_control and v5) or refill or outer_ss_step_selector or full_size_invariants" > gpurun_out/r05_run1_tests.log 2>&1
tail -5 gpurun_out/r05_run1_tests.log
B="HEYOKA_AMD_V5_OPTS"
timeout 600 python profiles/experiments/ab.py "$B=nomsq+nopack2+nosc" "$B=nopack2+nosc" "$B=nomsq+nosc" "$B=nomsq+nopack2" "$B=none" --dt 40 --rounds 5 > gpurun_out/r05_run1_ab.log 2>&1
cat gpurun_out/r05_run1_ab.log
