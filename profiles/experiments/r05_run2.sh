#!/bin/bash
# Round 5, GPU call 2: wide-read slab layout (ds_read_b128) of the v5 stepper: parity + A/B.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "(bench_length_parity and v5) or (loop_control and v5) or refill or outer_ss_step_selector" > gpurun_out/r05_run2_tests.log 2>&1
tail -5 gpurun_out/r05_run2_tests.log
B="HEYOKA_AMD_V5_OPTS"
timeout 600 python profiles/experiments/ab.py "$B=nowide" "$B=none" "$B=nowide+nomsq+nopack2+nosc" "HEYOKA_AMD_V5_PRIO=1" "HEYOKA_AMD_V5_GLUE_LAST=1" --dt 40 --rounds 5 > gpurun_out/r05_run2_ab.log 2>&1
cat gpurun_out/r05_run2_ab.log
