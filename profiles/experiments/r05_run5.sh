#!/bin/bash
# Round 5, GPU call 5: velocity exchange (8 instead of 10 LDS stores per round): parity + A/B.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "(bench_length_parity and v5) or (loop_control and v5) or refill or outer_ss_step_selector or full_size" > gpurun_out/r05_run5_tests.log 2>&1
tail -5 gpurun_out/r05_run5_tests.log
timeout 900 python profiles/experiments/ab.py "HEYOKA_AMD_V5_OPTS=novx" "HEYOKA_AMD_V5_OPTS=none" "HEYOKA_AMD_V5_OPTS=novx+nowide" "HEYOKA_AMD_V5_OPTS=nobanksearch" --dt 40 --rounds 5 > gpurun_out/r05_run5_ab.log 2>&1
cat gpurun_out/r05_run5_ab.log
