#!/bin/bash
# Round 5, GPU call 12: reactions fused into the acceleration sums of the one-lane-per-pair kernel (A/B against the exported
# reactions, coefficients in registers vs read from LDS at every use), parity tests of that kernel, raw stepper ABI test.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python profiles/experiments/ab.py "HEYOKA_AMD_V5_OPTS=nofrx" "HEYOKA_AMD_V5_OPTS=none" "HEYOKA_AMD_V5_OPTS=frxlds" --dt 40 --rounds 4 > gpurun_out/r05_run12_ab.log 2>&1
cat gpurun_out/r05_run12_ab.log
timeout 1200 python -m pytest tests -x -q -m gpu -k "raw_step or reference_counted or bench_length_parity or refill or outer_ss or contraction or single_step" > gpurun_out/r05_run12_tests.log 2>&1
tail -5 gpurun_out/r05_run12_tests.log
