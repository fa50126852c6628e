#!/bin/bash
# Round 5, GPU call 9: store-conflict-free placement of the operand arrays (A/B), raw stepper ABI test, module unloading test,
# two-body one vs two wavefronts per SIMD.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "raw_step or reference_counted or (bench_length_parity and v5) or refill" > gpurun_out/r05_run9_tests.log 2>&1
tail -12 gpurun_out/r05_run9_tests.log
timeout 900 python profiles/experiments/ab.py "HEYOKA_AMD_V5_OPTS=nostoreplace" "HEYOKA_AMD_V5_OPTS=none" --dt 40 --rounds 5 > gpurun_out/r05_run9_ab.log 2>&1
cat gpurun_out/r05_run9_ab.log
timeout 600 python profiles/experiments/ab_two_body_waves.py > gpurun_out/r05_run9_two_body_waves.log 2>&1
cat gpurun_out/r05_run9_two_body_waves.log
