#!/bin/bash
# Round 5, GPU call 11: raw stepper ABI test with the event jets of every system; timing experiment for fusing the reaction
# products into the acceleration sums: "norx" drops the three reaction stores and products of an order (WRONG results: the
# upper bound of the gain), the sixth pad field keeps five more per-lane doubles live through the orders and uses each in
# one dependent FMA per order (what the per-lane coefficients of the sums would cost).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -x -q -m gpu -k "raw_step or reference_counted" > gpurun_out/r05_run11_tests.log 2>&1
tail -5 gpurun_out/r05_run11_tests.log
timeout 900 python profiles/experiments/ab.py "HEYOKA_AMD_V5_OPTS=none" "HEYOKA_AMD_V5_OPTS=norx" "HEYOKA_AMD_V5_OPTS=norx,HEYOKA_AMD_V5_PAD=0:0:0:0:0:5" "HEYOKA_AMD_V5_OPTS=none,HEYOKA_AMD_V5_PAD=0:0:0:0:0:5" --dt 40 --rounds 4 > gpurun_out/r05_run11_ab.log 2>&1
cat gpurun_out/r05_run11_ab.log
