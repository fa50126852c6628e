#!/bin/bash
# Round 5, GPU call 23: bookkeeping block inside the slab + laundered system index in the retire block (A/B).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python profiles/experiments/ab.py "HEYOKA_AMD_V5_OPTS=nobkslab+nolaunder" "HEYOKA_AMD_V5_OPTS=none" "HEYOKA_AMD_V5_OPTS=nolaunder" --dt 40 --rounds 5 > gpurun_out/r05_run23_ab.log 2>&1
cat gpurun_out/r05_run23_ab.log
timeout 600 python -m pytest tests -x -q -m gpu -k "refill or bench_length_parity" > gpurun_out/r05_run23_tests.log 2>&1
tail -3 gpurun_out/r05_run23_tests.log
