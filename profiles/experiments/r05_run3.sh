#!/bin/bash
# Round 5, GPU call 3: early chain terms interleaved with the dependent section of a round: parity + A/B.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "(bench_length_parity and v5) or (loop_control and v5) or refill" > gpurun_out/r05_run3_tests.log 2>&1
tail -5 gpurun_out/r05_run3_tests.log
timeout 900 python profiles/experiments/ab.py "HEYOKA_AMD_V5_OPTS=noilv" "HEYOKA_AMD_V5_OPTS=none" "HEYOKA_AMD_V5_ILV=3:10:1" "HEYOKA_AMD_V5_ILV=3:13:0" "HEYOKA_AMD_V5_ILV=3:16:1" "HEYOKA_AMD_V5_ILV=3:18:1" "HEYOKA_AMD_V5_ILV=6:13:1" --dt 40 --rounds 5 > gpurun_out/r05_run3_ab.log 2>&1
cat gpurun_out/r05_run3_ab.log
