#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
timeout 300 python profiles/experiments/ab.py "X=1" "HEYOKA_AMD_V3_RECIP_DIV=1" 2>&1 | tail -2
HEYOKA_AMD_V3_RECIP_DIV=1 timeout 400 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "outer_ss or cluster or contraction or loop_control" --timeout=200 2>&1 | tail -12
