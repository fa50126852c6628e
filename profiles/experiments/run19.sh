#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "outer_ss or cluster or loop_control" --timeout=120 2>&1 | tail -3
timeout 300 python profiles/experiments/ab.py "HEYOKA_AMD_V3_POW_DIV=1" "X=1" 2>&1 | tail -2
