#!/bin/bash
# Reaction fusion on/off: correctness subset + throughput.
cd /root/repo
export PYTHONPATH=/root/repo
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "outer_ss or cluster or loop_control" --timeout=120 2>&1 | tail -3
for f in 1 0; do
  HEYOKA_AMD_V3_FUSE_RX=$f timeout 200 python profiles/experiments/exp_variant.py --dt 40 --calls 3 --tag fuse$f 2>&1 | tail -1
done
