#!/bin/bash
# Machine LICM on/off (SGPR pressure: hoisted constants push pointers / masks into v_writelane spills).
cd /root/repo
export PYTHONPATH=/root/repo
for fl in "-mllvm -disable-machine-licm" ""; do
  HEYOKA_AMD_HIPRTC_FLAGS="$fl" timeout 200 python profiles/experiments/exp_variant.py --dt 40 --calls 3 --tag "flags:$fl" 2>&1 | tail -1 | cut -c1-600
done
HEYOKA_AMD_HIPRTC_FLAGS="-mllvm -disable-machine-licm" timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "outer_ss or cluster or loop_control" --timeout=120 2>&1 | tail -3
for fl in "-mllvm -disable-machine-licm" ""; do
HEYOKA_AMD_HIPRTC_FLAGS="$fl" timeout 200 python bench.py --workload two_body --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
done
