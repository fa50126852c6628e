#!/bin/bash
# Round 3, first GPU call: tail diet A/B, parity tests of the cluster paths, diagnosis of the 64-lane v3 hang.
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 300 python profiles/experiments/ab.py "" "HEYOKA_AMD_NO_PACKED_TAIL=1" --dt 40 --rounds 3 2>&1 | tail -3 | tee gpurun_out/r33_ab.log
for st in step bounded all; do
  echo "== nbody7 $st"; timeout 60 python profiles/experiments/dbg_v3_64.py 7 $st 2>&1 | tail -8
  echo "rc=$?"
done 2>&1 | tee gpurun_out/r33_dbg.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_models.py -q -x -m gpu -k "outer_ss or contraction or cluster or propagate or nonfinite or time_dependent or full_size or reference_batch" --timeout=400 2>&1 | tail -6 | tee gpurun_out/r33_tests.log
