#!/bin/bash
# Prefetch of the next group under the static schedule (parked in the slab): correctness subset, headline A/B, single-step rate,
# step with events.
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "loop_control or lockstep or cluster_stepper or compact_taylor or propagate_grid or callback or raw_step or refill or 17_to_32 or bench_length" 2>&1 | tail -5
timeout 300 python profiles/experiments/ab.py "HEYOKA_AMD_NO_PREFETCH=1" "HEYOKA_AMD_X=1" --rounds 3 2>&1 | tail -2
for v in 1 0; do
  echo "NO_PREFETCH=$v"
  if [ $v = 1 ]; then export HEYOKA_AMD_NO_PREFETCH=1; else unset HEYOKA_AMD_NO_PREFETCH; fi
  timeout 200 python profiles/experiments/single_step.py 2>&1 | tail -2
  timeout 200 python profiles/experiments/events_scale.py --systems 1048576 --skip-lane-stepper --steps 6 2>&1 | tail -2
done
