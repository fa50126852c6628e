#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
for v in 1 0; do
  if [ $v = 1 ]; then export HEYOKA_AMD_EXACT_POW_DIV=1; else unset HEYOKA_AMD_EXACT_POW_DIV; fi
  timeout 200 python bench.py --workload two_body --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('exact_div=$v', b['value'], b['roofline']['frac'])"
done
unset HEYOKA_AMD_EXACT_POW_DIV
timeout 700 python -m pytest tests/test_gpu_parity.py tests/test_models.py -q -x -m gpu -k "two_body or np1body or contraction or random_systems or unrolled or single_step or golden or models_step" --timeout=300 2>&1 | tail -4
