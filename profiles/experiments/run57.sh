#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 300 python bench.py --workload two_body --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('two_body', d['value'], d['roofline']['frac'], d['roofline']['kernel_resources'])"
timeout 1500 python -m pytest tests -q -m gpu --timeout=600 -k "pendulum or two_body or golden or unrolled or tutorial or cr3bp or bit_identical or kepE or unary or atan2 or piecewise or events or node_jets or random_systems or models" 2>&1 | tail -8 | tee gpurun_out/r57_tests.log
