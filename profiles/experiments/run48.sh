#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --timeout=600 -k "ensemble or distributed or sharded or cpp or device_array" 2>&1 | tail -8 | tee gpurun_out/r48_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r48_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r48_bench.json'))
print(d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['hw_threads'], d['cpu_baseline']['per_core'])
for e in d['extra_workloads']:
    print(e['config']['workload'][:10], e.get('value'), e['roofline']['bound'], e['roofline']['frac'], e.get('cpu_baseline',{}).get('value'), e.get('cpu_baseline',{}).get('per_core'))
PY
