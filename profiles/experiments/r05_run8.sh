#!/bin/bash
# Round 5, GPU call 8: raw stepper ABI test, kernel trace of steps with firing events, the new long-horizon leg.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "raw_step" > gpurun_out/r05_run8_tests.log 2>&1
tail -12 gpurun_out/r05_run8_tests.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05_evtrace -o ev -- python $R/profiles/experiments/events_fire.py 1048576 > $R/gpurun_out/r05_run8_events.log 2>&1
tail -3 $R/gpurun_out/r05_run8_events.log
find $R/gpurun_out/r05_evtrace -name '*kernel_stats.csv' | head -1 | xargs cat | cut -c1-200 | head -14
cd $R
timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r05_run8_bench.json.log 2> gpurun_out/r05_run8_bench.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r05_run8_bench.json.log').read().split('\n') if x.startswith('{')][-1]
d=json.loads(l)
print(d['value'], d['roofline']['frac'])
for e in d['extra_workloads'][-2:]:
    print({k:v for k,v in e.items() if k not in ('config','roofline')})
PY
find gpurun_out/r05_evtrace -name '*.db' -size +8M -delete
