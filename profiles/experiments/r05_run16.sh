#!/bin/bash
# Round 5, GPU call 16: kernel trace of lock-step steps with two non-terminal + one terminal event (the terminal variant of the
# events leg reads 23 ms in its stepper phase against 4 ms without the terminal event: which kernel is it?).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05_evtrace_t -o ev -- python $R/profiles/experiments/events_fire.py 1048576 terminal > $R/gpurun_out/r05_run16_events.log 2>&1
grep "s per step" $R/gpurun_out/r05_run16_events.log | cut -c1-500
python - <<PY
import sqlite3, glob
db = glob.glob("$R/gpurun_out/r05_evtrace_t/**/*.db", recursive=True)[0]
c = sqlite3.connect(db).cursor()
for r in c.execute("select name, count(*), avg(end-start)/1e6, min(end-start)/1e6, max(end-start)/1e6 from kernels group by name order by 3 desc"):
    print(r)
print([round((e - s) / 1e6, 2) for s, e in c.execute("select start, end from kernels where name = 'hy_taylor' order by start")])
PY
find $R/gpurun_out/r05_evtrace_t -name '*.db' -size +8M -delete
