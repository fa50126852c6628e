#!/bin/bash
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=/root/repo
rm -rf /tmp/evprof
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/evprof -o ev -- python /root/repo/profiles/experiments/events_scale.py --systems 1048576 --steps 4 --skip-lane-stepper > /tmp/ev.log 2>&1
grep -v "^W2026\|^E2026" /tmp/ev.log | tail -8 | cut -c1-300
python - <<'PY'
import sqlite3, glob
for db in glob.glob('/tmp/evprof/**/*.db', recursive=True):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, count(*), sum(end - start), avg(end - start) from kernels group by name order by 3 desc"))
    for r in rows[:14]: print("%-40s %5d %12d %12.0f" % (r[0][:40], r[1], r[2], r[3]))
PY
