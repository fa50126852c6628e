#!/bin/bash
# Kernel trace + the two PMC passes of one workload (the part of profiles/collect.sh that bench.py's traffic lookup needs).
set -u
WL=${1:-nbody64}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd)
export TMPDIR=/tmp
OUT=$R/gpurun_out/mini_$WL
mkdir -p "$OUT"
cd /tmp
CMD="python $R/bench.py --workload $WL --no-cpu-baseline --steps 3 --warmup 1"
timeout 100 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- $CMD > "$OUT/kt.log" 2>&1
timeout 100 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pf" -o pf -- $CMD > "$OUT/pf.log" 2>&1
timeout 100 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pw" -o pw -- $CMD > "$OUT/pw.log" 2>&1
python $R/profiles/summarize_rocprof.py "$OUT/r02_$WL" "$(find $OUT/kt -name '*.db' | head -1)" "$(find $OUT/pf -name '*.db' | head -1)" \
    "$(find $OUT/pw -name '*.db' | head -1)" "$OUT/pf.log" "$OUT/kt.log" > "$OUT/summary.log" 2>&1
tail -3 "$OUT/summary.log" | cut -c1-200
find "$OUT" -name '*.db' -size +8M -delete
