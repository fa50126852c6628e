#!/bin/bash
# Regenerates the measurement artifacts kept under profiles/ on a GPU box:
#   bash profiles/collect.sh <round-tag>        (e.g. r01)
# For each workload: the bench line (with cpu_baseline), a `rocprofv3 --kernel-trace --stats` pass and two
# separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (never combined with tracing domains), summarised by
# profiles/summarize_rocprof.py into <tag>_<workload>_kernel_stats.txt / _pmc.json.
set -u
TAG=${1:-r01}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUT=gpurun_out/collect_$TAG
mkdir -p "$OUT"
for WL in outer_ss two_body nbody64; do
  case $WL in
    nbody64) PSYS="";;
    *) PSYS="";;
  esac
  timeout 900 python bench.py --workload $WL > "$OUT/bench_$WL.log" 2>&1
  tail -1 "$OUT/bench_$WL.log" | cut -c1-160
  CMD="python bench.py --workload $WL $PSYS --no-cpu-baseline"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt_$WL" -o kt -- $CMD > "$OUT/kt_$WL.log" 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pf_$WL" -o pf -- $CMD > "$OUT/pf_$WL.log" 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pw_$WL" -o pw -- $CMD > "$OUT/pw_$WL.log" 2>&1
  python profiles/summarize_rocprof.py "$OUT/${TAG}_$WL" "$(find $OUT/kt_$WL -name '*.db' | head -1)" \
      "$(find $OUT/pf_$WL -name '*.db' | head -1)" "$(find $OUT/pw_$WL -name '*.db' | head -1)" "$OUT/pf_$WL.log" "$OUT/kt_$WL.log" \
      > "$OUT/summary_$WL.log" 2>&1
  tail -8 "$OUT/summary_$WL.log"
  find "$OUT" -name '*.db' -size +8M -delete
done
