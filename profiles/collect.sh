#!/bin/bash
# Regenerates the measurement artifacts kept under profiles/ on a GPU box:
#   bash profiles/collect.sh <round-tag> [workloads...]        (e.g. r02 outer_ss two_body nbody64)
# For each workload: the bench line (with cpu_baseline), a `rocprofv3 --kernel-trace --stats` pass, two separate
# `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes and one pass of SQ counters (never combined with tracing domains),
# summarised by profiles/summarize_rocprof.py / pmc_dump.py into <tag>_<workload>_kernel_stats.txt / _pmc.json /
# _sq_counters.json. First: the FETCH/WRITE_SIZE calibration on known-byte 8 B/lane streams (profiles/ubench/stream8.hip).
set -u
TAG=${1:-r02}
shift || true
WLS=${@:-outer_ss two_body nbody64 outer_ss_forced_table nbody6_j2_mixed sine_lattice16_mixed}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$(pwd)
export TMPDIR=/tmp
OUT=$R/gpurun_out/collect_$TAG
mkdir -p "$OUT"
cd /tmp
if [ -x $R/profiles/ubench/stream8.bin ]; then
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/cal_f" -o c -- $R/profiles/ubench/stream8.bin > "$OUT/cal_f.log" 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/cal_w" -o c -- $R/profiles/ubench/stream8.bin > "$OUT/cal_w.log" 2>&1
  python $R/profiles/calibrate_pmc.py "$OUT/${TAG}_pmc_calibration.json" "$(find $OUT/cal_f -name '*.db' | head -1)" "$(find $OUT/cal_w -name '*.db' | head -1)" > "$OUT/cal.log" 2>&1
  tail -40 "$OUT/cal.log"
fi
for WL in $WLS; do
  timeout 900 python $R/bench.py --workload $WL --steps 20 --warmup 5 --no-extra-workloads > "$OUT/bench_$WL.log" 2>&1
  tail -1 "$OUT/bench_$WL.log" | cut -c1-200
  CMD="python $R/bench.py --workload $WL --no-cpu-baseline --no-extra-workloads --steps 4 --warmup 1"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt_$WL" -o kt -- $CMD > "$OUT/kt_$WL.log" 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pf_$WL" -o pf -- $CMD > "$OUT/pf_$WL.log" 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pw_$WL" -o pw -- $CMD > "$OUT/pw_$WL.log" 2>&1
  python $R/profiles/summarize_rocprof.py "$OUT/${TAG}_$WL" "$(find $OUT/kt_$WL -name '*.db' | head -1)" \
      "$(find $OUT/pf_$WL -name '*.db' | head -1)" "$(find $OUT/pw_$WL -name '*.db' | head -1)" "$OUT/pf_$WL.log" "$OUT/kt_$WL.log" \
      > "$OUT/summary_$WL.log" 2>&1
  tail -8 "$OUT/summary_$WL.log"
  DBS=""
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_WAVES" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_IFETCH" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $set -d "$OUT/sq${i}_$WL" -o sq -- $CMD > "$OUT/sq${i}_$WL.log" 2>&1
    DBS="$DBS $(find $OUT/sq${i}_$WL -name '*.db' | head -1)"
  done
  python $R/profiles/pmc_dump.py "$OUT/${TAG}_${WL}_sq_counters.json" hy_taylor "rocprofv3 --pmc passes (SQ counters in quad-cycles summed over the waves; GRBM_GUI_ACTIVE summed over the 8 XCDs), last hy_taylor dispatch of: $CMD" $DBS > "$OUT/sq_$WL.log" 2>&1
  tail -45 "$OUT/sq_$WL.log"
  find "$OUT" -name '*.db' -delete
done
