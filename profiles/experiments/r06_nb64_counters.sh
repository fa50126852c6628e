#!/bin/bash
# SQ counters of the N = 64 block kernel under variants of HEYOKA_AMD_BLOCK_OPTS (round 6): usage r06_nb64_counters.sh tag opts [tag opts ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
OUT=$R/gpurun_out/nb64_counters
mkdir -p "$OUT"
cd /tmp
while [ $# -ge 2 ]; do
  TAG=$1; export HEYOKA_AMD_BLOCK_OPTS=$2; shift 2
  CMD="python $R/bench.py --workload nbody64 --no-cpu-baseline --no-extra-workloads --steps 3 --warmup 1"
  DBS=""
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set -d "$OUT/${TAG}_sq$i" -o sq -- $CMD > "$OUT/${TAG}_sq$i.log" 2>&1
    DBS="$DBS $(find $OUT/${TAG}_sq$i -name '*.db' | head -1)"
  done
  python $R/profiles/pmc_dump.py "$OUT/r06_nb64_${TAG}_sq_counters.json" hy_taylor "HEYOKA_AMD_BLOCK_OPTS=$HEYOKA_AMD_BLOCK_OPTS: $CMD" $DBS > "$OUT/${TAG}_dump.log" 2>&1
  tail -30 "$OUT/${TAG}_dump.log"
  find "$OUT" -name '*.db' -delete
done
