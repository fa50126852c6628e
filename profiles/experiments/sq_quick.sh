#!/bin/bash
# SQ counters of the outer-SS stepper for one variant: sq_quick.sh <tag> [ENV=VAL ...]
set -u
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd)
export TMPDIR=/tmp PYTHONPATH=$R
OUT=$R/gpurun_out/sq_$TAG
mkdir -p "$OUT"
cd /tmp
CMD="python $R/bench.py --workload outer_ss --no-cpu-baseline --steps 2 --warmup 1"
DBS=""
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_WAVES" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_IFETCH"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d "$OUT/sq${i}" -o sq -- $CMD > "$OUT/sq${i}.log" 2>&1
  DBS="$DBS $(find $OUT/sq${i} -name '*.db' | head -1)"
done
python $R/profiles/pmc_dump.py "$R/gpurun_out/sq_${TAG}.json" hy_taylor "SQ counters, variant: $*" $DBS > "$OUT/dump.log" 2>&1
tail -40 "$OUT/dump.log"
find "$OUT" -name '*.db' -delete
