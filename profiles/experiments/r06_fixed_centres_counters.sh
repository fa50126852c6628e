#!/bin/bash
# Counters of the block-mode kernel (first-generation cluster phase) on model::fixed_centres with 100 masses, 262 144 systems.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
R=$(pwd); OUT=$R/gpurun_out/fc100; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
CMD="python $R/profiles/experiments/model_rates.py --only fixed"
DBS=""
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_WAVES" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_IFETCH" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set -d "$OUT/sq$i" -o sq -- $CMD > "$OUT/sq$i.log" 2>&1
  DBS="$DBS $(find $OUT/sq$i -name '*.db' | head -1)"
done
python $R/profiles/pmc_dump.py "$OUT/r06_fixed_centres100_sq_counters.json" hy_taylor "rocprofv3 --pmc passes, last hy_taylor dispatch of: $CMD" $DBS > "$OUT/dump.log" 2>&1
tail -50 "$OUT/dump.log"
find "$OUT" -name '*.db' -delete
