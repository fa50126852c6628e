#!/bin/bash
# SQ counters of a single-step launch (lock-step sweep of 1 048 576 outer-SS systems) next to the ones of the propagation loop
# (profiles/r05_outer_ss_sq_counters.json): where do the wavefronts of a sweep wait?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd)
OUT=$R/gpurun_out/sq_single_step
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CMD="python $R/profiles/experiments/single_step.py --kernels 5 --steps-last"
DBS=""
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_WAVES" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_IFETCH"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set -d "$OUT/sq$i" -o sq -- $CMD > "$OUT/sq$i.log" 2>&1
  DBS="$DBS $(find $OUT/sq$i -name '*.db' | head -1)"
done
python $R/profiles/pmc_dump.py "$R/gpurun_out/r05_single_step_sq_counters.json" hy_taylor "rocprofv3 --pmc passes, last hy_taylor dispatch (a single-step launch over 1 048 576 systems) of: $CMD" $DBS > "$OUT/dump.log" 2>&1
tail -40 "$OUT/dump.log"
find "$OUT" -name '*.db' -size +8M -delete
