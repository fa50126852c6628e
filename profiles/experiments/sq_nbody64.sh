#!/bin/bash
# Counters of the N = 64 block stepper for one variant: sq_nbody64.sh <tag> [ENV=VAL ...]
set -u
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd)
export TMPDIR=/tmp PYTHONPATH=$R
OUT=$R/gpurun_out/sq64_$TAG
mkdir -p "$OUT"
cd /tmp
CMD="python $R/bench.py --workload nbody64 --no-cpu-baseline --no-extra-workloads --steps 2 --warmup 1"
timeout 300 $CMD > "$OUT/bench.log" 2>&1
python - "$OUT/bench.log" <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench value %.4g  kernel_ms %.1f  per_launch_steps %.4g" % (d["value"], d["roofline"]["kernel_ms_avg"], d["config"]["system_steps_per_launch"]))
P
DBS=""
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_WAVES" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_IFETCH" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d "$OUT/sq${i}" -o sq -- $CMD > "$OUT/sq${i}.log" 2>&1
  DBS="$DBS $(find $OUT/sq${i} -name '*.db' | head -1)"
done
python $R/profiles/pmc_dump.py "$R/gpurun_out/sq64_${TAG}.json" hy_taylor "counters, N = 64 block stepper, variant: $*" $DBS > "$OUT/dump.log" 2>&1
tail -60 "$OUT/dump.log"
find "$OUT" -name '*.db' -delete
