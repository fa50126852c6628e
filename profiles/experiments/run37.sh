#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 120 python profiles/experiments/dbg_v5.py 70 2>&1 | tail -14 | tee gpurun_out/r37_dbg.log
timeout 300 python profiles/experiments/ab.py "" "HEYOKA_AMD_ONE_LANE=1" --dt 40 --rounds 3 2>&1 | tail -3 | tee gpurun_out/r37_ab.log
bash profiles/experiments/sq_quick.sh v5b HEYOKA_AMD_ONE_LANE=1 2>&1 | grep -E "SQ_ACTIVE_INST_VALU|SQ_WAVE_CYCLES|SQ_WAIT|SQ_INSTS_V|duration|GRBM|BANK|IDX"
