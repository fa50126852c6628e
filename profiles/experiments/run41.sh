#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
O=HEYOKA_AMD_ONE_LANE=1
timeout 600 python profiles/experiments/ab.py "$O,HEYOKA_AMD_V5_NO_EARLY=1" "$O,HEYOKA_AMD_V5_NO_EARLY=1,HEYOKA_AMD_V5_GLUE_LAST=1" "$O,HEYOKA_AMD_V5_EARLY_KMAX=8" "$O,HEYOKA_AMD_V5_EARLY_KMAX=12" "$O,HEYOKA_AMD_V5_EARLY_KMAX=14" "$O,HEYOKA_AMD_V5_EARLY_KMAX=16" --dt 40 --rounds 3 2>&1 | tail -6 | tee gpurun_out/r41_ab.log
timeout 120 python profiles/experiments/dbg_v5.py 70 2>&1 | tail -5 | tee gpurun_out/r41_dbg.log
