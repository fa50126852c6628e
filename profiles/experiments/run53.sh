#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 120 python profiles/experiments/dbg_v5.py 70 2>&1 | tail -12 | tee gpurun_out/r53_dbg.log
timeout 600 python profiles/experiments/ab.py "HEYOKA_AMD_ONE_LANE=0" "" "HEYOKA_AMD_V5_PRIO=0" --dt 60 --rounds 4 2>&1 | tail -3 | tee gpurun_out/r53_ab.log
