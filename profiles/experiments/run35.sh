#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 120 python profiles/experiments/dbg_v5.py 70 2>&1 | tail -20 | tee gpurun_out/r35_dbg.log
timeout 300 python profiles/experiments/ab.py "" "HEYOKA_AMD_ONE_LANE=1" --dt 40 --rounds 3 2>&1 | tail -3 | tee gpurun_out/r35_ab.log
