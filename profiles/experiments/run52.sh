#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 600 python profiles/experiments/ab.py "HEYOKA_AMD_V5_PRIO=0" "" "HEYOKA_AMD_V5_PRIO=2" --dt 60 --rounds 4 2>&1 | tail -3 | tee gpurun_out/r52_ab.log
