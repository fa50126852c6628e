#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python profiles/experiments/ab.py "HEYOKA_AMD_V5_OPTS=none" "HEYOKA_AMD_V5_OPTS=spread" "HEYOKA_AMD_V5_OPTS=spread,HEYOKA_AMD_V5_PRIO=1" "HEYOKA_AMD_V5_OPTS=spread,HEYOKA_AMD_V5_PRIO=0" --dt 40 --rounds 5 > gpurun_out/r05_run6_ab.log 2>&1
cat gpurun_out/r05_run6_ab.log
