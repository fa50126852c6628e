#!/bin/bash
# Round 5, GPU call 29: issue-priority scheme of the one-lane-per-pair kernel re-checked on the final kernel (A/B).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python profiles/experiments/ab.py "HEYOKA_AMD_V5_PRIO=2" "HEYOKA_AMD_V5_PRIO=0" "HEYOKA_AMD_V5_PRIO=1" "HEYOKA_AMD_V5_PRIO=3" --dt 40 --rounds 4 > gpurun_out/r05_run29_ab.log 2>&1
cat gpurun_out/r05_run29_ab.log
