#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python profiles/experiments/sensitivity.py > gpurun_out/r05_sensitivity.log 2>&1
cat gpurun_out/r05_sensitivity.log
