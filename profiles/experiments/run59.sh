#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1200 bash profiles/collect.sh r03 outer_ss two_body nbody64 > gpurun_out/r59_collect.log 2>&1
timeout 300 python profiles/experiments/ab_two_body.py > gpurun_out/r59_ab_two_body.log 2>&1
tail -4 gpurun_out/r59_ab_two_body.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r59_tests.log 2>&1
tail -5 gpurun_out/r59_tests.log
