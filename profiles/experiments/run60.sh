#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r60_tests.log 2>&1
tail -8 gpurun_out/r60_tests.log
timeout 300 python profiles/experiments/events_scale.py > gpurun_out/r60_events_scale.log 2>&1
tail -12 gpurun_out/r60_events_scale.log
