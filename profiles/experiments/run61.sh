#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
HEYOKA_AMD_EVENTS_TIMING=1 timeout 300 python profiles/experiments/events_scale.py --systems 1048576 --skip-lane-stepper --steps 4 > gpurun_out/r61_events_scale.log 2>&1
tail -40 gpurun_out/r61_events_scale.log
