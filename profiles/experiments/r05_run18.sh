#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
HEYOKA_AMD_EVENTS_TIMING_DBG=1 timeout 600 python profiles/experiments/events_leg_laps.py 1048576 6 > gpurun_out/r05_run18_laps.log 2>&1
grep "stepper-dbg" gpurun_out/r05_run18_laps.log | tail -64 | awk '{printf "%s=%s ", $2 $3, $(NF-1)} NR%4==0 {print ""}' | tail -12
