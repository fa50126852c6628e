#!/bin/bash
# Round 5, GPU call 17: per-phase laps of every step of the events leg (non-terminal and terminal variants).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python profiles/experiments/events_leg_laps.py 1048576 12 > gpurun_out/r05_run17_laps.log 2>&1
grep -c "stepper" gpurun_out/r05_run17_laps.log
grep "stepper" gpurun_out/r05_run17_laps.log | awk '{print $(NF-1)}' | tr '\n' ' '
echo
grep "dout + post" gpurun_out/r05_run17_laps.log | awk '{print $(NF-1)}' | tr '\n' ' '
echo
tail -2 gpurun_out/r05_run17_laps.log | cut -c1-1500
