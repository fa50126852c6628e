#!/bin/bash
# Round 5, GPU call 19: events leg after the pinned landing buffer of the records and the allocation-free host loop; event tests.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python profiles/experiments/events_leg_laps.py 1048576 6 > gpurun_out/r05_run19_laps.log 2>&1
grep "'value'" gpurun_out/r05_run19_laps.log | cut -c1-1800
timeout 1200 python -m pytest tests -x -q -m gpu -k "event or reference_cases or dropin or raw_step" > gpurun_out/r05_run19_tests.log 2>&1
tail -4 gpurun_out/r05_run19_tests.log
