#!/bin/bash
# Round 5, GPU call 30: events on a system with default masses (stepper with events through the flattened accelerations);
# every test with events.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "event or reference_event" > gpurun_out/r05_run30_tests.log 2>&1
tail -5 gpurun_out/r05_run30_tests.log
