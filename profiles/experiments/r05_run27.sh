#!/bin/bash
# Round 5, GPU call 27: equal / repeated masses with the accelerations flattened in the internal program (kernel reached and
# rate, against the planner without the pass); then the whole GPU suite.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python profiles/experiments/linearised.py 1048576 > gpurun_out/r05_run27_linearised.log 2>&1
cat gpurun_out/r05_run27_linearised.log | cut -c1-260
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05_run27_pytest.log 2>&1
tail -6 gpurun_out/r05_run27_pytest.log
