#!/bin/bash
# Round 3: full GPU test-suite after the tail diet + the 64-lane lane-pair kernel enabled.
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -x -m gpu --timeout=600 2>&1 | tail -8 | tee gpurun_out/r34_tests.log
