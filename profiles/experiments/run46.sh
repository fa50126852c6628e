#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --timeout=600 -k "lockstep or step_callback or propagate_grid or cpp or callback or events or c_output or continuous" 2>&1 | tail -15 | tee gpurun_out/r46_tests.log
