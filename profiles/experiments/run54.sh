#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --timeout=600 -x 2>&1 | tail -6 | tee gpurun_out/r54_tests.log
