#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu --timeout=600 -k "random_systems or models_step" 2>&1 | tail -30 | tee gpurun_out/r56_tests.log
