#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --timeout=600 -k "compact or table" 2>&1 | tail -15 | tee gpurun_out/r50_tests.log
