#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -x -m gpu --timeout=600 2>&1 | tail -8 | tee gpurun_out/r43_tests.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r43_bench.log
