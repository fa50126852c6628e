#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_capi.py tests/test_cpp_api.py -q -x -m gpu -k "event or grid" --timeout=300 2>&1 | tail -12
