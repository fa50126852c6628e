#!/bin/bash
# Stepper with events as a compile-time mode-4 specialisation (no propagation bookkeeping, no stores of unchanged values).
cd "${GRAFT_REPO_ROOT:-.}"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_cpp_api.py -x -q -m gpu -k "event or cpp or reference" 2>&1 | tail -5
timeout 200 python profiles/experiments/events_scale.py --systems 1048576 --skip-lane-stepper --steps 6 2>&1 | tail -2
bash profiles/experiments/run65.sh 2>&1 | tail -14
