#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_capi.py -q -x -m gpu -k "event or step or loop_control" --timeout=300 2>&1 | tail -4
timeout 300 python profiles/experiments/events_scale.py --systems 1048576 --steps 5 --skip-lane-stepper 2>&1 | tail -2 | cut -c1-160
timeout 300 python profiles/experiments/events_scale.py --systems 262144 --steps 5 2>&1 | tail -3 | cut -c1-160
