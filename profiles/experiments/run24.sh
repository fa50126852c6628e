#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
timeout 300 python profiles/experiments/ab.py "HEYOKA_AMD_NO_STATIC_SCHEDULE=1" "X=1" 2>&1 | tail -2
HEYOKA_AMD_NO_STATIC_SCHEDULE=1 timeout 300 python profiles/experiments/events_scale.py --systems 1048576 --steps 4 --skip-lane-stepper 2>&1 | tail -2 | cut -c1-160
timeout 300 python profiles/experiments/events_scale.py --systems 1048576 --steps 4 --skip-lane-stepper 2>&1 | tail -2 | cut -c1-160
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "outer_ss or cluster or loop_control" --timeout=120 2>&1 | tail -3
