#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "event" --timeout=300 2>&1 | tail -4
HEYOKA_AMD_EVENTS_TIMING=1 timeout 300 python profiles/experiments/events_scale.py --systems 1048576 --steps 3 --skip-lane-stepper 2>&1 | tail -8 | cut -c1-160
timeout 300 python profiles/experiments/ab.py "X=1" --rounds 3 2>&1 | tail -1
