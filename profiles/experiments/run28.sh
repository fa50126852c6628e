#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "event" --timeout=300 2>&1 | tail -4
timeout 400 python profiles/experiments/events_scale.py --systems 1048576 --steps 4 --skip-lane-stepper --propagate 20 2>&1 | tail -2 | cut -c1-420
