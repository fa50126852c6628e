#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "events or event_equations" --timeout=300 2>&1 | tail -8
timeout 300 python profiles/experiments/events_scale.py --systems 262144 --steps 4 2>&1 | tail -4
