#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
timeout 400 python profiles/experiments/events_scale.py --systems 1048576 --steps 2 --skip-lane-stepper --propagate 20 --d2 9.0 2>&1 | tail -2 | cut -c150-460
HEYOKA_AMD_EVENTS_TIMING=1 timeout 400 python profiles/experiments/events_scale.py --systems 1048576 --steps 1 --skip-lane-stepper --propagate 2 --d2 9.0 2>&1 | grep events | tail -12 | cut -c1-100
