#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
HEYOKA_AMD_EVENTS_TIMING=1 timeout 300 python profiles/experiments/events_scale.py --systems 1048576 --steps 2 --skip-lane-stepper 2>&1 | tail -22 | cut -c1-160
