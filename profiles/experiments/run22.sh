#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_capi.py tests/test_cpp_api.py -q -x -m gpu -k "event" --timeout=300 2>&1 | tail -8
HEYOKA_AMD_EVENTS_HOST_LOGIC=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "events_batch or events_on_the_cluster" --timeout=300 2>&1 | tail -3
timeout 300 python profiles/experiments/events_scale.py --systems 262144 --steps 4 --skip-lane-stepper 2>&1 | tail -3
timeout 300 python profiles/experiments/events_scale.py --systems 1048576 --steps 4 --skip-lane-stepper 2>&1 | tail -3
