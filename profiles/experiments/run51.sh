#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
timeout 600 python -m pytest tests -q -m gpu --timeout=600 -k "bit_identical_to_the_oracle" 2>&1 | tail -15
