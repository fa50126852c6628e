#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
timeout 1500 python -m pytest tests -q -m gpu --timeout=600 -k "contraction_off or bit_identical" 2>&1 | tail -6
