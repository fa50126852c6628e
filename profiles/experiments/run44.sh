#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_cpp_api.py -q -x -m gpu --timeout=600 -k "semantics or nonfinite or loop_control or bench_length or propagate or lockstep or ensemble or cpp" 2>&1 | tail -8 | tee gpurun_out/r44_tests.log
timeout 300 python profiles/experiments/ab.py "" "HEYOKA_AMD_ONE_LANE=0" --dt 60 --rounds 3 2>&1 | tail -2 | tee gpurun_out/r44_ab.log
