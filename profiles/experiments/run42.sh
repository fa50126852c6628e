#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
mkdir -p gpurun_out
O=HEYOKA_AMD_ONE_LANE=1,HEYOKA_AMD_V5_NO_EARLY=1
timeout 600 python profiles/experiments/ab.py "$O" "$O,HEYOKA_AMD_HIPRTC_FLAGS=-mllvm -amdgpu-sched-strategy=iterative-ilp" "$O,HEYOKA_AMD_HIPRTC_FLAGS=-mllvm -amdgpu-sched-strategy=max-ilp" "$O,HEYOKA_AMD_HIPRTC_FLAGS=-mllvm -amdgpu-sched-strategy=iterative-minreg" --dt 40 --rounds 3 2>&1 | tail -4 | tee gpurun_out/r42_ab.log
