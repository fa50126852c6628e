#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
timeout 300 python profiles/experiments/ab.py "HEYOKA_AMD_V3_FUSE_RX=0" "HEYOKA_AMD_V3_FUSE_RX=1" "HEYOKA_AMD_HIPRTC_FLAGS=-mllvm -disable-machine-licm" "HEYOKA_AMD_V3_FUSE_RX=0,HEYOKA_AMD_HIPRTC_FLAGS=-mllvm -disable-machine-licm" 2>&1 | tail -5
