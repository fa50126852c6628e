#!/bin/bash
cd /root/repo
export PYTHONPATH=/root/repo
timeout 300 python profiles/experiments/ab.py "HEYOKA_AMD_NO_NMAX=1,HEYOKA_AMD_RHO_2EXP=1" "HEYOKA_AMD_NO_NMAX=1" "HEYOKA_AMD_RHO_2EXP=1" "X=1" 2>&1 | tail -4
