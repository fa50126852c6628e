#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/r62_tests.log 2>&1
tail -8 gpurun_out/r62_tests.log
