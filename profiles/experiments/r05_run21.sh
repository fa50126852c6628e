#!/bin/bash
# Round 5, GPU call 21: propagate_grid() with the Taylor coefficients on demand (no callback) against one of every step
# (trivial callback); grid / dense-output tests.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python profiles/experiments/grid_on_demand.py 262144 2000 > gpurun_out/r05_run21_grid.log 2>&1
tail -4 gpurun_out/r05_run21_grid.log




