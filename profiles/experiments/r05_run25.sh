#!/bin/bash
# Round 5, GPU call 25: the long-horizon protocol through one propagate_grid() call (lock-step sweeps, coefficients on demand).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python profiles/experiments/grid_long_horizon.py 1048576 10000 > gpurun_out/r05_run25_grid.log 2>&1
tail -2 gpurun_out/r05_run25_grid.log
