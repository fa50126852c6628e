#!/bin/bash
# Round 5, GPU call 7: new GPU tests (nbody64 bench-length parity, hazard scan on the box's hiprtc, gather records, events),
# then the default bench line with the new legs.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu -k "nbody64 or nbody or hazard or gather or sharded or events or dropin or terminal" > gpurun_out/r05_run7_tests.log 2>&1
tail -8 gpurun_out/r05_run7_tests.log
timeout 900 python bench.py > gpurun_out/r05_run7_bench.json.log 2> gpurun_out/r05_run7_bench.err
tail -c 6000 gpurun_out/r05_run7_bench.json.log; tail -5 gpurun_out/r05_run7_bench.err
