#!/bin/bash
# Round 5, GPU call 22: the whole GPU suite, smoke() and the default bench line on the tree with the lock-step / setter changes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05_run22_pytest.log 2>&1
tail -6 gpurun_out/r05_run22_pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05_run22_smoke.log 2>&1
tail -3 gpurun_out/r05_run22_smoke.log
( time timeout 900 python bench.py ) > gpurun_out/r05_run22_bench_default.log 2>&1
tail -4 gpurun_out/r05_run22_bench_default.log | cut -c1-200
