#!/bin/bash
# Round 5, GPU call 26: the whole GPU suite and smoke() on the final tree.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05_run26_pytest.log 2>&1
tail -4 gpurun_out/r05_run26_pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05_run26_smoke.log 2>&1
tail -2 gpurun_out/r05_run26_smoke.log
