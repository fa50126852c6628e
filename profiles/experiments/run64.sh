#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/r64_tests.log 2>&1
tail -8 gpurun_out/r64_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r64_smoke.log 2>&1
tail -3 gpurun_out/r64_smoke.log
