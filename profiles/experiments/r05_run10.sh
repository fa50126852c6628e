#!/bin/bash
# Round 5, GPU call 10: the whole GPU suite on the tree with store placement, pruned switches and reference-counted code objects.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_run10_pytest.log 2>&1
tail -15 gpurun_out/r05_run10_pytest.log
