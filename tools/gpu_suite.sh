#!/bin/bash
# The whole GPU suite + smoke() on the GPU box:   tools/grun --timeout 3000 bash tools/gpu_suite.sh [-x]
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/suite
( time timeout 2700 python -m pytest tests -q -m gpu -rf "$@" ) > gpurun_out/suite/gpu_suite.txt 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/suite/gpu_suite.txt | tail -40
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
