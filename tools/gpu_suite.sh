#!/bin/bash
# The whole GPU suite + smoke() on the GPU box:   tools/grun --timeout 3000 bash tools/gpu_suite.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/suite
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > gpurun_out/suite/gpu_suite.txt 2>&1; tail -5 gpurun_out/suite/gpu_suite.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
