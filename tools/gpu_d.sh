#!/bin/bash
# experiment: do MIOpen's NHWC solvers (no NCHW<->NHWC transposes) perform when a real find picks them?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/d/miopen_db
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/d/miopen_db
export PYTORCH_MIOPEN_SUGGEST_NHWC=1
( time MIOPEN_FIND_MODE=1 timeout 1500 python tools/conv_formats.py ) > gpurun_out/d/conv_formats_find1.txt 2>&1
echo "rc=$?"; tail -20 gpurun_out/d/conv_formats_find1.txt
ls -la gpurun_out/d/miopen_db | head
( time MIOPEN_FIND_MODE=1 timeout 600 python tools/conv_formats.py ) > gpurun_out/d/conv_formats_find1_again.txt 2>&1
tail -14 gpurun_out/d/conv_formats_find1_again.txt
