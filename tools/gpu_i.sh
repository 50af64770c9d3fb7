#!/bin/bash
# MIOpen NHWC experiment, properly: steady-state time of every U-net convolution (fwd + bwd) with
# channels_last tensors when a real find has chosen the solver, vs NCHW under the FAST heuristics
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/i/db
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/i/db
( time MIOPEN_FIND_MODE=1 PYTORCH_MIOPEN_SUGGEST_NHWC=1 timeout 1200 python tools/conv_formats.py --only-channels-last ) > gpurun_out/i/nhwc_find1.txt 2>&1
grep -v MIOpen gpurun_out/i/nhwc_find1.txt | tail -14
( time MIOPEN_FIND_MODE=2 timeout 600 python tools/conv_formats.py ) > gpurun_out/i/fast.txt 2>&1
grep -v MIOpen gpurun_out/i/fast.txt | tail -14
# does a HYBRID-mode process pick the find results up from the user db without searching again?
( time MIOPEN_FIND_MODE=3 PYTORCH_MIOPEN_SUGGEST_NHWC=1 timeout 900 python tools/conv_formats.py --only-channels-last ) > gpurun_out/i/nhwc_hybrid_reuse.txt 2>&1
grep -v MIOpen gpurun_out/i/nhwc_hybrid_reuse.txt | tail -14
ls -la gpurun_out/i/db
