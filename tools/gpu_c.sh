#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_cli_and_interface.py tests/test_gpu_slab.py tests/test_gpu_backbone.py tests/test_dist_gpu.py -m gpu -q --durations=8 > gpurun_out/c/tests.log 2>&1; echo "tests rc=$?"
tail -25 gpurun_out/c/tests.log
timeout 600 python tools/rank_cost.py 1 8 > gpurun_out/c/rank_cost.txt 2>&1; echo "rank_cost rc=$?"; grep world gpurun_out/c/rank_cost.txt
