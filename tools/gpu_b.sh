#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b
timeout 600 python tools/rank_cost.py 1 2 4 8 > gpurun_out/b/rank_cost.txt 2>&1; echo "rank_cost rc=$?"; grep world gpurun_out/b/rank_cost.txt
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_backbone.py tests/test_gpu_ops.py tests/test_gpu_slab.py -q --durations=15 > gpurun_out/b/new_tests.log 2>&1; echo "new tests rc=$?"
tail -40 gpurun_out/b/new_tests.log
