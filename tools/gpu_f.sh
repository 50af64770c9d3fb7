#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f
timeout 1200 python -m pytest tests/test_gpu_slab.py tests/test_gpu_ops.py tests/test_dist_gpu.py tests/test_cli_and_interface.py -m gpu -q > gpurun_out/f/tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/f/tests.log
timeout 600 python tools/rank_cost.py 1 2 4 8 > gpurun_out/f/rank_cost.txt 2>&1; echo "rank_cost rc=$?"; grep world gpurun_out/f/rank_cost.txt
timeout 900 python tools/rank_cost.py --4k 8 > gpurun_out/f/rank_cost_4k.txt 2>&1; echo "rank_cost 4k rc=$?"; grep world gpurun_out/f/rank_cost_4k.txt
