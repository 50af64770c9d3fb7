cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_conv3x3.py tests/test_gpu_conv3x3_half.py tests/test_dist_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 400 python tools/rank_cost.py --ipc-self 8 2>&1 | grep "^world"
SBMC_CONV3X3_STREAMK=0 timeout 400 python tools/rank_cost.py --ipc-self 8 2>&1 | grep "^world" | sed "s/$/ [SBMC_CONV3X3_STREAMK=0]/"
