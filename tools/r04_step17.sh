cd $GRAFT_REPO_ROOT
o=gpurun_out/r04q; mkdir -p $o
timeout 400 python tools/rank_cost.py --rccl-self 8 2>&1 | grep "^world" | tee $o/rank_cost.txt
timeout 400 python tools/rank_cost.py --ipc-self 8 2>&1 | grep "^world" | tee -a $o/rank_cost.txt
timeout 900 python -m pytest tests/test_dist_gpu.py tests/test_gpu_rccl.py tests/test_gpu_wbank.py -q 2>&1 | tail -3
