cd $GRAFT_REPO_ROOT
o=gpurun_out/r04f; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_conv3x3_half.py tests/test_gpu_conv3x3.py tests/test_gpu_wbank.py -q 2>&1 | tail -6 > $o/tests.txt
cat $o/tests.txt
timeout 400 python tools/rank_cost.py --ipc-self 8 2>&1 | grep "^world" | tee $o/rank_cost.txt
timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | tee -a $o/rank_cost.txt
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_fullsize.py -q 2>&1 | tail -6 | tee $o/tests_cfg.txt
