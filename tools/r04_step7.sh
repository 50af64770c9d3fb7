cd $GRAFT_REPO_ROOT
o=gpurun_out/r04g; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_conv3x3.py tests/test_gpu_conv3x3_half.py tests/test_gpu_wbank.py -q 2>&1 | tail -8 > $o/tests.txt
cat $o/tests.txt
for sk in 1 0; do
SBMC_CONV3X3_STREAMK=$sk timeout 400 python tools/rank_cost.py --ipc-self 8 2>&1 | grep "^world" | sed "s/$/ [STREAMK=$sk]/" | tee -a $o/rank_cost.txt
SBMC_CONV3X3_STREAMK=$sk timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed "s/$/ [STREAMK=$sk]/" | tee -a $o/rank_cost.txt
done
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_backbone.py -q 2>&1 | tail -6 | tee $o/tests_cfg.txt
