cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_conv3x3.py tests/test_gpu_conv3x3_half.py tests/test_gpu_wbank.py -x -q -m gpu 2>&1 | tail -5
for lib in "" "$GRAFT_REPO_ROOT/.ab/libsbmc_abl0.so"; do
  SBMC_HIP_LIB=$lib timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed "s#\$# [${lib:-current}]#"
  SBMC_HIP_LIB=$lib timeout 400 python tools/rank_cost.py --ipc-self 8 2>&1 | grep "^world" | sed "s#\$# [${lib:-current}]#"
done
timeout 900 python tools/fuzz_conv3x3.py --cases 150 2>&1 | tail -1
