cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pointwise_scaled.py -q -x 2>&1 | grep -E "^E  |passed|failed|Error" | head -20
timeout 300 python tools/rank_cost.py 1 2>&1 | grep "^world"
SBMC_HIP_PW_WIDE_FUSED=0 timeout 300 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed 's/$/ [WIDE_FUSED=0]/'
timeout 300 python tools/rank_cost.py 1 2>&1 | grep "^world"
