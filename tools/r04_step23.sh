cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_backbone.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python tools/fuzz_pointwise.py --seconds 40 2>&1 | tail -1 | cut -c1-300
SBMC_HIP_PW_GWS=2 timeout 400 python tools/bench_pointwise.py --notest --bwd 2>&1 | tail -3
timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world"
