cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_backbone.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python tools/fuzz_pointwise.py --seconds 60 2>&1 | tail -1 | cut -c1-300
for m in 1 2; do echo "SBMC_HIP_PW_GWS=$m"; SBMC_HIP_PW_GWS=$m timeout 400 python tools/bench_pointwise.py --time --bwd 2>&1 | tail -12; done
for m in 1 2; do SBMC_HIP_PW_GWS=$m timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed "s/$/ [GWS=$m]/"; done
