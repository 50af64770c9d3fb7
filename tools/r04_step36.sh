cd $GRAFT_REPO_ROOT
export SBMC_HIP_PW_FWD_NT32=1
timeout 400 python tools/bench_pointwise.py --time 2>&1 | grep -v "^MIOpen" | tail -11 | cut -c1-80
timeout 1200 python -m pytest tests/test_gpu_backbone.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/fuzz_pointwise.py --seconds 40 2>&1 | tail -1 | cut -c1-200
unset SBMC_HIP_PW_FWD_NT32
timeout 400 python tools/bench_pointwise.py --notest --time 2>&1 | grep "^cin" | cut -c1-80
for v in 0 1; do SBMC_HIP_PW_FWD_NT32=$v timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed "s/$/ [NT32=$v]/"; done
for v in 0 1; do SBMC_HIP_PW_FWD_NT32=$v timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed "s/$/ [NT32=$v]/"; done
