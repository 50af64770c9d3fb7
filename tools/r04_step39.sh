cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_backbone.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/fuzz_pointwise.py --seconds 40 2>&1 | tail -1 | cut -c1-300
for lib in "" "$GRAFT_REPO_ROOT/.ab/libsbmc_head.so" "" "$GRAFT_REPO_ROOT/.ab/libsbmc_head.so"; do
  echo "lib=${lib:-current}"
  SBMC_HIP_LIB=$lib timeout 400 python tools/bench_pointwise.py --notest --bwd 2>&1 | tail -3
done
for lib in "" "$GRAFT_REPO_ROOT/.ab/libsbmc_head.so"; do
  SBMC_HIP_LIB=$lib timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed "s#\$# [${lib:-current}]#"
done
