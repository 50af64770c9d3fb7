cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in "" "$GRAFT_REPO_ROOT/.ab/libsbmc_hip_prev.so"; do
  echo "lib=${lib:-current}"
  SBMC_HIP_LIB=$lib timeout 400 python tools/bench_pointwise.py --notest --bwd 2>&1 | tail -3
done; done
for lib in "" "$GRAFT_REPO_ROOT/.ab/libsbmc_hip_prev.so"; do
  SBMC_HIP_LIB=$lib timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed "s#\$# [${lib:-current}]#"
done
