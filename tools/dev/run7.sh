cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_pointwise_scaled.py tests/test_gpu_golden.py -q 2>&1 | tail -2
for lib in "" pwabl1 pwabl2 pwabl3; do
  echo "== ${lib:-tree}"
  SBMC_HIP_LIB=${lib:+$PWD/.ab/lib$lib.so} python tools/bench_pw_scaled.py 2>&1 | grep "^fwd"
done
