cd $GRAFT_REPO_ROOT
for lib in "" widepairs "" widepairs; do
  echo -n "${lib:-tree}: "; SBMC_HIP_LIB=${lib:+$PWD/.ab/lib$lib.so} python tools/bench_pw_scaled.py --wide 2>&1 | tail -1
done
