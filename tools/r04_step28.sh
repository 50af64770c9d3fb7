cd $GRAFT_REPO_ROOT
for n in 0 1 2 3; do
  echo "PW_ABL=$n (1: no y stores, 2: no MFMAs, 3: no split / LDS writes)"
  SBMC_HIP_LIB=$GRAFT_REPO_ROOT/.ab/libsbmc_abl$n.so timeout 400 python tools/bench_pointwise.py --notest --time 2>&1 | grep "^cin" | cut -c1-75
done
