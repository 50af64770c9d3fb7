for l in "" $(ls sbmc_amd/libab_*.so); do echo "== ${l:-base}"; SBMC_HIP_LIB=$l timeout 300 python tools/bench_pointwise.py --notest $* 2>&1 | grep -E "fused|bwd"; done
