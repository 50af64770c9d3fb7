#!/bin/bash
# The wide layer's one-pass backward: rows two steps ahead (default) against SBMC_PW_WIDE_G2=0 and the previous commit's library
timeout 900 python -m pytest tests/test_gpu_pointwise_scaled.py tests/test_gpu_fullsize.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python tools/fuzz_pointwise.py --seconds 60 2>&1 | tail -1 | cut -c1-260
for rep in 1 2; do for v in 0 1; do
  echo "== SBMC_PW_WIDE_G2=$v"; SBMC_PW_WIDE_G2=$v timeout 300 python tools/bench_pw_scaled.py 2>&1 | grep "bwd 128->441" | cut -c1-150
done; done
bash tools/ab_knob.sh SBMC_PW_WIDE_G2 0 1 2>&1 | grep "rows 720"
