#!/bin/bash
# A/B of the wave-specialised 3x3 kernel (SBMC_CONV3X3_WS=1) against the one-wave-per-SIMD form: values, then time.
o=gpurun_out/ws; mkdir -p $o
for ws in 1 0; do
  echo "== SBMC_CONV3X3_WS=$ws: tests"
  SBMC_CONV3X3_WS=$ws timeout 1200 python -m pytest tests/test_gpu_conv3x3.py tests/test_gpu_backbone.py -x -q -m gpu 2>&1 | tail -3
done
for ws in 0 1; do
  echo "== SBMC_CONV3X3_WS=$ws: experiment"
  SBMC_CONV3X3_WS=$ws timeout 600 python tools/conv3x3_experiment.py --shapes all 2>&1 | grep "ours" | grep -v values | cut -c1-200
done
bash tools/ab_knob.sh SBMC_CONV3X3_WS 0 1
