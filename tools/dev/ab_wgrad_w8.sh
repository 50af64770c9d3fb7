#!/bin/bash
# A/B of the eight-wave weight gradient (SBMC_CONV3X3_WGRAD_W8=1): values, then time.
for v in 1 0; do
  echo "== SBMC_CONV3X3_WGRAD_W8=$v: tests"
  SBMC_CONV3X3_WGRAD_W8=$v timeout 1200 python -m pytest tests/test_gpu_conv3x3.py tests/test_gpu_backbone.py -x -q -m gpu 2>&1 | tail -2
done
for rep in 1 2; do for v in 0 1; do
  echo "== SBMC_CONV3X3_WGRAD_W8=$v: experiment"
  SBMC_CONV3X3_WGRAD_W8=$v timeout 600 python tools/conv3x3_experiment.py --shapes all 2>&1 | grep "weight gradient" | cut -c1-200 | tr '\n' '|'; echo
done; done
bash tools/ab_knob.sh SBMC_CONV3X3_WGRAD_W8 0 1
