#!/bin/bash
# A/B of the 1x1 backward's row requests: inline-assembly LDS-DMA with the kernel's own waits (default build) against the
# builtin + __syncthreads (.ab/libpwold.so: the previous commit's pointwise.hip)
echo "== tests (default build)"
timeout 1500 python -m pytest tests/test_gpu_pointwise_scaled.py tests/test_gpu_ops.py tests/test_gpu_backbone.py tests/test_gpu_pointwise_chain.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python tools/fuzz_pointwise.py --seconds 60 2>&1 | tail -1 | cut -c1-200
for rep in 1 2; do for v in pwold base; do
  lib=$PWD/.ab/lib$v.so; [ $v = base ] && lib=$PWD/sbmc_amd/libsbmc_hip.so
  echo "== $v"; SBMC_HIP_LIB=$lib timeout 300 python tools/bench_pw_scaled.py 2>&1 | grep "bwd" | cut -c1-150
done; done
for rep in 1 2; do for v in pwold base; do
  lib=$PWD/.ab/lib$v.so; [ $v = base ] && lib=$PWD/sbmc_amd/libsbmc_hip.so
  SBMC_HIP_LIB=$lib timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed "s/$/ [$v]/"
done; done
