#!/bin/bash
# Microbenchmark of library variants of the wave-specialised 3x3 kernel (tools/dev/build_variant.sh), two passes each.
export SBMC_CONV3X3_WS=1
for rep in 1 2; do for v in base "$@"; do
  lib=$PWD/.ab/lib$v.so; [ $v = base ] && lib=$PWD/sbmc_amd/libsbmc_hip.so
  echo "== $v"
  SBMC_HIP_LIB=$lib timeout 600 python tools/conv3x3_experiment.py --shapes all 2>&1 | grep "ours" | grep -v "values\|weight" | cut -c1-110
done; done
