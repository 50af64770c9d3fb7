#!/bin/bash
# Same-box A/B of the training step (and an interior rank of 8) between the library in the tree and another build:
#   cp sbmc_amd/libsbmc_hip.so .ab/libsbmc_head.so   (before changing a kernel; .ab/ travels with the snapshot, git ignores it)
#   tools/grun bash tools/ab_step.sh [.ab/libsbmc_head.so]
cd ${GRAFT_REPO_ROOT:-/root/repo}
other=${1:-.ab/libsbmc_head.so}
for lib in "" "$PWD/$other" "" "$PWD/$other"; do
  SBMC_HIP_LIB=$lib timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed "s#\$# [${lib:-tree}]#"
done
for lib in "" "$PWD/$other"; do
  SBMC_HIP_LIB=$lib timeout 400 python tools/rank_cost.py --ipc-self 8 2>&1 | grep "^world" | sed "s#\$# [${lib:-tree}]#"
done
