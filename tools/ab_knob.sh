#!/bin/bash
# Same-box A/B of an integer environment knob on the training step (whole frame, an interior rank of 8, a rank of 4):
#   tools/grun bash tools/ab_knob.sh SBMC_CONV3X3_SK_SAVED 0 4 3
cd ${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
for round in 1 2; do for v in "$@"; do
  env $name=$v timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed "s#\$# [$name=$v]#"
done; done
for v in "$@"; do
  env $name=$v timeout 400 python tools/rank_cost.py --ipc-self 8 2>&1 | grep "^world" | sed "s#\$# [$name=$v]#"
  env $name=$v timeout 400 python tools/rank_cost.py --ipc-self 4 2>&1 | grep "^world" | sed "s#\$# [$name=$v]#"
done
