#!/bin/bash
# same-box A/B of an environment knob on the forward-only workloads:   tools/grun bash tools/ab_knob_infer.sh SBMC_CONV3X3_NOSIGNS 0
cd $GRAFT_REPO_ROOT
k=$1; v=$2
for i in 1 2 3; do for set in "" "$v"; do
  for wl in "--workload infer --spp 4" "--workload infer --spp 32 --fp16-activations"; do
    if [ -n "$set" ]; then export $k=$set; else unset $k; fi
    timeout 400 python bench.py $wl --steps 10 --warmup 5 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$k=${set:-default}', '$wl', d['ms_per_step'], d.get('ms_per_step_median'))"
  done
done; done
