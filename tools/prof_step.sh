#!/bin/bash
# Kernel statistics of the training step (rocprofv3) + per-category split, and a same-box A/B of the step over env knobs:
#   tools/grun --timeout 1500 "bash tools/prof_step.sh <tag> [KNOB=V ...]"     -> gpurun_out/<tag>/
tag=${1:-step}; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
o=$root/gpurun_out/$tag
mkdir -p $o
cd $root
for knobs in "" "$@"; do
  for rep in 1 2; do env $knobs timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed "s#\$# [${knobs:-default}]#"; done
done | tee $o/${tag}_ab.txt
bash tools/prof_model.sh > $o/prof_model.log 2>&1
f=$(find gpurun_out/prof_model -name "*kernel_stats.csv" | head -1)
cp $f $o/${tag}_model_kernel_stats.csv
python tools/prof_rank_cat.py $o/${tag}_model_kernel_stats.csv 4 > $o/${tag}_model_categories.txt
head -16 $o/${tag}_model_categories.txt
grep -E "pw_|Cijk" $o/${tag}_model_categories.txt | head -40
rm -rf gpurun_out/prof_model
