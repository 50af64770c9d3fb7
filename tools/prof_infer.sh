#!/bin/bash
# rocprofv3 kernel statistics of the inference workload (BASELINE configs[1]: forward only, 4 spp) + its categories:
#   gpurun -- bash tools/prof_infer.sh r06   ->  gpurun_out/<tag>_infer4_kernel_stats.csv, _categories.txt
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=/tmp/prof_infer
mkdir -p $out $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
export MIOPEN_FIND_MODE=FAST
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o infer -- python $root/bench.py --workload infer --spp 4 --steps 6 --warmup 3 --no-cpu-baseline > $out/infer.log 2>&1
f=$(ls $out/*/infer_kernel_stats.csv $out/infer_kernel_stats.csv 2>/dev/null | head -1)
cp $f $root/gpurun_out/${tag}_infer4_kernel_stats.csv
python $root/tools/prof_rank_cat.py $root/gpurun_out/${tag}_infer4_kernel_stats.csv 11 > $root/gpurun_out/${tag}_infer4_categories.txt
grep -h '^{' $out/infer.log | head -c 400; echo
head -40 $root/gpurun_out/${tag}_infer4_categories.txt
