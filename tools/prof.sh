#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun):  ./tools/prof.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the splat workload and of the model workload
#   2. PMC passes (one counter group per run; never combined with trace domains other than
#      --kernel-trace) for the splat kernels: FETCH_SIZE / WRITE_SIZE / TCC / SQ
# Raw output goes to gpurun_out/prof_<tag>/, condensed summaries to gpurun_out/profiles_<tag>/
# (copy those into profiles/ to commit them).
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_$tag
sum=$root/gpurun_out/profiles_$tag
mkdir -p $out $sum
cd /tmp && export TMPDIR=/tmp
export MIOPEN_FIND_MODE=FAST
rocprofv3 --kernel-trace --stats --output-format csv -d $out/splat -o splat -- python $root/bench.py --workload splat --steps 5 --warmup 2 --no-cpu-baseline > $out/splat.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/model -o model -- python $root/bench.py --workload model --steps 6 --warmup 3 --no-cpu-baseline --no-stages > $out/model.log 2>&1
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $out/p$i -o p -- python $root/bench.py --workload splat --steps 1 --warmup 1 --no-cpu-baseline > $out/p$i.log 2>&1
done
python $root/tools/pmc_summary.py $out $sum $tag
grep -h '^{' $out/splat.log $out/model.log > $sum/${tag}_bench_lines_under_rocprof.jsonl
ls $sum
