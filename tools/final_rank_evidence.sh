#!/bin/bash
# The sharded-step evidence of the final tree in one GPU call:   tools/grun --timeout 2400 "bash tools/final_rank_evidence.sh r05"
# per-rank cost (IPC mailboxes / RCCL to self, 8 and 4 ranks, 4K), the 8-rank partition as 8 processes on this GPU, kernel
# statistics + categories of an interior rank of 8, the fp16-activation lines.
tag=${1:-r05}
cd $GRAFT_REPO_ROOT
o=gpurun_out/$tag
mkdir -p $o
( for mode in "--ipc-self" "--rccl-self"; do timeout 400 python tools/rank_cost.py $mode 8 2>&1 | grep "^world"; done
  timeout 400 python tools/rank_cost.py --ipc-self 4 2>&1 | grep "^world"
  timeout 600 python tools/rank_cost.py --ipc-self --4k 8 2>&1 | grep "^world" | sed "s/$/ [3840x2160]/"
  timeout 600 python tools/rank_cost.py 1 2>&1 | grep "^world" ) | tee $o/${tag}_rank_cost.txt
SBMC_BENCH_BACKEND=gloo SBMC_BENCH_SINGLE_DEVICE=1 OMP_NUM_THREADS=8 timeout 1200 python bench.py --gpus 8 --steps 3 --warmup 2 --no-cpu-baseline --no-stages > $o/${tag}_bench_8ranks_one_gpu.json 2> $o/bench8.err; echo "8 ranks on one GPU rc=$?"
head -c 300 $o/${tag}_bench_8ranks_one_gpu.json; echo
bash tools/prof_rank.sh 8 --ipc-self > $o/prof_rank.log 2>&1; cp gpurun_out/q/rank8_stats.csv $o/${tag}_rank8_kernel_stats.csv; rm -rf gpurun_out/q
python tools/prof_rank_cat.py $o/${tag}_rank8_kernel_stats.csv 5 > $o/${tag}_rank8_categories.txt 2>/dev/null; head -16 $o/${tag}_rank8_categories.txt
timeout 900 python bench.py --workload infer --spp 32 --fp16-activations > $o/infer32_fp16.json 2>/dev/null
timeout 900 python bench.py --fp16-activations --no-cpu-baseline --no-stages > $o/train_fp16.json 2>/dev/null
cat $o/infer32_fp16.json $o/train_fp16.json | grep '^{' > $o/${tag}_bench_fp16_activations.jsonl
python - <<PY
import json
for l in open("$o/${tag}_bench_fp16_activations.jsonl"):
    d = json.loads(l); print("fp16 activations:", d["config"]["workload"][:40], d["value"], d["ms_per_step"])
PY
