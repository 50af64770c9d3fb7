#!/bin/bash
# final round-2 measurements: bench lines, profiles, rank costs, the whole GPU suite, smoke
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/p
( time timeout 1500 python bench.py > gpurun_out/p/r02_bench.json 2> gpurun_out/p/r02_bench.err ); echo "bench rc=$?"
head -c 420 gpurun_out/p/r02_bench.json; echo
for spp in 4 8 32; do timeout 900 python bench.py --workload infer --spp $spp > gpurun_out/p/infer$spp.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/p/infer$spp.json'));print('infer',$spp, d['value'], d['ms_per_step'])"; done
timeout 900 python bench.py --workload infer --spp 32 --fp16-activations > gpurun_out/p/infer32_fp16.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/p/infer32_fp16.json'));print('infer 32 fp16', d['value'], d['ms_per_step'])"
timeout 900 python bench.py --fp16-activations --no-cpu-baseline --no-stages > gpurun_out/p/train_fp16.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/p/train_fp16.json'));print('train fp16', d['value'], d['ms_per_step'])"
timeout 900 python scripts/bench_ops.py > gpurun_out/p/r02_bench_ops.jsonl 2>/dev/null
bash tools/prof.sh r02 > gpurun_out/p/prof.log 2>&1; echo "prof rc=$?"
timeout 900 python tools/rank_cost.py 1 2 4 8 > gpurun_out/p/rank_cost.txt 2>&1; grep world gpurun_out/p/rank_cost.txt
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/p/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/p/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
SBMC_BENCH_BACKEND=gloo SBMC_BENCH_SINGLE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 4 --warmup 3 --no-cpu-baseline > gpurun_out/p/bench_2rank_gloo.json 2> gpurun_out/p/bench_2rank_gloo.err; echo "2-rank rc=$?"; head -c 300 gpurun_out/p/bench_2rank_gloo.json; echo
