#!/bin/bash
# round 2, first GPU pass: new slab tests, whole GPU suite, bench line, rank cost bound, 2-rank dry run
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/a
timeout 900 python -m pytest tests/test_gpu_slab.py -x -q > gpurun_out/a/slab.log 2>&1; echo "slab rc=$?" 
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/a/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -3 gpurun_out/a/slab.log; tail -5 gpurun_out/a/gpu_tests.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/a/bench.json 2> gpurun_out/a/bench.err; echo "bench rc=$?"
timeout 600 python tools/rank_cost.py 1 2 4 8 > gpurun_out/a/rank_cost.txt 2>&1; echo "rank_cost rc=$?"; cat gpurun_out/a/rank_cost.txt | tail -5
SBMC_BENCH_BACKEND=gloo SBMC_BENCH_SINGLE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/a/bench_2rank_gloo.json 2> gpurun_out/a/bench_2rank_gloo.err; echo "2-rank rc=$?"
SBMC_BENCH_BACKEND=gloo SBMC_BENCH_SINGLE_DEVICE=1 timeout 600 python bench.py --gpus 2 --workload splat --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/a/bench_2rank_splat.json 2> gpurun_out/a/bench_2rank_splat.err; echo "2-rank splat rc=$?"
head -c 600 gpurun_out/a/bench_2rank_gloo.json; echo; head -c 600 gpurun_out/a/bench_2rank_splat.json; echo
head -c 1500 gpurun_out/a/bench.json
