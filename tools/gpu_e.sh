#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/e
timeout 1200 python -m pytest tests/test_gpu_ops.py -q -x -k "float16 or half or fp16 or pointwise" > gpurun_out/e/n4_tests.log 2>&1; echo "n4 tests rc=$?"; tail -15 gpurun_out/e/n4_tests.log
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/e/all_gpu_tests.log 2>&1; echo "all gpu tests rc=$?"; tail -14 gpurun_out/e/all_gpu_tests.log
timeout 600 python bench.py --fp16-activations --steps 5 --warmup 3 --no-cpu-baseline --no-stages > gpurun_out/e/bench_train_fp16.json 2> gpurun_out/e/bench_train_fp16.err; echo "fp16 train rc=$?"; head -c 700 gpurun_out/e/bench_train_fp16.json; echo
timeout 600 python bench.py --workload infer --spp 32 --fp16-activations --steps 5 --warmup 3 > gpurun_out/e/bench_infer32_fp16.json 2> gpurun_out/e/bench_infer32_fp16.err; echo "fp16 infer rc=$?"; head -c 500 gpurun_out/e/bench_infer32_fp16.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/e/prof_rank8 -o r8 -- python $GRAFT_REPO_ROOT/tools/rank_cost.py 8 > $GRAFT_REPO_ROOT/gpurun_out/e/prof_rank8.log 2>&1; echo "prof rc=$?"
tail -2 $GRAFT_REPO_ROOT/gpurun_out/e/prof_rank8.log
