#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "half or fp16 or float16 or pointwise" > gpurun_out/g/tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/g/tests.log
timeout 600 python tools/bench_pointwise_f16.py > gpurun_out/g/pw_f16.txt 2>&1; echo "bench rc=$?"; cat gpurun_out/g/pw_f16.txt | grep -v MIOpen
timeout 600 python bench.py --workload infer --spp 32 --fp16-activations --steps 5 --warmup 3 > gpurun_out/g/bench_infer32_fp16.json 2> /dev/null; head -c 400 gpurun_out/g/bench_infer32_fp16.json; echo
timeout 600 python bench.py --fp16-activations --steps 5 --warmup 3 --no-cpu-baseline --no-stages > gpurun_out/g/bench_train_fp16.json 2> /dev/null; head -c 400 gpurun_out/g/bench_train_fp16.json; echo
