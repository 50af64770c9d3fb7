cd $GRAFT_REPO_ROOT
o=gpurun_out/r04r; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_backbone.py tests/test_gpu_configs.py -q 2>&1 | tail -3
timeout 600 python bench.py --workload infer --spp 32 --fp16-activations > $o/infer32_fp16.json 2>/dev/null; head -c 260 $o/infer32_fp16.json; echo
timeout 600 python bench.py --fp16-activations --no-cpu-baseline --no-stages > $o/train_fp16.json 2>/dev/null; head -c 260 $o/train_fp16.json; echo
cat $o/infer32_fp16.json $o/train_fp16.json | grep '^{' > $o/r04_bench_fp16_activations.jsonl
