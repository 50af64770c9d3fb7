cd $GRAFT_REPO_ROOT
o=gpurun_out/r04m; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_conv3x3_half.py tests/test_gpu_conv3x3.py -q 2>&1 | tail -3 | tee $o/tests.txt
timeout 600 python bench.py --fp16-activations --no-cpu-baseline --no-stages --steps 10 > $o/train_fp16.json 2>/dev/null; head -c 300 $o/train_fp16.json; echo
timeout 600 python bench.py --workload infer --spp 32 --fp16-activations > $o/infer32_fp16.json 2>/dev/null; head -c 300 $o/infer32_fp16.json; echo
