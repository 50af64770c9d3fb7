cd $GRAFT_REPO_ROOT
o=gpurun_out/r04o; mkdir -p $o
SBMC_BENCH_BACKEND=gloo SBMC_BENCH_SINGLE_DEVICE=1 OMP_NUM_THREADS=8 timeout 1200 python bench.py --gpus 8 --steps 3 --warmup 2 --no-cpu-baseline --no-stages > $o/bench8.json 2> $o/bench8.err.txt; echo "8 ranks on one GPU rc=$?"
grep -v "^\[Gloo\|^MIOpen\|socket.cpp\|amdgpu.ids" $o/bench8.err.txt | tail -40
grep "^{" $o/bench8.json | head -c 1500; echo
timeout 600 python -m pytest tests/test_gpu_backbone.py -q -k "half_layer or pointwise or chain" 2>&1 | tail -3
timeout 600 python bench.py --workload infer --spp 32 --fp16-activations > $o/infer32_fp16.json 2>/dev/null; head -c 260 $o/infer32_fp16.json; echo
timeout 600 python bench.py --fp16-activations --no-cpu-baseline --no-stages --steps 10 > $o/train_fp16.json 2>/dev/null; head -c 260 $o/train_fp16.json; echo
