cd $GRAFT_REPO_ROOT
o=gpurun_out/r04l; mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_backbone.py tests/test_gpu_golden.py tests/test_dist_gpu.py tests/test_gpu_conv3x3_half.py -q 2>&1 | tail -4 | tee $o/tests.txt
for ps in 1 0; do
SBMC_POOL_SKIP=$ps timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed "s/$/ [POOL_SKIP=$ps]/" | tee -a $o/rank_cost.txt
done
timeout 400 python tools/rank_cost.py --ipc-self 8 2>&1 | grep "^world" | tee -a $o/rank_cost.txt
timeout 600 python bench.py --fp16-activations --no-cpu-baseline --no-stages --steps 10 > $o/train_fp16.json 2>/dev/null; head -c 300 $o/train_fp16.json; echo
