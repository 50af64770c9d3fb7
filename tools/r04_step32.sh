cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_conv3x3.py tests/test_gpu_conv3x3_half.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python tools/fuzz_conv3x3.py --cases 100 2>&1 | tail -1
for lib in "" "$GRAFT_REPO_ROOT/.ab/libsbmc_head.so"; do
  SBMC_HIP_LIB=$lib timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed "s#\$# [${lib:-current}]#"
done
for lib in "" "$GRAFT_REPO_ROOT/.ab/libsbmc_head.so"; do
  SBMC_HIP_LIB=$lib timeout 400 python tools/rank_cost.py --ipc-self 8 2>&1 | grep "^world" | sed "s#\$# [${lib:-current}]#"
  SBMC_HIP_LIB=$lib timeout 600 python bench.py --fp16-activations --no-cpu-baseline --no-stages --steps 6 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp16 train ms', d['ms_per_step'])"
done
