cd $GRAFT_REPO_ROOT
o=gpurun_out/r04j; mkdir -p $o
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $o/tests_all.txt; cat $o/tests_all.txt
timeout 600 python bench.py --fp16-activations --no-cpu-baseline --no-stages --steps 10 > $o/train_fp16.json 2>/dev/null; head -c 300 $o/train_fp16.json; echo
timeout 600 python bench.py --workload infer --spp 32 --fp16-activations > $o/infer32_fp16.json 2>/dev/null; head -c 300 $o/infer32_fp16.json; echo
