cd $GRAFT_REPO_ROOT
o=gpurun_out/r04k; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_backbone.py -q -k "wide or regressor" 2>&1 | tail -3 | tee $o/tests.txt
timeout 600 python bench.py --fp16-activations --no-cpu-baseline --no-stages --steps 10 > $o/train_fp16.json 2>/dev/null; head -c 300 $o/train_fp16.json; echo
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_fullsize.py -q 2>&1 | tail -3 | tee $o/tests_cfg.txt
