cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_pointwise_scaled.py tests/test_gpu_backbone.py -q 2>&1 | tail -3
python tools/bench_pw_scaled.py 2>&1 | grep -E "^fwd|^bwd"
