cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pointwise_scaled.py tests/test_gpu_backbone.py -q -x 2>&1 | grep -E "^E  |passed|failed|Error" | head -10
python tools/bench_pw_scaled.py 2>&1 | grep -E "^fwd|^bwd"
