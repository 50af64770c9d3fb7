cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_conv3x3_half.py -x -q -m gpu 2>&1 | tail -2
bash tools/measure_round.sh r04
