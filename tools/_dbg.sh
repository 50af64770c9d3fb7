for d in 0 1 2 4 32 0; do echo "DBG $d: $(SBMC_CONV3_DBG=$d python tools/conv3x3_experiment.py --shapes 720p --reps 20 2>&1 | grep '720x1280' | sed -e 's/.*ours/ours/' | cut -c1-60)"; done
python -m pytest tests/test_gpu_conv3x3.py -q -x 2>&1 | tail -2
