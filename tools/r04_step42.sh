cd $GRAFT_REPO_ROOT
for n in 0 1 2 3 4 5; do
  lib=$GRAFT_REPO_ROOT/.ab/libsbmc_cv$n.so; [ $n = 0 ] && lib=""
  echo "CV_ABL=$n (1 no stage barriers, 2 no weight staging, 3 no patch staging, 4 no operand fetches, 5 no MFMAs)"
  SBMC_HIP_LIB=$lib timeout 300 python tools/conv3x3_experiment.py --shapes 720p --reps 10 2>&1 | grep "ours" | sed 's/MIOpen [0-9.]* ms ([0-9]* TFLOP.s)//' | cut -c1-150
done
