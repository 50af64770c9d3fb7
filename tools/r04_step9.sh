cd $GRAFT_REPO_ROOT
o=gpurun_out/r04i; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_conv3x3.py -q 2>&1 | tail -3 | tee $o/tests.txt
for sk in 1 0; do
  export SBMC_CONV3X3_STREAMK=$sk
  bash tools/prof_rank.sh 8 --ipc-self > $o/prof_rank_sk$sk.log 2>&1
  cp gpurun_out/q/rank8_stats.csv $o/rank8_sk$sk.csv
  python tools/prof_rank_cat.py $o/rank8_sk$sk.csv > $o/rank8_sk$sk.txt
  echo "== STREAMK=$sk"; head -4 $o/rank8_sk$sk.txt; grep -i "conv3_\|fixup" $o/rank8_sk$sk.txt
  rm -rf gpurun_out/q
done
unset SBMC_CONV3X3_STREAMK
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$o/prof -o m -- python $GRAFT_REPO_ROOT/bench.py --fp16-activations --steps 6 --warmup 3 --no-cpu-baseline --no-stages > $GRAFT_REPO_ROOT/$o/fp16.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $o/prof -name "*kernel_stats.csv" | head -1); cp $f $o/fp16_train_kernel_stats.csv; rm -rf $o/prof
python tools/prof_rank_cat.py $o/fp16_train_kernel_stats.csv 9 > $o/fp16_train_categories.txt; head -45 $o/fp16_train_categories.txt
