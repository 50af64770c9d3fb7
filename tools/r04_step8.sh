cd $GRAFT_REPO_ROOT
o=gpurun_out/r04h; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_backbone.py -q -k "wide or regressor_output or pointwise" 2>&1 | tail -4 | tee $o/tests.txt
for sk in 1 0; do
  export SBMC_CONV3X3_STREAMK=$sk
  bash tools/prof_rank.sh 8 --ipc-self > $o/prof_rank_sk$sk.log 2>&1
  cp gpurun_out/q/rank8_stats.csv $o/rank8_sk$sk.csv
  python tools/prof_rank_cat.py $o/rank8_sk$sk.csv > $o/rank8_sk$sk.txt
  echo "== STREAMK=$sk"; head -8 $o/rank8_sk$sk.txt; grep -i "conv3_\|fixup" $o/rank8_sk$sk.txt
  rm -rf gpurun_out/q
done
unset SBMC_CONV3X3_STREAMK
for w in 1 0; do SBMC_HIP_PW_GW_WIDE=$w timeout 400 python tools/rank_cost.py 1 2>&1 | grep "^world" | sed "s/$/ [GW_WIDE=$w]/" | tee -a $o/rank_cost.txt; done
