cd $GRAFT_REPO_ROOT
o=gpurun_out/r04v; mkdir -p $o
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$o/prof -o m -- python $GRAFT_REPO_ROOT/bench.py --fp16-activations --steps 6 --warmup 3 --no-cpu-baseline --no-stages > $GRAFT_REPO_ROOT/$o/fp16.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $o/prof -name "*kernel_stats.csv" | head -1); cp $f $o/fp16_train_kernel_stats.csv; rm -rf $o/prof
python tools/prof_rank_cat.py $o/fp16_train_kernel_stats.csv 11 > $o/fp16_train_categories.txt; head -48 $o/fp16_train_categories.txt
