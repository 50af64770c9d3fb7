cd $GRAFT_REPO_ROOT
o=gpurun_out/r04e; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_conv3x3_half.py tests/test_gpu_conv3x3.py -q -x 2>&1 | tail -15 > $o/tests.txt
cat $o/tests.txt
timeout 600 python bench.py --fp16-activations --no-cpu-baseline --no-stages --steps 10 > $o/train_fp16.json 2>$o/err1.txt; head -c 300 $o/train_fp16.json; echo
SBMC_CONV3X3_HALF=0 timeout 600 python bench.py --fp16-activations --no-cpu-baseline --no-stages --steps 10 > $o/train_fp16_miopen.json 2>/dev/null; head -c 300 $o/train_fp16_miopen.json; echo
timeout 600 python bench.py --workload infer --spp 32 --fp16-activations > $o/infer32_fp16.json 2>/dev/null; head -c 300 $o/infer32_fp16.json; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$o/prof -o model -- python $GRAFT_REPO_ROOT/bench.py --workload model --steps 6 --warmup 3 --no-cpu-baseline --no-stages > $GRAFT_REPO_ROOT/$o/model.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $o/prof -name "*kernel_stats.csv" | head -1); cp $f $o/model_kernel_stats.csv; rm -rf $o/prof
python tools/prof_rank_cat.py $o/model_kernel_stats.csv 9 > $o/model_categories.txt; head -60 $o/model_categories.txt
