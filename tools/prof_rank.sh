# rocprofv3 kernel stats of one rank of N (tools/rank_cost.py N):  gpurun -- bash tools/prof_rank.sh N
cd /tmp && export TMPDIR=/tmp
n=$1
out=$GRAFT_REPO_ROOT/gpurun_out/q/prof_rank$n
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r -- python $GRAFT_REPO_ROOT/tools/rank_cost.py $n > $out.log 2>&1
grep world $out.log
f=$(find $out -name r_kernel_stats.csv | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/q/rank${n}_stats.csv
