cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/q/prof_rank8 -o r8 -- python $GRAFT_REPO_ROOT/tools/rank_cost.py 8 > $GRAFT_REPO_ROOT/gpurun_out/q/prof_rank8.log 2>&1
grep world $GRAFT_REPO_ROOT/gpurun_out/q/prof_rank8.log
