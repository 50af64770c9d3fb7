cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r/prof_fp16 -o f -- python $GRAFT_REPO_ROOT/bench.py --fp16-activations --steps 6 --warmup 3 --no-cpu-baseline --no-stages > $GRAFT_REPO_ROOT/gpurun_out/r/fp16.log 2>&1
echo rc=$?
