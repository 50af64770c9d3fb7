#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/l
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/l/prof -o m -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-stages > $GRAFT_REPO_ROOT/gpurun_out/l/bench.log 2>&1
echo rc=$?
