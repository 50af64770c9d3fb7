#!/bin/bash
# rocprofv3 kernel-trace summary of the model workload only (top kernels by total time)
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_model
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export MIOPEN_FIND_MODE=FAST
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o model -- python $root/bench.py --workload model --steps 2 --warmup 2 --no-cpu-baseline --no-stages > $out/model.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms (4 steps): %.1f" % (tot / 1e6))
for r in rows[:32]:
    print("%-70s %5s %9.2f ms %5.1f%%  avg %.3f" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
          float(r["TotalDurationNs"]) / tot * 100, float(r["AverageNs"]) / 1e6))
PY
