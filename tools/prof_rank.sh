# rocprofv3 kernel stats of one rank of N (tools/rank_cost.py [--rccl-self] N):  gpurun -- bash tools/prof_rank.sh N [--rccl-self]
cd /tmp && export TMPDIR=/tmp
n=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/q/prof_rank$n
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r -- python $GRAFT_REPO_ROOT/tools/rank_cost.py "$@" $n > $out.log 2>&1
grep world $out.log
f=$(find $out -name r_kernel_stats.csv | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/q/rank${n}_stats.csv
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("kernel time per step (5 steps): %.2f ms" % (tot / 5e6))
for r in rows:
    if "ccl" in r["Name"].lower() or "SendRecv" in r["Name"] or "AllReduce" in r["Name"] or "halo::" in r["Name"]:
        print("  %8.3f ms/step %5d calls  %s" % (int(r["TotalDurationNs"]) / 5e6, int(r["Calls"]), r["Name"][:90]))
PY
