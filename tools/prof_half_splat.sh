# PMC pass over the fp16-storage splat stage (bench.py's model workload runs it as a stage): is the half strip
# kernel memory- or VALU-bound?   gpurun -- bash tools/prof_half_splat.sh
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/half_splat
mkdir -p $out
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/pmc -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 2 --no-cpu-baseline > $out/pmc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/st -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 2 --no-cpu-baseline > $out/st.log 2>&1
python - <<'PY'
import csv, glob, collections, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/half_splat"
f = glob.glob(out + "/pmc/**/p_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "strip_kernel" in n:
        acc[n.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as o:
    for k, d in acc.items():
        for c, v in sorted(d.items()):
            o.write("%-60s %-22s avg=%.6g n=%d\n" % (k, c, sum(v) / len(v), len(v)))
    s = glob.glob(out + "/st/**/s_kernel_stats.csv", recursive=True)[0]
    for r in csv.DictReader(open(s)):
        if "strip_kernel" in r["Name"]:
            o.write("%-60s calls=%s avg_ns=%s\n" % (r["Name"].split("(")[0], r["Calls"], r["AverageNs"]))
print(open(out + "/summary.txt").read())
PY
