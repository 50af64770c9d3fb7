#!/bin/bash
# Counters of the forward strip kernel per launch, logits at four offsets of one allocation (tools/placement_experiment7.py):
# separate --pmc passes, kernel-trace only.  Summary -> gpurun_out/placement_pmc.txt
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_place
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python $root/tools/placement_experiment7.py 2>&1 | grep "logits at" > $out/plain.txt
i=0
for pmc in "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUBBLE_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $out/p$i -o p -- python $root/tools/placement_experiment7.py > $out/p$i.log 2>&1
done
python - $out $root/gpurun_out/placement_pmc.txt <<'PY'
import collections, csv, glob, os, sys
root, dst = sys.argv[1], sys.argv[2]
lines = ["# tools/placement_pmc.sh: splat_fwd_strip_kernel<21,3>, 8 samples per launch, logits at 0 / 14 / 28 / 42 GB of one 60 GB allocation,",
         "# 6 launches per offset; plain run (HIP events):"] + ["#   " + l.strip() for l in open(os.path.join(root, "plain.txt"))]
for d in sorted(glob.glob(os.path.join(root, "p[0-9]"))):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "splat_fwd_strip_kernel" in r["Kernel_Name"]]
    dur = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "splat_fwd_strip_kernel" in r["Kernel_Name"]:
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    per = collections.OrderedDict()
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        per.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = list(per)
    lines.append("pass %s: %d launches" % (os.path.basename(d), len(ids)))
    for g in range(0, len(ids), 6):
        grp = ids[g:g + 6]
        names = sorted(per[grp[0]])
        ms = [dur.get(i, float("nan")) for i in grp]
        lines.append("  offset #%d: ms under the counters %s" % (g // 6, " ".join("%.3f" % m for m in ms)))
        for n in names:
            vals = [per[i].get(n, float("nan")) for i in grp]
            lines.append("      %-40s mean %.4g   (%s)" % (n, sum(vals) / len(vals), " ".join("%.4g" % v for v in vals)))
open(dst, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
