#!/bin/bash
# PMC counters of the 3 x 3 convolution kernels (one counter group per run, kernel-trace only).
# Summary -> gpurun_out/conv3x3_pmc.txt
root=${GRAFT_REPO_ROOT:-/root/repo}
out=/tmp/prof_cv
mkdir -p $out $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $out/p$i -o p -- python $root/tools/conv3x3_experiment.py --shapes ${1:-720p} --reps 5 > $out/p$i.log 2>&1
done
python - $out $root/gpurun_out/conv3x3_pmc.txt <<'PY'
import collections, csv, glob, os, sys
root, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
dur = collections.defaultdict(lambda: [0.0, 0])
def key(n):
    return n.replace("void ", "").split("(")[0]
for f in sorted(glob.glob(os.path.join(root, "p*", "*counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "conv3_" not in n and "igemm" not in n:
            continue
        acc[(key(n)[:60], row["Counter_Name"])][0] += float(row["Counter_Value"]); acc[(key(n)[:60], row["Counter_Name"])][1] += 1
for f in sorted(glob.glob(os.path.join(root, "p1", "*kernel_trace.csv"))):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "conv3_" in n or "igemm" in n:
            dur[key(n)[:60]][0] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6; dur[key(n)[:60]][1] += 1
per = collections.defaultdict(dict)
for (k, c), (s, n) in acc.items():
    per[k][c] = s / n
lines = ["# rocprofv3 --pmc (separate passes) on tools/conv3x3_experiment.py: averages per launch over the launches of each kernel",
         "# HBM bytes = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)"]
for k in sorted(per):
    d = per[k]
    t = dur[k][0] / max(dur[k][1], 1)
    line = "%-60s launches %3d avg %.3f ms" % (k, dur[k][1], t)
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        line += " | HBM read %.2f GB write %.2f GB" % (2 * d["FETCH_SIZE"] * 1024 / 1e9, d["WRITE_SIZE"] * 1024 / 1e9)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        cyc = d["GRBM_GUI_ACTIVE"] / 8.0
        line += " | MFMA busy %.1f %% at %.2f GHz" % (100 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), cyc / (t * 1e-3) / 1e9)
    if "SQ_LDS_BANK_CONFLICT" in d and "SQ_LDS_IDX_ACTIVE" in d:
        line += " | LDS conflicts %.1f %% of LDS active" % (100 * d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1))
    lines.append(line)
    lines.append("    raw: " + ", ".join("%s=%.4g" % (c, v) for c, v in sorted(d.items())))
open(dst, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
