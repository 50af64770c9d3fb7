#!/bin/bash
# PMC counters of the fused 1x1-layer kernels (one counter group per run, kernel-trace only): HBM traffic, MFMA busy
# cycles, LDS bank conflicts, and where a wave's cycles go (waiting / issue-stalled / active).
#   bash tools/prof_pointwise.sh [tool]      tool: bench_pw_scaled.py (default: the round-5 two-plane kernels beside the
#   three-plane ones, and the 441-channel layer's one-pass backward) or "bench_pointwise.py --notest --time --bwd"
# Summary -> gpurun_out/profiles_pw/r02_pointwise_pmc.txt
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_pw
sum=$root/gpurun_out/profiles_pw
mkdir -p $out $sum
cd /tmp && export TMPDIR=/tmp
i=0
tool=${*:-bench_pw_scaled.py}
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $out/p$i -o p -- python $root/tools/$tool > $out/p$i.log 2>&1
done
python - $out $sum/r02_pointwise_pmc.txt <<'PY'
import collections, csv, glob, os, sys
root, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
dur = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(os.path.join(root, "p*", "*counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "sbmc::pw_" not in n:
            continue
        k = n.replace("void ", "").split("(")[0]
        acc[(k, row["Counter_Name"])][0] += float(row["Counter_Value"]); acc[(k, row["Counter_Name"])][1] += 1
for f in sorted(glob.glob(os.path.join(root, "p1", "*kernel_trace.csv"))):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "sbmc::pw_" in n:
            k = n.replace("void ", "").split("(")[0]
            dur[k][0] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6; dur[k][1] += 1
per = collections.defaultdict(dict)
for (k, c), (s, n) in acc.items():
    per[k][c] = s / n
lines = ["# rocprofv3 --pmc (separate passes) on the tool's launches: 8 x [96 / 128 -> 128 / 441, 1280x720] layers",
         "# HBM bytes = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024 (gfx950: FETCH_SIZE counts 128-B requests at 64 B)",
         "# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); clock = GRBM_GUI_ACTIVE / 8 / duration"]
for k in sorted(per):
    d = per[k]
    line = "%-44s launches %3d avg %.3f ms" % (k, dur[k][1], dur[k][0] / max(dur[k][1], 1))
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        line += " | HBM read %.2f GB write %.2f GB" % (2 * d["FETCH_SIZE"] * 1024 / 1e9, d["WRITE_SIZE"] * 1024 / 1e9)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        cyc = d["GRBM_GUI_ACTIVE"] / 8.0          # the counter is summed over the 8 XCDs
        line += " | MFMA busy %.1f %% of cycles at %.2f GHz" % (100 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
                                                              cyc / (dur[k][0] / max(dur[k][1], 1) * 1e-3) / 1e9)
    if "SQ_LDS_BANK_CONFLICT" in d and "SQ_LDS_IDX_ACTIVE" in d:
        line += " | LDS bank-conflict cycles %.1f %% of LDS active" % (100 * d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1))
    if "SQ_WAIT_ANY" in d and "SQ_WAVE_CYCLES" in d:
        wc = d["SQ_WAVE_CYCLES"]
        line += " | of a wave's cycles: waiting (s_waitcnt / barrier) %.0f %%, issue-stalled %.0f %%, issuing %.0f %% (VALU %.0f %%, LDS %.0f %%)" % (
            100 * d["SQ_WAIT_ANY"] / wc, 100 * d.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * d["SQ_ACTIVE_INST_ANY"] / wc,
            100 * d["SQ_ACTIVE_INST_VALU"] / wc, 100 * d["SQ_ACTIVE_INST_LDS"] / wc)
    lines.append(line)
    lines.append("    raw: " + ", ".join("%s=%.4g" % (c, v) for c, v in sorted(d.items())))
open(dst, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
