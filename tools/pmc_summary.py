"""Condenses rocprofv3 output (tools/prof.sh) into small committed summaries.

  <tag>_splat_kernel_stats.csv / <tag>_model_kernel_stats.csv : rocprofv3 --stats tables with
        kernel names shortened (the dominant kernels keep their full name)
  <tag>_pmc.txt  : per (kernel, counter) average over dispatches
  <tag>_pmc.json : HBM traffic per launch of the sbmc kernels, corrected as
        MI355X_MICROARCH.md prescribes: bytes = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024
        (on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes)
"""
import collections
import csv
import glob
import json
import os
import sys

root, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def short(name, n=150):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[:n] + "..."


for wl in ("splat", "model"):
    files = glob.glob(os.path.join(root, wl, "*kernel_stats.csv"))
    if not files:
        continue
    with open(files[0]) as fh, open(os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, wl)), "w") as oh:
        rd = csv.DictReader(fh)
        wr = csv.writer(oh)
        wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for row in rd:
            wr.writerow([short(row["Name"]), row["Calls"], row["TotalDurationNs"], row["AverageNs"],
                         row["Percentage"], row["MinNs"], row["MaxNs"], row["StdDev"]])

acc = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(os.path.join(root, "p*", "*counter_collection.csv"))):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            if "sbmc" not in name:
                continue
            key = (short(name.split("(")[0]), row["Counter_Name"])
            acc[key][0] += float(row["Counter_Value"])
            acc[key][1] += 1
lines = []
per_kernel = collections.defaultdict(dict)
for (k, c), (s, n) in sorted(acc.items()):
    lines.append("%-48s %-22s avg=%.6g n=%d" % (k, c, s / n, n))
    per_kernel[k][c] = s / n
open(os.path.join(out, "%s_pmc.txt" % tag), "w").write("\n".join(lines) + "\n")
traffic = {}
for k, d in per_kernel.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        traffic[k] = {"FETCH_SIZE_KB": d["FETCH_SIZE"], "WRITE_SIZE_KB": d["WRITE_SIZE"],
                      "hbm_bytes_per_launch": int(2 * d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024)}
# which tree and when (the GPU box has no .git: the launching side leaves the commit in .commit_for_profiles)
import datetime
commit = None
for cand in (os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), ".commit_for_profiles"),):
    if os.path.exists(cand):
        commit = open(cand).read().strip()
# what the profiled kernels were built from: bench.py quotes these figures only while the sources are the same
import hashlib
repo = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
src_hash = hashlib.sha256()
for f in ("sbmc_amd/csrc/splat_fused.hip", "sbmc_amd/csrc/common.hpp"):
    src_hash.update(open(os.path.join(repo, f), "rb").read())
traffic["_meta"] = {"kernel_source_sha256": src_hash.hexdigest(), "commit": commit, "date": datetime.datetime.utcnow().strftime("%Y-%m-%d %H:%M UTC"),
                    "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- python bench.py "
                               "--workload splat --steps 1 --warmup 1 (tools/prof.sh)",
                    "correction": "bytes = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024 (gfx950: FETCH_SIZE counts 128-byte "
                                  "requests at 64 bytes, MI355X_MICROARCH.md)"}
json.dump(traffic, open(os.path.join(out, "%s_pmc.json" % tag), "w"), indent=1, sort_keys=True)
print("\n".join(lines[:0]))
