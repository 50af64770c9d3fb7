"""Per-category split of a rocprofv3 kernel_stats.csv of tools/rank_cost.py (5 steps per run: 2 warm-up + 3 timed).
    python tools/prof_rank_cat.py <r_kernel_stats.csv> [steps]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
CATS = [
    ("conv3x3 fwd/dgrad", r"sbmc::conv3_kernel|conv3_fixup"),
    ("conv3x3 wgrad", r"conv3_wgrad"),
    ("conv3x3 weight prep / absmax", r"prep_weights|absmax"),
    ("1x1 fwd", r"pw_fwd|pw_chain_fwd|pw_wide_fwd"),
    ("1x1 bwd", r"pw_bwd|pw_chain_bwd|pw_gw_wide|pw_wide_bwd|pw_wide_prep"),
    ("hipBLASLt / rocBLAS", r"Cijk_|rocblas|gemm"),
    ("MIOpen", r"igemm|miopen|naive_conv|batched_transpose"),
    ("splat", r"splat_|gather_|s2g_|kw_"),
    ("halo put/get/merge", r"halo::|halo_"),
    ("bias/act", r"bias_act|ctx_act"),
    ("resample/pool/transposes (own)", r"upcat|upsample|transpose2d|maxpool|pool|slice_channels"),
    ("weight bank", r"wbank_"),
    ("fill / memset", r"fillBuffer|FillFunctor|memset"),
    ("weight norm", r"weight_norm"),
    ("adam / optimizer", r"adam|multi_tensor|foreach"),
    ("copies", r"copyBuffer|CatArray|copy_|direct_copy|elementwise_kernel_manual_unroll"),
    ("rccl", r"ccl|AllReduce|SendRecv"),
]
agg = {}
for r in rows:
    name = r["Name"]
    for cat, pat in CATS:
        if re.search(pat, name):
            break
    else:
        cat = "other torch"
    a = agg.setdefault(cat, [0, 0])
    a[0] += int(r["TotalDurationNs"]); a[1] += int(r["Calls"])
tot = sum(a[0] for a in agg.values()); calls = sum(a[1] for a in agg.values())
print("kernel time per step: %.2f ms, %d launches per step" % (tot / steps / 1e6, calls / steps))
for cat, (ns, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("  %8.3f ms/step %7.1f launches/step  %s" % (ns / steps / 1e6, c / steps, cat))
print("top kernels:")
for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"]))[:45]:
    print("  %8.3f ms/step %7.1f /step %8.1f us avg  %s" % (int(r["TotalDurationNs"]) / steps / 1e6, int(r["Calls"]) / steps,
                                                  int(r["TotalDurationNs"]) / int(r["Calls"]) / 1e3, r["Name"][:110]))
