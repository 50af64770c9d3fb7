#!/bin/bash
# MFMA utilisation of the whole conv stack (MIOpen 3x3 convolutions, own 1x1 kernels, hipBLASLt GEMMs) over
# the training step: one PMC pass (kernel-trace only) + one timing pass of the same command.
#   gpurun -- bash tools/prof_conv_stack.sh     ->  gpurun_out/conv_stack/r02_conv_stack_mfma.txt
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/conv_stack
mkdir -p $out
cmd="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-stages --no-cpu-baseline"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/pmc -o p -- $cmd > $out/pmc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/st -o s -- $cmd > $out/st.log 2>&1
python - <<'PY'
import collections, csv, glob, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/conv_stack"
def fam(n):
    if n.startswith("igemm_fwd"): return "MIOpen igemm fwd (3x3, NHWC fp32)"
    if n.startswith("igemm_bwd"): return "MIOpen igemm bwd-data"
    if n.startswith("igemm_wrw"): return "MIOpen igemm wrw"
    if "pw_fwd" in n: return "sbmc pw_fwd (1x1)"
    if "pw_bwd" in n: return "sbmc pw_bwd (1x1)"
    if n.startswith("Cijk"): return "hipBLASLt GEMMs (441-channel backward, context products)"
    return None
acc = collections.defaultdict(lambda: collections.defaultdict(float))
f = glob.glob(out + "/pmc/**/p_counter_collection.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    k = fam(r["Kernel_Name"])
    if k:
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
dur = collections.defaultdict(float); calls = collections.defaultdict(int); total = 0.0
steps = 0.0
s = glob.glob(out + "/st/**/s_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(s)):
    total += float(r["TotalDurationNs"])
    if "splat_bwd_strip_kernel" in r["Name"]:
        steps += int(r["Calls"])             # one all-samples backward launch per training step
    k = fam(r["Name"])
    if k:
        dur[k] += float(r["TotalDurationNs"]); calls[k] += int(r["Calls"])
with open(out + "/r02_conv_stack_mfma.txt", "w") as o:
    o.write("# MFMA utilisation of the conv stack over the 1280x720x8spp training step (bench.py --steps 2 --warmup 2,\n"
            "# %d steps profiled; rocprofv3 --pmc pass + a separate --kernel-trace --stats pass of the same command)\n"
            "# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): fraction of SIMD-cycles with the\n"
            "# matrix pipe busy while the kernel runs (fp32 MFMA peak 157 TFLOP/s = 100 %%)\n" % steps)
    tb = tg = 0.0
    for k in sorted(acc, key=lambda k: -dur[k]):
        a = acc[k]
        busy = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        tb += a["SQ_VALU_MFMA_BUSY_CYCLES"]; tg += a["GRBM_GUI_ACTIVE"]
        o.write("%-58s %7.1f ms/step (%4.1f %% of kernel time, %4d launches/step)  MFMA busy %5.1f %%\n" % (
            k, dur[k] / steps / 1e6, 100 * dur[k] / total, calls[k] / steps, 100 * busy))
    o.write("%-58s %7.1f ms/step (%4.1f %% of kernel time)                        MFMA busy %5.1f %%\n" % (
        "conv stack, all of the above", sum(dur.values()) / steps / 1e6, 100 * sum(dur.values()) / total,
        100 * tb / (tg / 8.0 * 1024.0)))
    o.write("all kernels of a step: %.1f ms\n" % (total / steps / 1e6))
print(open(out + "/r02_conv_stack_mfma.txt").read())
PY
