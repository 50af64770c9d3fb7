#!/bin/bash
# What do FETCH_SIZE / WRITE_SIZE report for streams of KNOWN size?  (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies the
# 128-byte requests of 16-B/lane streams at 64 bytes -- "other access widths are uncalibrated: calibrate on a known byte
# count in your own access pattern".)  tools/stream_ceiling.hip moves a [441, 720, 1280] fp32 tensor (1.626 GB) as a float4
# copy (V0) and in the splat kernels' pattern, DWORD buffer loads / stores of 256-byte row segments (V1-V3).
#   tools/grun "bash tools/calibrate_fetch.sh"  ->  gpurun_out/fetch_calibration.txt
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 $root/tools/stream_ceiling.hip -o /tmp/sc || exit 1
for pmc in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d /tmp/cal_$pmc -o c -- /tmp/sc > /tmp/cal_$pmc.log 2>&1
done
python - <<'PY' | tee $root/gpurun_out/fetch_calibration.txt
import collections, csv, glob
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("/tmp/cal_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "").split("(")[0]
        acc[(k, row["Counter_Name"])][0] += float(row["Counter_Value"]); acc[(k, row["Counter_Name"])][1] += 1
true = 441 * 720 * 1280 * 4 / 1e9
print("# tools/calibrate_fetch.sh: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) on tools/stream_ceiling.hip; every kernel")
print("# reads the [441, 720, 1280] fp32 tensor once = %.3f GB (and the copies write as much)" % true)
ks = sorted(set(k for k, _ in acc))
for k in ks:
    f = acc.get((k, "FETCH_SIZE"), [0, 1]); w = acc.get((k, "WRITE_SIZE"), [0, 1])
    fg, wg = f[0] / max(f[1], 1) * 1024 / 1e9, w[0] / max(w[1], 1) * 1024 / 1e9
    print("%-34s FETCH_SIZE %.3f GB = %.3f of the bytes read (factor to apply: %.2f) | WRITE_SIZE %.3f GB = %.3f of the bytes written" % (
        k, fg, fg / true, true / fg if fg else 0, wg, wg / true))
PY
