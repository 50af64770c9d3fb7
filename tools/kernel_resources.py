"""Register / scratch / occupancy report of every gfx950 kernel in sbmc_amd/csrc (hipcc
-Rpass-analysis=kernel-resource-usage; cross-compiles without a GPU).

    python tools/kernel_resources.py [--all] [file.hip ...]

Prints the kernels that SPILL (scratch > 0) -- with --all, every kernel.  A hot kernel that starts
spilling after an edit loses a third of its speed without failing any test (round 2: the forward strip
kernel went from 66 VGPRs / no scratch to 72 + 11 spilled through an innocent-looking early `continue`);
tests/test_kernel_resources.py keeps the hot kernels spill-free.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sbmc_amd", "csrc")


def _demangle(names):
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not tool:
        return {n: n for n in names}
    out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def resources(path):
    """-> list of dicts {name, vgprs, agprs, spill, scratch, occupancy, lds} for the kernels of one .hip file."""
    text = None
    if os.path.dirname(os.path.abspath(path)) == CSRC and not os.environ.get("SBMC_KERNEL_RESOURCES_COMPILE"):
        # the library's own build keeps the remarks of every object it compiles (sbmc_amd/build.py: same flags, same
        # compiler, keyed on the source + header hash): no second three-minute compile of pointwise.hip
        sys.path.insert(0, ROOT)
        try:
            from sbmc_amd import build as _build
            text = _build.cached_remarks(os.path.basename(path))
        except Exception:      # noqa: BLE001 -- any trouble with the cache: compile
            text = None
        finally:
            sys.path.pop(0)
    if text is None:
        hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        with tempfile.TemporaryDirectory() as tmp:
            res = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-c", path,
                                  "-o", os.path.join(tmp, "o.o"), "-Rpass-analysis=kernel-resource-usage"],
                                 capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(res.stderr[-2000:])
        text = res.stderr
    blocks = re.split(r"remark: Function Name: ", text)[1:]
    names = [b.split(" ")[0] for b in blocks]
    dem = _demangle(names)
    out = []
    for name, b in zip(names, blocks):
        def g(key):
            m = re.search(key + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        out.append(dict(name=dem[name], vgprs=g("VGPRs"), agprs=g("AGPRs"), spill=g("VGPRs Spill"),
                        scratch=g(r"ScratchSize \[bytes/lane\]"), occupancy=g(r"Occupancy \[waves/SIMD\]"),
                        lds=g(r"LDS Size \[bytes/block\]")))
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    files = args or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    for f in files:
        rows = resources(f)
        bad = [r for r in rows if r["scratch"] > 0 or r["spill"] > 0]
        print("%s: %d kernels, %d spilling" % (os.path.basename(f), len(rows), len(bad)))
        for r in (rows if "--all" in sys.argv else bad):
            print("   vgpr %3d agpr %3d spill %3d scratch %4d B occ %d  %s" % (
                r["vgprs"], r["agprs"], r["spill"], r["scratch"], r["occupancy"], r["name"][:120]))
