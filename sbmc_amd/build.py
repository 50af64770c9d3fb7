"""Builds libsbmc_hip.so (the C-ABI library declared in include/sbmc_hip.h).

    python -m sbmc_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The library is built IN-TREE
(sbmc_amd/libsbmc_hip.so) so that it travels with the source snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libsbmc_hip.so")
SOURCES = ["plain_ops.hip", "splat_fused.hip", "bias_act.hip", "pointwise.hip", "resample.hip", "nhwc_ops.hip"]
DEPS = SOURCES + ["common.hpp", os.path.join(ROOT, "include", "sbmc_hip.h")]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for d in DEPS:
        path = d if os.path.isabs(d) else os.path.join(CSRC, d)
        if os.path.getmtime(path) > t:
            return True
    return False


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
           "-o", LIB + ".tmp"] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
