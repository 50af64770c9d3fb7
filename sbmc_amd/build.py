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
SOURCES = ["plain_ops.hip", "splat_fused.hip", "bias_act.hip", "pointwise.hip", "resample.hip", "nhwc_ops.hip", "halo.hip", "conv3x3.hip"]
DEPS = SOURCES + ["common.hpp", os.path.join(ROOT, "include", "sbmc_hip.h")]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


STAMP = LIB + ".srchash"


def _source_hash():
    """Hash of everything the library is built from (sources, headers, flags)."""
    import hashlib
    h = hashlib.sha256((ARCH + " -O3 -std=c++17").encode())
    for d in DEPS:
        path = d if os.path.isabs(d) else os.path.join(CSRC, d)
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    return h.hexdigest()


def is_stale():
    """Is the library missing or built from other sources than the ones present?  By CONTENT (a hash of
    the sources recorded next to the library at build time), not by modification time: the library travels
    to GPU boxes inside source snapshots whose file times need not survive the copy, and a spurious rebuild
    there would run once per rank."""
    if not os.path.exists(LIB):
        return True
    try:
        with open(STAMP) as f:
            return f.read().strip() != _source_hash()
    except OSError:
        return True


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    tmp = "%s.%d.tmp" % (LIB, os.getpid())              # several ranks may build at once: private temporaries,
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared",   # atomic renames
           "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
           "-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    digest = _source_hash()
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    with open(tmp + ".stamp", "w") as f:
        f.write(digest + "\n")
    os.replace(tmp + ".stamp", STAMP)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
