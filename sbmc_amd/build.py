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
SOURCES = ["plain_ops.hip", "splat_fused.hip", "bias_act.hip", "pointwise.hip", "pointwise_chain.hip", "pointwise_chain_bwd.hip", "resample.hip", "nhwc_ops.hip", "halo.hip", "conv3x3.hip"]
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
    for path in [os.path.join(CSRC, src) for src in SOURCES] + _headers():
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


OBJ_DIR = os.path.join(HERE, ".obj")
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


_COMPILER_ID = None


def _compiler_id():
    """`hipcc --version` (a compiler upgrade must not link objects of the old one)."""
    global _COMPILER_ID
    if _COMPILER_ID is None:
        try:
            _COMPILER_ID = subprocess.check_output([_hipcc(), "--version"], stderr=subprocess.STDOUT).decode("utf-8", "replace")
        except (OSError, subprocess.CalledProcessError):
            _COMPILER_ID = "unknown"
    return _COMPILER_ID


def _headers():
    """Every header a source may include: all of csrc/*.hpp|*.h and the ABI header."""
    own = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h", ".inc")))
    return own + [os.path.join(ROOT, "include", "sbmc_hip.h")]


def _object_hash(src):
    """Hash of what ONE object file is built from: its source, every header of csrc/, the flags, the compiler."""
    import hashlib
    h = hashlib.sha256((" ".join(FLAGS) + "\0" + _compiler_id()).encode())
    for path in [os.path.join(CSRC, src)] + _headers():
        with open(path, "rb") as f:
            h.update(os.path.basename(path).encode() + b"\0" + f.read())
    return h.hexdigest()


def cached_remarks(src):
    """The resource remarks hipcc printed when .obj/<src>.o was built, if that object is the current source's; else None."""
    obj = os.path.join(OBJ_DIR, src + ".o")
    try:
        with open(obj + ".hash") as f:
            if f.read().strip() != _object_hash(src):
                return None
        with open(obj + ".remarks") as f:
            return f.read()
    except OSError:
        return None


def _compile(src, verbose):
    """csrc/<src> -> .obj/<src>.o unless an object of the same content hash is there (the objects are a local
    cache: only the linked library travels with a snapshot)."""
    obj = os.path.join(OBJ_DIR, src + ".o")
    stamp = obj + ".hash"
    digest = _object_hash(src)
    try:
        with open(stamp) as f:
            if f.read().strip() == digest and os.path.exists(obj) and os.path.exists(obj + ".remarks"):
                return obj
    except OSError:
        pass
    tmp = "%s.%d.tmp" % (obj, os.getpid())
    # (the compiler's per-kernel resource remarks -- registers, spills, scratch, occupancy -- are kept next to the object:
    # tools/kernel_resources.py and tests/test_kernel_resources.py read them instead of compiling the file once more)
    cmd = [_hipcc()] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", "-o", tmp, os.path.join(CSRC, src)]
    if verbose:
        print(" ".join(cmd), flush=True)
    try:
        res = subprocess.run(cmd, stderr=subprocess.PIPE)
        err = res.stderr.decode("utf-8", "replace")
        if res.returncode != 0:
            sys.stderr.write(err)
            raise subprocess.CalledProcessError(res.returncode, cmd)
        other = [ln for ln in err.splitlines() if "remark:" not in ln and "[-Rpass-analysis" not in ln]
        if any(ln.strip() for ln in other):
            sys.stderr.write("\n".join(other) + "\n")                  # (warnings: always shown)
        with open(tmp + ".remarks", "w") as f:
            f.write(err)
        os.replace(tmp + ".remarks", obj + ".remarks")
        os.replace(tmp, obj)
    finally:
        for t in (tmp, tmp + ".remarks"):
            if os.path.exists(t):
                os.remove(t)
    with open(tmp + ".hash", "w") as f:
        f.write(digest + "\n")
    os.replace(tmp + ".hash", stamp)
    return obj


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):          # the cached objects and their stamps only: another rank that is
            if f.endswith((".o", ".hash", ".remarks")):    # building right now owns the "<obj>.<pid>.tmp" files
                try:
                    os.remove(os.path.join(OBJ_DIR, f))
                except OSError:
                    pass
    digest = _source_hash()
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as pool:    # one hipcc per source
        objs = list(pool.map(lambda s: _compile(s, verbose), SOURCES))
    tmp = "%s.%d.tmp" % (LIB, os.getpid())              # several ranks may build at once: private temporaries,
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-fPIC", "-shared", "-fvisibility=hidden", "-o", tmp] + objs   # atomic renames
    if verbose:
        print(" ".join(cmd), flush=True)
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
