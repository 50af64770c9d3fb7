"""MIOpen find results for the U-nets' convolutions, shipped with the package.

Since round 3 the fp32 3x3 convolutions run on the package's own kernels (csrc/conv3x3.hip) and MIOpen convolves only
what those do not take: half activations under torch.autocast(float16), channel counts that are not multiples of
32 / 128, or everything with SBMC_CONV3X3=0.  For that path:

The fastest fp32 3x3 solvers MIOpen has on gfx950 are NHWC-native implicit-GEMM kernels.  PyTorch picks
a solver through MIOpen's "find"; this package runs MIOpen in its FAST find mode (a full find costs minutes
on the first step), which takes a find-db record when there is one and a heuristic otherwise -- and the
heuristic's choice for channels-last fp32 tensors is up to 50x off (a grouped-convolution kernel for the
weight gradient: 124 ms instead of 2.4 ms at 1280x720).  The records in ``miopen_db/*.ufdb.txt`` are MIOpen's
own find results for the convolution shapes of ``Multisteps`` (``tools/make_miopen_db.py`` produced them on
an MI355X with MIOPEN_FIND_MODE=1); installed as MIOpen's *user* find-db they make the FAST mode pick the
measured-best solver at once.  The file name carries the GPU (gfx950, 256 CUs) and the MIOpen build: on any
other combination MIOpen simply does not see the records, and ``modules.unet_channels_last`` -- which
*measures* both layouts before it commits to one -- keeps the U-nets in NCHW.
"""
import atexit
import glob
import os
import shutil
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
DB_DIR = os.path.join(HERE, "miopen_db")
_INSTALLED = None


def install():
    """Points MIOpen's user find-db at a private scratch copy of the shipped records and defaults the find mode
    to FAST.  Must run before the first convolution; called at package import.  Returns the directory ("":
    nothing installed).  SBMC_MIOPEN_DB=0 switches this off altogether; if the user has set
    MIOPEN_USER_DB_PATH the records are merged into that directory only with SBMC_MIOPEN_DB=merge."""
    global _INSTALLED
    if _INSTALLED is not None:
        return _INSTALLED
    choice = os.environ.get("SBMC_MIOPEN_DB", "1").lower()
    if choice in ("0", "off", "no", "false"):
        # opt-out: MIOpen's environment is left exactly as the user set it up (the U-nets then stay planar
        # unless MIOpen's own find-db knows the channels-last shapes: modules.unet_channels_last measures)
        _INSTALLED = ""
        return _INSTALLED
    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
    target = os.environ.get("MIOPEN_USER_DB_PATH")
    if target and choice != "merge":
        # the user chose a find-db directory: it is theirs -- nothing is written into it unless they ask for
        # it with SBMC_MIOPEN_DB=merge
        _INSTALLED = ""
        return _INSTALLED
    if not target:
        # a private scratch copy per process (MIOpen appends what it learns to its user db; the shipped
        # records stay read-only), removed at exit
        target = tempfile.mkdtemp(prefix="sbmc_amd_miopen_db_")
        atexit.register(shutil.rmtree, target, True)
        os.environ["MIOPEN_USER_DB_PATH"] = target
    try:
        os.makedirs(target, exist_ok=True)
        for src in glob.glob(os.path.join(DB_DIR, "*.ufdb.txt")):
            dst = os.path.join(target, os.path.basename(src))
            _merge(src, dst)
    except OSError:
        pass            # read-only location: MIOpen then works without the records (NCHW stays)
    _INSTALLED = target
    return target


def _merge(src, dst):
    """Adds the records of `src` whose key `dst` does not hold yet (a find-db is one `key=value` per line)."""
    if not os.path.exists(dst):
        shutil.copyfile(src, dst)
        return
    have = set()
    with open(dst) as f:
        for line in f:
            have.add(line.split("=", 1)[0])
    new = []
    with open(src) as f:
        for line in f:
            if line.strip() and line.split("=", 1)[0] not in have:
                new.append(line if line.endswith("\n") else line + "\n")
    if new:
        with open(dst, "a") as f:
            f.writelines(new)
