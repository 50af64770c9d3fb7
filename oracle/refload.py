"""Load the hot-path part of the *reference* Python package on top of the oracle.

TEST INFRASTRUCTURE, AUTHORING CONTAINER ONLY: ``/root/reference`` does not
exist on the GPU box, so nothing that runs there may call this.  It is used by
``oracle/pin_against_reference.py`` (runs the reference's own unit tests against
the oracle) and ``tests/golden/make_golden.py`` (emits the committed fixtures).

The reference's ``sbmc/__init__.py`` pulls in lz4 / pyexr / skimage / ttools,
none of which exist in this image, so the four hot-path
files are loaded one by one into a synthetic ``sbmc`` package with

* ``sbmc.halide_ops``            := ``oracle.sbmc_oracle`` (the six op names)
* ``ttools.get_logger``          := ``logging.getLogger``
* ``ttools.modules.image_operators.crop_like`` := centre-crop of the last two
  dims to the target's size (torch-tools 0.0.36 is not in the tree; in
  ``Multisteps`` the call is an identity because every regressor conv is 1x1,
  models.py:98-102,206 -- parity of crop_like itself is unpinned).

No reference source is copied: the files are executed from where they lie.
"""
import importlib.util
import logging
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("SBMC_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "sbmc", "functions.py"))


def _crop_like(src, tgt):
    sh, sw = src.shape[-2:]
    th_, tw = tgt.shape[-2:]
    dy, dx = (sh - th_) // 2, (sw - tw) // 2
    if dy == 0 and dx == 0:
        return src
    return src[..., dy:dy + th_, dx:dx + tw]


def _install_ttools_stub():
    if "ttools" in sys.modules and getattr(sys.modules["ttools"], "_sbmc_stub", False):
        return
    tt = types.ModuleType("ttools")
    tt._sbmc_stub = True
    tt.get_logger = logging.getLogger
    tt.ModelInterface = type("ModelInterface", (object,), {})   # base class of sbmc/interfaces.py:35 (not in tree)
    mods = types.ModuleType("ttools.modules")
    imops = types.ModuleType("ttools.modules.image_operators")
    imops.crop_like = _crop_like
    mods.image_operators = imops
    tt.modules = mods
    sys.modules["ttools"] = tt
    sys.modules["ttools.modules"] = mods
    sys.modules["ttools.modules.image_operators"] = imops


def load_reference(ops_module=None):
    """Returns the synthetic ``sbmc`` package (functions, modules, losses, models)."""
    if not available():
        raise RuntimeError("reference tree not found at %s" % REFERENCE_ROOT)
    if ops_module is None:
        from oracle import sbmc_oracle as ops_module
    _install_ttools_stub()
    for name in [m for m in sys.modules if m == "sbmc" or m.startswith("sbmc.")]:
        del sys.modules[name]
    pkg = types.ModuleType("sbmc")
    pkg.__path__ = []  # mark as package
    sys.modules["sbmc"] = pkg
    sys.modules["sbmc.halide_ops"] = ops_module
    pkg.halide_ops = ops_module
    for sub in ("functions", "modules", "losses", "models"):
        path = os.path.join(REFERENCE_ROOT, "sbmc", sub + ".py")
        spec = importlib.util.spec_from_file_location("sbmc." + sub, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules["sbmc." + sub] = mod
        spec.loader.exec_module(mod)
        setattr(pkg, sub, mod)
    # reference models.py uses an undefined LOG in its error paths (models.py:62,66)
    pkg.Multisteps = pkg.models.Multisteps
    pkg.KPCN = pkg.models.KPCN
    pkg.losses_mod = pkg.losses
    return pkg


def load_reference_interfaces(pkg):
    """Adds the reference's sbmc/interfaces.py (training step: loss, clip, Adam) to `pkg`."""
    path = os.path.join(REFERENCE_ROOT, "sbmc", "interfaces.py")
    spec = importlib.util.spec_from_file_location("sbmc.interfaces", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["sbmc.interfaces"] = mod
    spec.loader.exec_module(mod)
    pkg.interfaces = mod
    return mod
