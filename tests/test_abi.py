"""The C-ABI library: loads, and exports every symbol include/sbmc_hip.h declares (no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sbmc_hip.h")).read()
    return sorted(set(re.findall(r"SBMC_API\s+[\w\s\*]+?\b(sbmc_\w+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for name in ("sbmc_scatter2gather_f32", "sbmc_kernel_weighting_fwd_f32",
                 "sbmc_kernel_weighting_bwd_f32", "sbmc_splat_update_fwd_f32",
                 "sbmc_splat_update_bwd_f32", "sbmc_hip_abi_version"):
        assert name in syms


def test_library_builds_and_exports_every_declared_symbol():
    from sbmc_amd import build, _lib
    path = build.build()  # hipcc cross-compiles gfx950 without a GPU
    assert os.path.exists(path)
    handle = ctypes.CDLL(path)
    for name in declared_symbols():
        assert hasattr(handle, name), "%s missing from %s" % (name, path)
    assert set(_lib.SYMBOLS) == set(declared_symbols())
    handle.sbmc_hip_abi_version.restype = ctypes.c_int
    assert handle.sbmc_hip_abi_version() == _lib.ABI_VERSION


def test_loader_fails_loudly_when_the_library_is_missing(monkeypatch, tmp_path):
    from sbmc_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HipExtensionMissing):
        _lib.lib()


def test_argument_validation_needs_no_gpu():
    """Bad dimensions are rejected before any launch (returns SBMC_HIP_EINVAL = -1)."""
    from sbmc_amd import _lib
    lib = _lib.lib()
    assert lib.sbmc_scatter2gather_f32(None, None, 1, 4, 4, 0, 3, None) == -1
    assert lib.sbmc_kernel_weighting_fwd_f32(None, None, None, None, 1, 3, 4, 4, 3, 3, None) == -1
    assert lib.sbmc_splat_update_fwd_f32(*([None] * 10), 1, 3, 8, 8, 4, None) == -1   # even k
    assert lib.sbmc_splat_update_fwd_f32(*([None] * 10), 1, 9, 8, 8, 3, None) == -1   # too many channels
    assert lib.sbmc_splat_update_supported(3, 21) == 1
    assert lib.sbmc_splat_update_supported(3, 4) == 0
    assert lib.sbmc_splat_update_supported(8, 21) == 0    # tile LDS budget exceeded at k=21, c=8
    assert lib.sbmc_splat_update_supported(8, 5) == 1
    assert lib.sbmc_splat_all_supported(3, 5, 64, 64) == 1 and lib.sbmc_splat_all_supported(4, 5, 64, 64) == 0
    assert lib.sbmc_splat_f16_supported(3, 21, 64, 64) == 1 and lib.sbmc_splat_f16_supported(3, 5, 64, 64) == 0
    assert lib.sbmc_splat_update_bwd_scratch_bytes(1, 3, 10, 10, 21) >= 10 * 10 * 4
    assert b"invalid" in lib.sbmc_hip_strerror(-1)
    # empty problems are a no-op, also without a device
    assert lib.sbmc_scatter2gather_f32(None, None, 0, 4, 4, 3, 3, None) == 0
    # fused 1x1 layers: at most 128 input channels, whole float4s, 32-bit byte offsets per batch element
    hw = 1280 * 720
    assert lib.sbmc_pointwise_supported(128, 441, hw) == 1 and lib.sbmc_pointwise_supported(93, 128, hw) == 1
    assert lib.sbmc_pointwise_supported(129, 128, hw) == 0 and lib.sbmc_pointwise_supported(128, 128, hw + 2) == 0
    assert lib.sbmc_pointwise_supported(128, 128, 3840 * 2160) == 1
    assert lib.sbmc_pointwise_supported(128, 128, 2 * 3840 * 2160) == 0
    assert lib.sbmc_pointwise_bwd_supported(128, 128, hw) == 1 and lib.sbmc_pointwise_bwd_supported(128, 441, hw) == 0
    assert lib.sbmc_pointwise_fwd_f32(*([None] * 5), 2, 1, 129, 128, 64, 0, 1, 0.0, None) == -1
    assert lib.sbmc_pointwise_fwd_f32(*([None] * 5), 3, 2, 128, 128, 64, 1, 1, 0.0, None) == -1     # b % s
    assert lib.sbmc_pointwise_bwd_f32(*([None] * 9), 1, 2, 1, 128, 441, 64, 0, 1, 0.0, None) == -1
    assert lib.sbmc_pointwise_fwd_f32(*([None] * 5), 0, 1, 128, 128, 64, 0, 1, 0.0, None) == 0
    assert lib.sbmc_bias_act_chunks(8, 128, hw) >= 1


def test_cpu_entry_points_refuse_without_a_registered_backend():
    import torch as th
    from sbmc_amd import halide_ops
    halide_ops.register_cpu_ops_for_testing(None)
    with pytest.raises(RuntimeError):
        halide_ops.kernel_weighting_cpu_float32(th.zeros(1, 1, 2, 2), th.zeros(1, 1, 1, 2, 2),
                                                th.zeros(1, 1, 2, 2), th.zeros(1, 2, 2))
    with pytest.raises(RuntimeError):
        halide_ops.kernel_weighting_cuda_float32(th.zeros(1, 1, 2, 2), th.zeros(1, 1, 1, 2, 2),
                                                 th.zeros(1, 1, 2, 2), th.zeros(1, 2, 2))


def test_product_never_imports_the_oracle():
    """Nothing under sbmc_amd/ may import or reference oracle/ (it is test infrastructure)."""
    import glob
    for path in glob.glob(os.path.join(ROOT, "sbmc_amd", "**", "*.py"), recursive=True):
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), path
        assert "sbmc_oracle" not in src, path
    for path in glob.glob(os.path.join(ROOT, "sbmc_amd", "csrc", "*")):
        assert "oracle" not in open(path).read(), path


def test_product_never_installs_a_host_implementation():
    """`halide_ops.register_cpu_ops_for_testing` is a test-harness seam: no product module (package or
    scripts) may call it -- with the oracle or with anything else -- so the `*_cpu_float32` names can
    only ever resolve inside a test or bench.py's cpu_baseline leg."""
    import glob
    files = glob.glob(os.path.join(ROOT, "sbmc_amd", "**", "*.py"), recursive=True)
    files += glob.glob(os.path.join(ROOT, "scripts", "*.py"))
    for path in files:
        src = open(path).read()
        uses = [m.start() for m in re.finditer(r"register_cpu_ops_for_testing|_CPU_OPS\s*=", src)]
        if path.endswith(os.path.join("sbmc_amd", "halide_ops.py")):
            # the definition, its docstring mention and the two assignments inside it: nothing else
            assert len(re.findall(r"register_cpu_ops_for_testing\s*\(", src)) == 1, path
            continue
        assert not uses, "%s touches the test-only host-op seam" % path


def test_halo_mailbox_geometry():
    """sbmc_halo_bytes is host arithmetic (no GPU): header + two rings of nslots slots; bad geometry -> 0."""
    from sbmc_amd import _lib
    L = _lib.lib()
    assert L.sbmc_halo_bytes(1 << 20, 4) == 4096 + 2 * 4 * (1 << 20)
    assert L.sbmc_halo_bytes(0, 4) == 0 and L.sbmc_halo_bytes(1 << 20, 0) == 0 and L.sbmc_halo_bytes(100, 2) == 0
