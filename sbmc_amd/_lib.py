"""ctypes binding of libsbmc_hip.so (C ABI: include/sbmc_hip.h).

The library must exist: there is no fallback of any kind.  If it is missing or
does not export every symbol of the header, importing this module's ``lib()``
raises -- loudly -- instead of degrading to a slow path.
"""
import ctypes
import os
import warnings

import torch  # noqa: F401  (must be loaded first: brings in the HIP runtime the library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT_LIB_PATH = os.path.join(_HERE, "libsbmc_hip.so")
# SBMC_HIP_LIB: alternative build of the same ABI (kernel A/B experiments)
LIB_PATH = os.environ.get("SBMC_HIP_LIB") or _DEFAULT_LIB_PATH

#: every extern "C" symbol include/sbmc_hip.h declares
SYMBOLS = (
    "sbmc_hip_abi_version",
    "sbmc_hip_strerror",
    "sbmc_scatter2gather_f32",
    "sbmc_kernel_weighting_fwd_f32",
    "sbmc_kernel_weighting_bwd_f32",
    "sbmc_scatter2gather_f16",
    "sbmc_kernel_weighting_fwd_f16",
    "sbmc_kernel_weighting_bwd_f16",
    "sbmc_splat_update_supported",
    "sbmc_splat_update_bwd_scratch_bytes",
    "sbmc_splat_update_fwd_f32",
    "sbmc_splat_update_bwd_f32",
    "sbmc_splat_all_supported",
    "sbmc_splat_merge_fwd_f32",
    "sbmc_splat_all_bwd_f32",
    "sbmc_splat_slab_supported",
    "sbmc_splat_slab_fwd_f32",
    "sbmc_splat_slab_bwd_f32",
    "sbmc_splat_slab_fwd_f16",
    "sbmc_splat_slab_bwd_f16",
    "sbmc_splat_f16_supported",
    "sbmc_splat_update_fwd_f16",
    "sbmc_splat_update_bwd_f16",
    "sbmc_splat_all_bwd_f16",
    "sbmc_gather_update_supported",
    "sbmc_gather_update_fwd_f32",
    "sbmc_gather_update_bwd_f32",
    "sbmc_bias_act_chunks",
    "sbmc_bias_act_fwd_f32",
    "sbmc_bias_act_bwd_f32",
    "sbmc_ctx_act_fwd_f32",
    "sbmc_ctx_act_bwd_f32",
    "sbmc_pointwise_supported",
    "sbmc_pointwise_fwd_f32",
    "sbmc_pointwise_fwd_f16",
    "sbmc_pointwise_bwd_supported",
    "sbmc_pointwise_bwd_groups",
    "sbmc_pointwise_bwd_f32",
    "sbmc_pointwise_gw_wide_supported",
    "sbmc_pointwise_gw_wide_groups",
    "sbmc_pointwise_gw_wide_f32",
    "sbmc_pointwise_gw_wide_f16",
    "sbmc_pointwise_bwd_f16",
    "sbmc_pointwise_fwd_signs_f32",
    "sbmc_pointwise_bwd_signs_f32",
    "sbmc_pointwise_fwd_mean_f32",
    "sbmc_pointwise_fwd_mean_f16",
    "sbmc_pointwise_fwd_scaled_f32",
    "sbmc_pointwise_bwd_scaled_f32",
    "sbmc_pointwise_wide_bwd_ws_bytes",
    "sbmc_pointwise_wide_bwd_f32",
    "sbmc_pointwise_chain_supported",
    "sbmc_pointwise_chain_fwd_f32",
    "sbmc_pointwise_wide_fwd_supported",
    "sbmc_pointwise_wide_fwd_f32",
    "sbmc_pointwise_chain_bwd_supported",
    "sbmc_pointwise_chain_bwd_groups",
    "sbmc_pointwise_chain_bwd_f32",
    "sbmc_splat_all_bwd_bound_f32",
    "sbmc_upsample2x_cat_supported",
    "sbmc_upsample2x_cat_fwd_f32",
    "sbmc_upsample2x_cat_bwd_f32",
    "sbmc_upsample2x_cat_slab_fwd_f32",
    "sbmc_upsample2x_cat_slab_bwd_f32",
    "sbmc_transpose2d_f32",
    "sbmc_transpose2d_f16",
    "sbmc_maxpool2_nhwc_fwd",
    "sbmc_maxpool2_nhwc_bwd_add",
    "sbmc_maxpool2_nhwc_bwd_add_adj_f32",
    "sbmc_bias_act_nhwc_supported",
    "sbmc_bias_act_nhwc_chunks",
    "sbmc_bias_act_nhwc_fwd_f32",
    "sbmc_bias_act_nhwc_bwd_f32",
    "sbmc_bias_act_nhwc_fwd_signs_f32",
    "sbmc_bias_act_nhwc_bwd_signs_f32",
    "sbmc_upsample2x_cat_nhwc_supported",
    "sbmc_upsample2x_cat_nhwc_fwd_f32",
    "sbmc_upsample2x_cat_nhwc_bwd_f32",
    "sbmc_upsample2x_cat_nhwc_slab_fwd_f32",
    "sbmc_upsample2x_cat_nhwc_slab_bwd_f32",
    "sbmc_upsample2x_cat_nhwc_bwd_adj_f32",
    "sbmc_upsample2x_cat_nhwc_slab_fwd_f16",
    "sbmc_upsample2x_cat_nhwc_slab_bwd_f16",
    "sbmc_halo_bytes",
    "sbmc_halo_alloc",
    "sbmc_halo_free",
    "sbmc_halo_open",
    "sbmc_halo_close",
    "sbmc_halo_status",
    "sbmc_halo_status_to",
    "sbmc_halo_put",
    "sbmc_halo_get",
    "sbmc_halo_merge_state_fwd_f32",
    "sbmc_halo_merge_state_bwd_f32",
    "sbmc_transpose2d_amax_f32",
    "sbmc_bias_act_nhwc_fwd_amax_f32",
    "sbmc_bias_act_nhwc_bwd_amax_f32",
    "sbmc_conv3x3_supported",
    "sbmc_conv3x3_weights_bytes",
    "sbmc_conv3x3_absmax_f32",
    "sbmc_conv3x3_absmax_raise_f32",
    "sbmc_conv3x3_prepare_weights_f32",
    "sbmc_conv3x3_workspace_bytes",
    "sbmc_conv3x3_nhwc_f32",
    "sbmc_conv3x3_bias_act_nhwc_f32",
    "sbmc_conv3x3_adj_supported",
    "sbmc_conv3x3_adj_partial_rows",
    "sbmc_conv3x3_adj_nhwc_f32",
    "sbmc_conv3x3_wgrad_supported",
    "sbmc_conv3x3_wgrad_scratch_bytes",
    "sbmc_conv3x3_wgrad_f32",
    "sbmc_conv3x3_wgrad_bias_f32",
    "sbmc_conv3x3_nhwc_f16",
    "sbmc_conv3x3_bias_act_nhwc_f16",
    "sbmc_conv3x3_wgrad_bias_f16",
    "sbmc_bias_act_nhwc_bwd_signs_f16",
    "sbmc_wbank_forward_f32",
    "sbmc_wbank_backward_f32",
)
ABI_VERSION = 8
WBANK_MAX = 24


class WBankEntry(ctypes.Structure):
    """include/sbmc_hip.h: sbmc_wbank_entry"""
    _fields_ = [("v", ctypes.c_void_p), ("g", ctypes.c_void_p), ("w", ctypes.c_void_p), ("norm", ctypes.c_void_p),
                ("wp_fwd", ctypes.c_void_p), ("wp_bwd", ctypes.c_void_p),
                ("cout", ctypes.c_int), ("cin", ctypes.c_int), ("kh", ctypes.c_int), ("kw", ctypes.c_int)]


class WBankGrad(ctypes.Structure):
    """include/sbmc_hip.h: sbmc_wbank_grad"""
    _fields_ = [("gw", ctypes.c_void_p), ("v", ctypes.c_void_p), ("g", ctypes.c_void_p), ("norm", ctypes.c_void_p),
                ("gv", ctypes.c_void_p), ("gg", ctypes.c_void_p),
                ("s_co", ctypes.c_long), ("s_ci", ctypes.c_long), ("s_ky", ctypes.c_long), ("s_kx", ctypes.c_long),
                ("cout", ctypes.c_int), ("cin", ctypes.c_int), ("kh", ctypes.c_int), ("kw", ctypes.c_int)]
MAX_CHANNELS = 8

_LIB = None


class HipExtensionMissing(RuntimeError):
    pass


def lib():
    """Loads (once) and returns the ctypes handle; raises if unavailable."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if LIB_PATH == _DEFAULT_LIB_PATH:
        # not built yet (fresh checkout: *.so is git-ignored) or older than its sources: build the
        # real thing with hipcc.  This is a build step, not a fallback -- a compile error is the
        # error the user sees.
        from . import build as _build
        try:
            stale = _build.is_stale()
        except OSError:
            stale = not os.path.exists(LIB_PATH)     # installed without sources: take the .so as is
        if stale:
            try:
                _build.build()
            except Exception as e:
                if not os.path.exists(LIB_PATH):
                    raise HipExtensionMissing(
                        "%s is not built and building it failed (%s: %s). Build it with `python -m "
                        "sbmc_amd.build` (hipcc, gfx950); sbmc_amd has no CPU or PyTorch fallback for "
                        "its operators." % (LIB_PATH, type(e).__name__, e)) from e
                warnings.warn("%s is older than its sources and rebuilding it failed (%s: %s); "
                              "loading the stale library" % (LIB_PATH, type(e).__name__, e))
    if not os.path.exists(LIB_PATH):
        raise HipExtensionMissing(
            "%s not found: build it with `python -m sbmc_amd.build` (hipcc, gfx950). "
            "sbmc_amd has no CPU or PyTorch fallback for its operators." % LIB_PATH)
    handle = ctypes.CDLL(LIB_PATH)
    for name in SYMBOLS:
        if not hasattr(handle, name):
            raise HipExtensionMissing("%s does not export %s" % (LIB_PATH, name))
    p, i = ctypes.c_void_p, ctypes.c_int
    handle.sbmc_hip_abi_version.restype = i
    handle.sbmc_hip_strerror.restype = ctypes.c_char_p
    handle.sbmc_hip_strerror.argtypes = [i]
    handle.sbmc_scatter2gather_f32.argtypes = [p, p, i, i, i, i, i, p]
    handle.sbmc_kernel_weighting_fwd_f32.argtypes = [p] * 4 + [i] * 6 + [p]
    handle.sbmc_kernel_weighting_bwd_f32.argtypes = [p] * 7 + [i] * 6 + [p]
    handle.sbmc_scatter2gather_f16.argtypes = handle.sbmc_scatter2gather_f32.argtypes
    handle.sbmc_kernel_weighting_fwd_f16.argtypes = handle.sbmc_kernel_weighting_fwd_f32.argtypes
    handle.sbmc_kernel_weighting_bwd_f16.argtypes = handle.sbmc_kernel_weighting_bwd_f32.argtypes
    handle.sbmc_splat_update_supported.argtypes = [i, i]
    handle.sbmc_splat_update_fwd_f32.argtypes = [p] * 10 + [i] * 5 + [p]
    handle.sbmc_splat_update_bwd_f32.argtypes = [p] * 19 + [i] * 5 + [p]
    handle.sbmc_splat_update_bwd_scratch_bytes.argtypes = [i] * 5
    handle.sbmc_splat_all_supported.argtypes = [i] * 4
    handle.sbmc_splat_f16_supported.argtypes = [i] * 4
    handle.sbmc_splat_merge_fwd_f32.argtypes = [p] * 9 + [i] * 5 + [p]
    handle.sbmc_splat_all_bwd_f32.argtypes = [p] * 13 + [i] * 6 + [p]
    handle.sbmc_splat_slab_supported.argtypes = [i] * 6
    handle.sbmc_splat_slab_fwd_f32.argtypes = [p] * 6 + [i] * 9 + [p]
    handle.sbmc_splat_slab_bwd_f32.argtypes = [p] * 13 + [i] * 8 + [p]
    handle.sbmc_splat_slab_fwd_f16.argtypes = handle.sbmc_splat_slab_fwd_f32.argtypes
    handle.sbmc_splat_slab_bwd_f16.argtypes = handle.sbmc_splat_slab_bwd_f32.argtypes
    handle.sbmc_splat_update_fwd_f16.argtypes = handle.sbmc_splat_update_fwd_f32.argtypes
    handle.sbmc_splat_update_bwd_f16.argtypes = handle.sbmc_splat_update_bwd_f32.argtypes
    handle.sbmc_splat_all_bwd_f16.argtypes = handle.sbmc_splat_all_bwd_f32.argtypes
    handle.sbmc_gather_update_supported.argtypes = [i] * 4
    handle.sbmc_gather_update_fwd_f32.argtypes = handle.sbmc_splat_update_fwd_f32.argtypes
    handle.sbmc_gather_update_bwd_f32.argtypes = handle.sbmc_splat_update_bwd_f32.argtypes
    handle.sbmc_bias_act_chunks.argtypes = [i, i, ctypes.c_long]
    handle.sbmc_bias_act_fwd_f32.argtypes = [p, p, i, i, ctypes.c_long, i, ctypes.c_float, p]
    handle.sbmc_bias_act_bwd_f32.argtypes = [p, p, p, p, i, i, ctypes.c_long, i, ctypes.c_float, p]
    handle.sbmc_ctx_act_fwd_f32.argtypes = [p, p, p, i, i, i, ctypes.c_long, i, i, ctypes.c_float, p]
    handle.sbmc_ctx_act_bwd_f32.argtypes = [p, p, p, p, p, i, i, i, ctypes.c_long, i, i, ctypes.c_float, p]
    handle.sbmc_pointwise_supported.argtypes = [i, i, ctypes.c_long]
    handle.sbmc_pointwise_fwd_f32.argtypes = [p] * 5 + [i, i, i, i, ctypes.c_long, i, i, ctypes.c_float, p]
    handle.sbmc_pointwise_fwd_f16.argtypes = [p, i, p, p, p, p, i, i, i, i, ctypes.c_long, i, i, ctypes.c_float, p]
    handle.sbmc_pointwise_bwd_supported.argtypes = [i, i, ctypes.c_long]
    handle.sbmc_pointwise_bwd_groups.argtypes = [i, i, i, ctypes.c_long]
    handle.sbmc_pointwise_gw_wide_supported.argtypes = [i, i, ctypes.c_long]
    handle.sbmc_pointwise_gw_wide_groups.argtypes = [i, ctypes.c_long]
    handle.sbmc_pointwise_gw_wide_f32.argtypes = [p, p, p, p, i, i, i, ctypes.c_long, p]
    handle.sbmc_pointwise_gw_wide_f16.argtypes = [p, p, p, p, i, i, i, ctypes.c_long, p]
    handle.sbmc_pointwise_bwd_f32.argtypes = [p] * 9 + [i, i, i, i, i, ctypes.c_long, i, i, ctypes.c_float, p]
    handle.sbmc_pointwise_fwd_signs_f32.argtypes = [p] * 6 + [i, i, i, i, ctypes.c_long, i, i, ctypes.c_float, p]
    handle.sbmc_pointwise_bwd_signs_f32.argtypes = handle.sbmc_pointwise_bwd_f32.argtypes
    handle.sbmc_pointwise_fwd_mean_f32.argtypes = [p] * 7 + [i, i, i, i, i, ctypes.c_long, i, i, ctypes.c_float, p]
    handle.sbmc_pointwise_fwd_scaled_f32.argtypes = [p] * 7 + [i, p, p, i, i, i, i, ctypes.c_long, i, i, ctypes.c_float, p]
    handle.sbmc_pointwise_bwd_scaled_f32.argtypes = [p] * 9 + [i] + [p] * 4 + [i, i, i, i, ctypes.c_long, i, i, ctypes.c_float, p]
    handle.sbmc_pointwise_wide_bwd_ws_bytes.argtypes = []
    handle.sbmc_pointwise_wide_bwd_f32.argtypes = [p] * 10 + [i, i, i, ctypes.c_long, p]
    handle.sbmc_pointwise_chain_supported.argtypes = [i, i, p, ctypes.c_long]
    handle.sbmc_pointwise_wide_fwd_supported.argtypes = [i, i, ctypes.c_long]
    handle.sbmc_pointwise_chain_bwd_supported.argtypes = [i, ctypes.c_long]
    handle.sbmc_pointwise_chain_bwd_groups.argtypes = [i, i, i, ctypes.c_long]
    handle.sbmc_pointwise_chain_bwd_f32.argtypes = [p] * 16 + [i, i, i, ctypes.c_long, i, i, ctypes.c_float, i, ctypes.c_float, p]
    handle.sbmc_pointwise_wide_fwd_f32.argtypes = [p] * 5 + [i, i, i, ctypes.c_long, i, ctypes.c_float, p]
    handle.sbmc_pointwise_chain_fwd_f32.argtypes = [p] * 9 + [i, p, p, p, i, i, i, ctypes.c_long, i, p]
    handle.sbmc_splat_all_bwd_bound_f32.argtypes = [p] * 15 + [i] * 8 + [p]
    handle.sbmc_pointwise_fwd_mean_f16.argtypes = [p] * 6 + [i, i, i, i, i, ctypes.c_long, i, i, ctypes.c_float, p]
    handle.sbmc_pointwise_bwd_f16.argtypes = [p, p, p, i, p, p, p, p, p, p, i, i, i, i, i, ctypes.c_long, i, i,
                                              ctypes.c_float, p]
    handle.sbmc_upsample2x_cat_supported.argtypes = [i, i]
    handle.sbmc_upsample2x_cat_fwd_f32.argtypes = [p, p, p, i, i, i, i, i, p]
    handle.sbmc_upsample2x_cat_bwd_f32.argtypes = [p, p, i, i, i, i, i, p]
    handle.sbmc_upsample2x_cat_slab_fwd_f32.argtypes = [p, p, p, i, i, i, i, i, i, i, p]
    handle.sbmc_upsample2x_cat_slab_bwd_f32.argtypes = [p, p, i, i, i, i, i, i, i, p]
    handle.sbmc_transpose2d_f32.argtypes = [p, p, i, i, i, p]
    handle.sbmc_transpose2d_f16.argtypes = [p, p, i, i, i, p]
    handle.sbmc_maxpool2_nhwc_fwd.argtypes = [p, p, i, i, i, i, i, p]
    handle.sbmc_maxpool2_nhwc_bwd_add.argtypes = [p, p, p, p, i, i, i, i, i, p]
    handle.sbmc_maxpool2_nhwc_bwd_add_adj_f32.argtypes = [p, p, p, p, p, ctypes.c_float, p, p, i, i, i, i, p]
    handle.sbmc_bias_act_nhwc_supported.argtypes = [i]
    handle.sbmc_bias_act_nhwc_chunks.argtypes = [ctypes.c_long, i]
    handle.sbmc_bias_act_nhwc_fwd_f32.argtypes = [p, p, ctypes.c_long, i, i, ctypes.c_float, p]
    handle.sbmc_bias_act_nhwc_bwd_f32.argtypes = [p, p, p, p, ctypes.c_long, i, i, ctypes.c_float, p]
    handle.sbmc_bias_act_nhwc_fwd_signs_f32.argtypes = [p, p, p, ctypes.c_long, i, i, ctypes.c_float, p]
    handle.sbmc_bias_act_nhwc_bwd_signs_f32.argtypes = [p, p, p, p, ctypes.c_long, i, i, ctypes.c_float, p]
    handle.sbmc_upsample2x_cat_nhwc_supported.argtypes = [i, i, i, i]
    handle.sbmc_upsample2x_cat_nhwc_fwd_f32.argtypes = [p, p, p, i, i, i, i, i, p]
    handle.sbmc_upsample2x_cat_nhwc_bwd_f32.argtypes = [p, p, p, i, i, i, i, i, p]
    handle.sbmc_upsample2x_cat_nhwc_slab_fwd_f32.argtypes = [p, p, p, i, i, i, i, i, i, i, p]
    handle.sbmc_upsample2x_cat_nhwc_slab_bwd_f32.argtypes = [p, p, p, i, i, i, i, i, i, i, p]
    handle.sbmc_upsample2x_cat_nhwc_bwd_adj_f32.argtypes = [p, p, p, p, ctypes.c_float, p, p, i, i, i, i, i, p]
    handle.sbmc_upsample2x_cat_nhwc_slab_fwd_f16.argtypes = handle.sbmc_upsample2x_cat_nhwc_slab_fwd_f32.argtypes
    handle.sbmc_upsample2x_cat_nhwc_slab_bwd_f16.argtypes = handle.sbmc_upsample2x_cat_nhwc_slab_bwd_f32.argtypes
    ll, u = ctypes.c_longlong, ctypes.c_uint
    handle.sbmc_halo_bytes.argtypes = [ll, i]
    handle.sbmc_halo_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(p), ctypes.c_char_p]
    handle.sbmc_halo_free.argtypes = [p]
    handle.sbmc_halo_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(p)]
    handle.sbmc_halo_close.argtypes = [p]
    handle.sbmc_halo_status.argtypes = [p, ctypes.POINTER(u)]
    handle.sbmc_halo_status_to.argtypes = [p, p, p]
    handle.sbmc_halo_put.argtypes = [p] * 5 + [ll, ll, ll, u, u, i, ll, ll, p, p]
    handle.sbmc_halo_get.argtypes = [p] * 7 + [i, ll, ll, ll, ll, p, p, ll, ll, ll, ll, u, u, i, ll, ll, p, p]
    handle.sbmc_wbank_forward_f32.argtypes = [p, i, p]
    handle.sbmc_wbank_backward_f32.argtypes = [p, i, p]
    handle.sbmc_halo_merge_state_fwd_f32.argtypes = [p] * 7 + [i] * 7 + [u, u, i, ll, ll, p]
    handle.sbmc_halo_merge_state_bwd_f32.argtypes = [p] * 7 + [i] * 7 + [p]
    lg = ctypes.c_long
    handle.sbmc_transpose2d_amax_f32.argtypes = [p, p, p, i, i, i, p]
    handle.sbmc_bias_act_nhwc_fwd_amax_f32.argtypes = [p, p, p, p, lg, i, i, ctypes.c_float, p]
    handle.sbmc_bias_act_nhwc_bwd_amax_f32.argtypes = [p, p, p, p, p, lg, i, i, ctypes.c_float, p]
    handle.sbmc_conv3x3_supported.argtypes = [i] * 5
    handle.sbmc_conv3x3_weights_bytes.argtypes = [i, i]
    handle.sbmc_conv3x3_absmax_f32.argtypes = [p, lg, p, p]
    handle.sbmc_conv3x3_absmax_raise_f32.argtypes = [p, lg, p, p]
    handle.sbmc_conv3x3_prepare_weights_f32.argtypes = [p, lg, lg, lg, lg, lg, i, i, i, p, p]
    handle.sbmc_conv3x3_workspace_bytes.argtypes = []
    handle.sbmc_conv3x3_nhwc_f32.argtypes = [p, p, p, p, i, i, i, i, i, p, p]
    handle.sbmc_conv3x3_bias_act_nhwc_f32.argtypes = [p] * 7 + [i] * 6 + [ctypes.c_float, p, p]
    handle.sbmc_conv3x3_adj_supported.argtypes = [i] * 5
    handle.sbmc_conv3x3_adj_partial_rows.argtypes = []
    handle.sbmc_conv3x3_adj_nhwc_f32.argtypes = [p, p, p, p, ctypes.c_float, p, p, p, i, i, i, i, i, p, p]
    handle.sbmc_conv3x3_wgrad_supported.argtypes = [i] * 5
    handle.sbmc_conv3x3_wgrad_scratch_bytes.argtypes = [i] * 5
    handle.sbmc_conv3x3_wgrad_f32.argtypes = [p, p, p, p, p, lg, lg, lg, lg, p, i, i, i, i, i, p]
    handle.sbmc_conv3x3_wgrad_bias_f32.argtypes = [p, p, p, p, p, lg, lg, lg, lg, p, i, i, i, i, i, p, i, i, p, p]
    handle.sbmc_conv3x3_nhwc_f16.argtypes = [p, p, p, i, i, i, i, i, p, p]
    handle.sbmc_conv3x3_bias_act_nhwc_f16.argtypes = [p, p, p, p, p, i, i, i, i, i, i, ctypes.c_float, p, p]
    handle.sbmc_conv3x3_wgrad_bias_f16.argtypes = [p, p, p, lg, lg, lg, lg, p, i, i, i, i, i, p, i, i, p, p]
    handle.sbmc_bias_act_nhwc_bwd_signs_f16.argtypes = [p, p, p, p, lg, i, i, ctypes.c_float, p]
    for name in SYMBOLS[2:]:
        getattr(handle, name).restype = i
    handle.sbmc_halo_bytes.restype = ctypes.c_size_t
    handle.sbmc_pointwise_wide_bwd_ws_bytes.restype = ctypes.c_size_t
    handle.sbmc_conv3x3_weights_bytes.restype = ctypes.c_size_t
    handle.sbmc_conv3x3_workspace_bytes.restype = ctypes.c_size_t
    handle.sbmc_conv3x3_wgrad_scratch_bytes.restype = ctypes.c_size_t
    handle.sbmc_splat_update_bwd_scratch_bytes.restype = ctypes.c_size_t
    if handle.sbmc_hip_abi_version() != ABI_VERSION:
        raise HipExtensionMissing("ABI version mismatch: rebuild with `python -m sbmc_amd.build --force`")
    _LIB = handle
    return _LIB


def check(rc, what):
    if rc != 0:
        msg = lib().sbmc_hip_strerror(rc)
        raise RuntimeError("%s failed: %s (code %d)" % (what, msg.decode() if msg else "?", rc))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
