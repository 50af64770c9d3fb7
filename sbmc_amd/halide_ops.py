"""Drop-in for the reference extension module ``sbmc.halide_ops``.

The reference builds ``sbmc.halide_ops`` from two Halide generators
(setup.py:65-90); its pybind wrapper exposes six callables that take torch
tensors -- inputs first, then caller-allocated outputs -- and are called only by
``sbmc/functions.py`` (:57-59, :68-70, :96-98, :110-114).  This module exports
the same six names with the same positional signatures and forwards the
``*_cuda_float32`` ones to the hand-written gfx950 kernels behind the C ABI of
``include/sbmc_hip.h`` (ROCm tensors report ``is_cuda == True``, so the
reference's dispatch keeps working unchanged).

The ``*_cpu_float32`` names are NOT implemented here: the reference's CPU path is
Halide-generated x86 code, which is outside this build.  They raise unless a test
harness has installed a host implementation through
``register_cpu_ops_for_testing`` (tests install the parity oracle so that the
Python host logic above the operators can be exercised without a GPU).  The
product never installs one itself.
"""
import torch as th

from . import _lib

__all__ = [
    "scatter2gather_cpu_float32", "scatter2gather_cuda_float32",
    "kernel_weighting_cpu_float32", "kernel_weighting_cuda_float32",
    "kernel_weighting_grad_cpu_float32", "kernel_weighting_grad_cuda_float32",
    # half storage: the names the reference's scheme `<op>_<device>_<dtype>`
    # (setup.py:65-84) gives the same operators on torch.float16 tensors; ROCm only
    "scatter2gather_cuda_float16", "kernel_weighting_cuda_float16", "kernel_weighting_grad_cuda_float16",
]

_CPU_OPS = None


def register_cpu_ops_for_testing(module):
    """Install (or, with None, remove) a host implementation of the ``_cpu_`` names.

    Test-harness hook only; see the module docstring.
    """
    global _CPU_OPS
    _CPU_OPS = module


def _cpu(name):
    if _CPU_OPS is None:
        raise RuntimeError(
            "%s: sbmc_amd implements the SBMC operators for MI355X (ROCm) tensors only; "
            "the reference's Halide CPU path is not part of this build. Move the tensors "
            "to the GPU." % name)
    return getattr(_CPU_OPS, name)


def _check(name, tensors, dtype=th.float32):
    dev = tensors[0].device
    for t in tensors:
        if not isinstance(t, th.Tensor):
            raise RuntimeError("%s: expected torch tensors" % name)
        if not t.is_cuda:
            raise RuntimeError("%s: expected ROCm device tensors (got %s)" % (name, t.device))
        if t.device != dev:
            raise RuntimeError("%s: tensors are on different devices" % name)
        if t.dtype != dtype:
            raise RuntimeError("%s: expected %s tensors (got %s)" % (name, str(dtype).replace("torch.", ""), t.dtype))
        if not t.is_contiguous():
            raise RuntimeError("%s: expected contiguous tensors" % name)
    return dev


def _scatter2gather(weights, output, dtype, sym):
    dev = _check("scatter2gather", (weights, output), dtype)
    if weights.dim() != 5 or output.shape != weights.shape:
        raise RuntimeError("scatter2gather: weights and output should be [bs, kh, kw, h, w]")
    if output.data_ptr() == weights.data_ptr() and weights.numel() > 0:
        raise RuntimeError("scatter2gather: output must not alias the input")
    bs, kh, kw, h, w = weights.shape
    with th.cuda.device(dev):
        rc = getattr(_lib.lib(), sym)(
            _lib.ptr(weights), _lib.ptr(output), bs, h, w, kh, kw, _lib.current_stream(dev))
    _lib.check(rc, "scatter2gather")


def scatter2gather_cuda_float32(weights, output):
    """reference: scatter2gather generator, src/scatter2gather.cpp:59-93."""
    _scatter2gather(weights, output, th.float32, "sbmc_scatter2gather_f32")


def scatter2gather_cuda_float16(weights, output):
    """The same permutation on torch.float16 tensors (bit exact)."""
    _scatter2gather(weights, output, th.float16, "sbmc_scatter2gather_f16")


def _kernel_weighting(data, weights, output, sum_w, dtype, sym):
    dev = _check("kernel_weighting", (data, weights, output, sum_w), dtype)
    if data.dim() != 4 or weights.dim() != 5:
        raise RuntimeError("kernel_weighting: data should be 4d, weights 5d")
    bs, c, h, w = data.shape
    _, kh, kw, _, _ = weights.shape
    if tuple(weights.shape) != (bs, kh, kw, h, w):
        raise RuntimeError("kernel_weighting: weights should be [bs, kh, kw, h, w] matching data")
    if output.shape != data.shape or tuple(sum_w.shape) != (bs, h, w):
        raise RuntimeError("kernel_weighting: bad output / sum_w shape")
    with th.cuda.device(dev):
        rc = getattr(_lib.lib(), sym)(
            _lib.ptr(data), _lib.ptr(weights), _lib.ptr(output), _lib.ptr(sum_w),
            bs, c, h, w, kh, kw, _lib.current_stream(dev))
    _lib.check(rc, "kernel_weighting")


def kernel_weighting_cuda_float32(data, weights, output, sum_w):
    """reference: kernel_weighting generator, src/kernel_weighting.cpp:128-191."""
    _kernel_weighting(data, weights, output, sum_w, th.float32, "sbmc_kernel_weighting_fwd_f32")


def kernel_weighting_cuda_float16(data, weights, output, sum_w):
    """The same operator on torch.float16 tensors: fp32 products and sums, one rounding on the store."""
    _kernel_weighting(data, weights, output, sum_w, th.float16, "sbmc_kernel_weighting_fwd_f16")


def _kernel_weighting_grad(data, weights, sum_w, d_output, d_sum_w, d_data, d_weights, dtype, sym):
    dev = _check("kernel_weighting_grad",
                 (data, weights, sum_w, d_output, d_sum_w, d_data, d_weights), dtype)
    bs, c, h, w = data.shape
    _, kh, kw, _, _ = weights.shape
    if tuple(weights.shape) != (bs, kh, kw, h, w):
        raise RuntimeError("kernel_weighting_grad: weights should be [bs, kh, kw, h, w] matching data")
    if (d_output.shape != data.shape or d_data.shape != data.shape
            or d_weights.shape != weights.shape or tuple(d_sum_w.shape) != (bs, h, w)):
        raise RuntimeError("kernel_weighting_grad: inconsistent shapes")
    with th.cuda.device(dev):
        rc = getattr(_lib.lib(), sym)(
            _lib.ptr(data), _lib.ptr(weights), _lib.ptr(sum_w), _lib.ptr(d_output),
            _lib.ptr(d_sum_w), _lib.ptr(d_data), _lib.ptr(d_weights),
            bs, c, h, w, kh, kw, _lib.current_stream(dev))
    _lib.check(rc, "kernel_weighting_grad")


def kernel_weighting_grad_cuda_float32(data, weights, sum_w, d_output, d_sum_w,
                                       d_data, d_weights):
    """reference: kernel_weighting_grad generator, src/kernel_weighting.cpp:193-238."""
    _kernel_weighting_grad(data, weights, sum_w, d_output, d_sum_w, d_data, d_weights,
                           th.float32, "sbmc_kernel_weighting_bwd_f32")


def kernel_weighting_grad_cuda_float16(data, weights, sum_w, d_output, d_sum_w,
                                       d_data, d_weights):
    """The same operator on torch.float16 tensors."""
    _kernel_weighting_grad(data, weights, sum_w, d_output, d_sum_w, d_data, d_weights,
                           th.float16, "sbmc_kernel_weighting_bwd_f16")


def scatter2gather_cpu_float32(weights, output):
    return _cpu("scatter2gather_cpu_float32")(weights, output)


def kernel_weighting_cpu_float32(data, weights, output, sum_w):
    return _cpu("kernel_weighting_cpu_float32")(data, weights, output, sum_w)


def kernel_weighting_grad_cpu_float32(data, weights, sum_w, d_output, d_sum_w,
                                      d_data, d_weights):
    return _cpu("kernel_weighting_grad_cpu_float32")(
        data, weights, sum_w, d_output, d_sum_w, d_data, d_weights)
