"""CPU oracle for the SBMC splat hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``sbmc_amd/`` may import this module.  It is used by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` as the
checker / timed CPU port, never as the product path.

Two layers (citations relative to /root/reference):

1. The three native operators, restated in C (``sbmc_oracle.c``) from the
   reference's Halide generators src/kernel_weighting.cpp:28-124 and
   src/scatter2gather.cpp:29-52 and exposed here under the six names the
   reference extension module ``sbmc.halide_ops`` exports (setup.py:65-84):
   ``{scatter2gather,kernel_weighting,kernel_weighting_grad}_{cpu,cuda}_float32``.
   Both the ``_cpu_`` and ``_cuda_`` spellings run the same C code on host
   tensors (there is no GPU in an oracle).

2. The Python composition the reference builds on top of those operators,
   restated with plain torch-CPU ops so that torch autograd yields the
   reference's backward: ``Scatter2Gather`` / ``KernelWeighting``
   (sbmc/functions.py:39-115), ``kernel_apply`` (sbmc/modules.py:338-361) and
   ``progressive_kernel_apply`` (sbmc/modules.py:376-473).

Parity pinning: see the header of ``sbmc_oracle.c``.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch as th

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsbmc_oracle.so")
_SRC = os.path.join(_HERE, "sbmc_oracle.c")
_INC = os.path.join(_HERE, "sbmc_oracle_ops.inc")
_LIB = None


def _source_hash():
    import hashlib
    h = hashlib.sha256()
    for path in (_SRC, _INC, os.path.join(_HERE, "Makefile")):
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False):
    """Compile ``sbmc_oracle.c`` into ``libsbmc_oracle.so`` (gcc + OpenMP).  Staleness is decided by
    content (a hash of the sources recorded beside the library), not by file times, which need not
    survive the copy to a GPU box."""
    stamp = _SO + ".srchash"
    digest = _source_hash()
    stale = not os.path.exists(_SO)
    if not stale:
        try:
            with open(stamp) as f:
                stale = f.read().strip() != digest
        except OSError:
            stale = True
    if force or stale:
        subprocess.check_call(["make", "-s", "-B", "-C", _HERE])
        tmp = "%s.%d.tmp" % (stamp, os.getpid())
        with open(tmp, "w") as f:
            f.write(digest + "\n")
        os.replace(tmp, stamp)
    return _SO


def lib():
    global _LIB
    if _LIB is None:
        build()
        _LIB = ctypes.CDLL(_SO)
        fp = ctypes.c_void_p
        i = ctypes.c_int
        _LIB.sbmc_oracle_kernel_weighting.argtypes = [fp, fp, fp, fp, i, i, i, i, i, i]
        _LIB.sbmc_oracle_kernel_weighting_grad.argtypes = [fp] * 7 + [i] * 6
        _LIB.sbmc_oracle_scatter2gather.argtypes = [fp, fp, i, i, i, i, i]
        _LIB.sbmc_oracle_kernel_weighting_f64.argtypes = _LIB.sbmc_oracle_kernel_weighting.argtypes
        _LIB.sbmc_oracle_kernel_weighting_grad_f64.argtypes = _LIB.sbmc_oracle_kernel_weighting_grad.argtypes
        _LIB.sbmc_oracle_scatter2gather_f64.argtypes = _LIB.sbmc_oracle_scatter2gather.argtypes
        for f in (_LIB.sbmc_oracle_kernel_weighting,
                  _LIB.sbmc_oracle_kernel_weighting_grad,
                  _LIB.sbmc_oracle_scatter2gather,
                  _LIB.sbmc_oracle_kernel_weighting_f64,
                  _LIB.sbmc_oracle_kernel_weighting_grad_f64,
                  _LIB.sbmc_oracle_scatter2gather_f64):
            f.restype = ctypes.c_int
    return _LIB


def _chk(*tensors):
    """Returns "" for float32 tensors (the oracle proper: the reference ops are *_float32) or "_f64"
    for float64 ones (the same expressions evaluated in double: the tests' yardstick where two fp32
    implementations are both limited by cancellation -- never compared as "the reference")."""
    dt = tensors[0].dtype
    for t in tensors:
        if t.is_cuda:
            raise RuntimeError("the oracle only runs on host tensors")
        if t.dtype != dt or dt not in (th.float32, th.float64):
            raise RuntimeError("the oracle takes float32 tensors (or all-float64 ones for its double evaluation)")
        if not t.is_contiguous():
            raise RuntimeError("the oracle expects contiguous tensors")
    return "" if dt == th.float32 else "_f64"


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


# -- the six names of sbmc.halide_ops (reference setup.py:65-84) --------------
def scatter2gather_cpu_float32(weights, output):
    sfx = _chk(weights, output)
    bs, kh, kw, h, w = weights.shape
    assert output.shape == weights.shape
    rc = getattr(lib(), "sbmc_oracle_scatter2gather" + sfx)(_p(weights), _p(output), bs, h, w, kh, kw)
    if rc:
        raise RuntimeError("sbmc_oracle_scatter2gather failed (%d)" % rc)


def kernel_weighting_cpu_float32(data, weights, output, sum_w):
    sfx = _chk(data, weights, output, sum_w)
    bs, c, h, w = data.shape
    _, kh, kw, _, _ = weights.shape
    assert weights.shape == (bs, kh, kw, h, w)
    assert output.shape == data.shape and sum_w.shape == (bs, h, w)
    rc = getattr(lib(), "sbmc_oracle_kernel_weighting" + sfx)(
        _p(data), _p(weights), _p(output), _p(sum_w), bs, c, h, w, kh, kw)
    if rc:
        raise RuntimeError("sbmc_oracle_kernel_weighting failed (%d)" % rc)


def kernel_weighting_grad_cpu_float32(data, weights, sum_w, d_output, d_sum_w,
                                      d_data, d_weights):
    sfx = _chk(data, weights, sum_w, d_output, d_sum_w, d_data, d_weights)
    bs, c, h, w = data.shape
    _, kh, kw, _, _ = weights.shape
    rc = getattr(lib(), "sbmc_oracle_kernel_weighting_grad" + sfx)(
        _p(data), _p(weights), _p(sum_w), _p(d_output), _p(d_sum_w),
        _p(d_data), _p(d_weights), bs, c, h, w, kh, kw)
    if rc:
        raise RuntimeError("sbmc_oracle_kernel_weighting_grad failed (%d)" % rc)


scatter2gather_cuda_float32 = scatter2gather_cpu_float32
kernel_weighting_cuda_float32 = kernel_weighting_cpu_float32
kernel_weighting_grad_cuda_float32 = kernel_weighting_grad_cpu_float32


# -- autograd wrappers (restating sbmc/functions.py:39-115) -------------------
class Scatter2Gather(th.autograd.Function):
    """functions.py:39-71: the op is its own adjoint."""

    @staticmethod
    def forward(ctx, data):
        assert data.dim() == 5, "data should be 5d"
        data = data.contiguous()
        out = th.empty_like(data)
        scatter2gather_cpu_float32(data, out)
        return out

    @staticmethod
    def backward(ctx, d_output):
        d_output = d_output.contiguous()
        d_data = th.empty_like(d_output)
        scatter2gather_cpu_float32(d_output, d_data)
        return d_data


class KernelWeighting(th.autograd.Function):
    """functions.py:74-115."""

    @staticmethod
    def forward(ctx, data, weights):
        bs, c, h, w = data.shape
        data = data.contiguous()
        weights = weights.contiguous()
        output = th.empty_like(data)
        sum_w = data.new_empty(bs, h, w)
        kernel_weighting_cpu_float32(data, weights, output, sum_w)
        ctx.save_for_backward(data, weights, sum_w)
        return output, sum_w

    @staticmethod
    def backward(ctx, d_output, d_sum_w):
        data, weights, sum_w = ctx.saved_tensors
        d_data = th.empty_like(data)
        d_weights = th.empty_like(weights)
        kernel_weighting_grad_cpu_float32(
            data, weights, sum_w, d_output.contiguous(), d_sum_w.contiguous(),
            d_data, d_weights)
        return d_data, d_weights


# -- module compositions (restating sbmc/modules.py) --------------------------
def kernel_apply(data, kernels, softmax=True, splat=True):
    """modules.py:338-361 (KernelApply.forward)."""
    bs, k2, h, w = kernels.shape
    k = int(np.sqrt(k2))
    kernels = kernels.view(bs, k, k, h, w)
    if splat:
        kernels = Scatter2Gather.apply(kernels)
    if softmax:
        kernels = kernels.view(bs, k * k, h, w)
        kernels = th.nn.functional.softmax(kernels, dim=1)
        kernels = kernels.view(bs, k, k, h, w)
    output, sum_w = KernelWeighting.apply(data, kernels)
    return output, sum_w.unsqueeze(1)


def progressive_kernel_apply(data, kernels, sum_r, sum_w, max_w, splat=False):
    """modules.py:376-473 (ProgressiveKernelApply.forward).

    Out-of-place restatement of the in-place ``sub_``/``exp_`` sequence: the
    values and the autograd graph are the same (the reference mutates the
    Scatter2Gather output, or -- with splat=False -- a view of its input).
    """
    bs, k2, h, w = kernels.shape
    k = int(np.sqrt(k2))
    kernels = kernels.view(bs, k, k, h, w)
    if splat:
        kernels = Scatter2Gather.apply(kernels)          # :425
    kernels_view = kernels.reshape(bs, k * k, h, w)
    kmax = kernels_view.max(1, keepdim=True)[0]          # :429

    if sum_r is None:                                    # :431
        if sum_w is not None or max_w is not None:
            raise RuntimeError("all of sum_r, sum_w, max_w should be none")
        max_w = kmax                                     # :438
        kernels = th.exp(kernels - max_w.unsqueeze(1))   # :439-442
        sum_r, sum_w = KernelWeighting.apply(data.contiguous(), kernels.contiguous())
        sum_w = sum_w.unsqueeze(1)
    else:
        new_max = th.max(kmax, max_w)                    # :450
        scaler = th.exp(max_w - new_max)                 # :453
        sum_r = sum_r * scaler                           # :456-457
        sum_w = sum_w * scaler
        max_w = new_max
        kernels = th.exp(kernels - max_w.unsqueeze(1))   # :461-462
        new_sum_r, new_sum_w = KernelWeighting.apply(data.contiguous(), kernels.contiguous())
        new_sum_w = new_sum_w.unsqueeze(1)
        sum_r = sum_r + new_sum_r                        # :470-471
        sum_w = sum_w + new_sum_w
    return sum_r, sum_w, max_w
