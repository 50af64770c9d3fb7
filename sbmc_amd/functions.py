"""Autograd operators of the SBMC splat path, MI355X-native.

API mirror of the reference's ``sbmc/functions.py`` (:39-115): the classes
``Scatter2Gather`` and ``KernelWeighting`` keep their names, argument order,
returned shapes and dispatch rule (``_is_cuda`` -> ``*_cuda_float32``, else
``*_cpu_float32``).  ``SplatUpdate`` is new: a single autograd operator for the
body of ``ProgressiveKernelApply.forward`` (sbmc/modules.py:422-471) backed by the
fused gfx950 kernels of ``csrc/splat_fused.hip``.
"""
import os

import torch as th

from . import _lib
from . import halide_ops as ops
from .utils import knob

__all__ = ["Scatter2Gather", "KernelWeighting", "SplatUpdate", "splat_update_supported",
           "SplatAll", "splat_all_supported", "splat_all_supported_dims", "splat_slab_supported", "gather_update_supported", "BiasAct", "CtxAct"]


# Optional per-call device timing (used by bench.py for the roofline figure): when a list
# is installed, every fused call appends (name, start_event, end_event), recorded on the
# stream the kernels are launched on (torch's current stream).
_KERNEL_TIMINGS = None
_KERNEL_TIMINGS_ONLY = None


def enable_kernel_timing(store, only=None):
    """store: a list to append (name, start, end) event triples to, or None to disable.  only: a tuple of name
    prefixes -- calls of other names record nothing (bench.py keeps the splat operators' two event pairs per step on
    INSIDE its timed steps, and the few hundred pairs of everything else out of them)."""
    global _KERNEL_TIMINGS, _KERNEL_TIMINGS_ONLY
    _KERNEL_TIMINGS = store
    _KERNEL_TIMINGS_ONLY = None if (store is None or only is None) else tuple(only)


class _timed(object):
    def __init__(self, name, device):
        self.name, self.device = name, device
        self.store = None

    def __enter__(self):
        store = _KERNEL_TIMINGS
        if store is not None and (_KERNEL_TIMINGS_ONLY is None or self.name.startswith(_KERNEL_TIMINGS_ONLY)):
            self.store = store
            self.start = th.cuda.Event(enable_timing=True)
            self.end = th.cuda.Event(enable_timing=True)
            self.start.record(th.cuda.current_stream(self.device))

    def __exit__(self, *exc):
        if self.store is not None:
            self.end.record(th.cuda.current_stream(self.device))
            self.store.append((self.name, self.start, self.end))
            self.store = None
        return False


def _require_f32(op, **tensors):
    """The C ABI takes raw float pointers: anything but a float32 GPU tensor (a half tensor under
    autocast, say) would be read or written out of bounds.  Fail loudly instead."""
    for name, v in tensors.items():
        if v is not None and (v.dtype != th.float32 or not v.is_cuda):
            raise TypeError("%s: %s must be a float32 GPU tensor, got %s on %s" % (op, name, v.dtype, v.device))


def _is_cuda(*args):
    """True if any argument lives on a GPU (reference functions.py:30-36)."""
    for arg in args:
        if arg.is_cuda:
            return True
    return False


class Scatter2Gather(th.autograd.Function):
    """Converts (transposes) scatter kernels into gather kernels.

    Kernel weights at (x, y) for offset (dx, dy) (i.e. scatter[., dy, dx, y, x])
    are put at gather[., -dy, -dx, y+dy, x+dx]  (reference functions.py:39-71).

    Args:
      data(th.Tensor)[bs, k_h, k_w, h, w]: scatter kernel weights.
    Returns:
      (th.Tensor)[bs, k_h, k_w, h, w]: gather kernel weights.
    """

    @staticmethod
    def forward(ctx, data):
        assert len(data.shape) == 5, "data should be 5d"
        data = data.contiguous()
        output = th.empty_like(data)
        if _is_cuda(data):
            # the reference has float32 only; half tensors take the `_float16` instantiation
            (ops.scatter2gather_cuda_float16 if data.dtype == th.float16 else ops.scatter2gather_cuda_float32)(
                data, output)
        else:
            ops.scatter2gather_cpu_float32(data, output)
        return output

    @staticmethod
    def backward(ctx, d_output):
        # the operator is its own adjoint (reference functions.py:62-71)
        d_output = d_output.contiguous()
        d_data = th.empty_like(d_output)
        if _is_cuda(d_output):
            (ops.scatter2gather_cuda_float16 if d_output.dtype == th.float16 else ops.scatter2gather_cuda_float32)(
                d_output, d_data)
        else:
            ops.scatter2gather_cpu_float32(d_output, d_data)
        return d_data


class KernelWeighting(th.autograd.Function):
    """Locally-weighted sum of the input values using kernel weights.

    Args:
      data(th.Tensor)[bs, c, h, w]: input values to be locally averaged.
      weights(th.Tensor)[bs, k_h, k_w, h, w]: kernel weights.  Channels are
        filtered independently.
    Returns:
      output(th.Tensor)[bs, c, h, w]:
        output[., c, y, x] = sum_{dx, dy} weights[., dy, dx, y, x] * data[., c, y+dy, x+dx].
      sum_w(th.Tensor)[bs, h, w]: sum of weights per pixel.
    (reference functions.py:74-115)
    """

    @staticmethod
    def forward(ctx, data, weights):
        bs, c, h, w = data.shape
        data = data.contiguous()
        weights = weights.contiguous()
        output = th.empty_like(data)
        sum_w = data.new_empty(bs, h, w)
        if _is_cuda(data, weights):
            if data.dtype == th.float16 and weights.dtype == th.float16:
                ops.kernel_weighting_cuda_float16(data, weights, output, sum_w)
            else:
                ops.kernel_weighting_cuda_float32(data, weights, output, sum_w)
        else:
            ops.kernel_weighting_cpu_float32(data, weights, output, sum_w)
        ctx.save_for_backward(data, weights, sum_w)
        return output, sum_w

    @staticmethod
    def backward(ctx, d_output, d_sum_w):
        data, weights, sum_w = ctx.saved_tensors
        d_output = d_output.contiguous()
        d_sum_w = d_sum_w.contiguous()
        d_data = th.empty_like(data)
        d_weights = th.empty_like(weights)
        if _is_cuda(d_output, d_sum_w):
            if data.dtype == th.float16 and weights.dtype == th.float16:
                ops.kernel_weighting_grad_cuda_float16(
                    data, weights, sum_w, d_output.half(), d_sum_w.half(), d_data, d_weights)
            else:
                ops.kernel_weighting_grad_cuda_float32(
                    data, weights, sum_w, d_output, d_sum_w, d_data, d_weights)
        else:
            ops.kernel_weighting_grad_cpu_float32(
                data, weights, sum_w, d_output, d_sum_w, d_data, d_weights)
        return d_data, d_weights


def _half_ok(c, k, h, w):
    return bool(_lib.lib().sbmc_splat_f16_supported(int(c), int(k), int(h), int(w)))


def splat_update_supported(data, kernels):
    """True when the fused kernels can take these operands: GPU tensors, odd k, <= 8 channels,
    fp32 radiance, fp32 logits -- or fp16 logits where the strip kernels apply (k = 21)."""
    if not (data.is_cuda and kernels.is_cuda):
        return False
    if data.dtype != th.float32 or kernels.dtype not in (th.float32, th.float16):
        return False
    k2 = kernels.shape[1]
    k = int(round(k2 ** 0.5))
    if k * k != k2:
        return False
    if kernels.dtype == th.float16:
        return _half_ok(data.shape[1], k, kernels.shape[-2], kernels.shape[-1])
    return bool(_lib.lib().sbmc_splat_update_supported(int(data.shape[1]), k))


class SplatUpdate(th.autograd.Function):
    """One progressive splat update, fused (forward AND backward).

    Same inputs / outputs as ``ProgressiveKernelApply(splat=True).forward``
    (reference sbmc/modules.py:376-473):

    Args:
      data(th.Tensor)[bs, c, h, w]: sample radiance.
      kernels(th.Tensor)[bs, k*k, h, w]: sample-centred (splat) kernel logits.
      sum_r(None or th.Tensor)[bs, c, h, w], sum_w, max_w(None or th.Tensor)[bs, 1, h, w]:
        running state; all None on the initialisation call.
    Returns:
      sum_r[bs, c, h, w], sum_w[bs, 1, h, w], max_w[bs, 1, h, w]: updated state.

    Unlike the reference composition it does not modify ``kernels`` and keeps only
    the logits (not the 1.6 GB of exponentiated gather weights) for backward.
    """

    @staticmethod
    def forward(ctx, data, kernels, sum_r, sum_w, max_w, gather=False):
        if sum_r is None:
            if sum_w is not None or max_w is not None:
                raise RuntimeError("all of sum_r, sum_w, max_w should be none")
        elif sum_w is None or max_w is None:
            raise RuntimeError("all of sum_r, sum_w, max_w should be provided")
        bs, k2, h, w = kernels.shape
        k = int(round(k2 ** 0.5))
        c = data.shape[1]
        if tuple(data.shape) != (bs, c, h, w):
            raise RuntimeError("data should be [bs, c, h, w] matching kernels [bs, k*k, h, w]")
        _require_f32("SplatUpdate", data=data, sum_r=sum_r, sum_w=sum_w, max_w=max_w,
                     kernels=None if kernels.dtype == th.float16 else kernels)
        if kernels.dtype == th.float16 and (gather or not kernels.is_cuda or not _half_ok(c, k, h, w)):
            raise TypeError("SplatUpdate: half logits are only taken by the k=21 strip kernels")
        data = data.contiguous()
        kernels = kernels.contiguous()
        first = sum_r is None
        if not first:
            sum_r, sum_w, max_w = sum_r.contiguous(), sum_w.contiguous(), max_w.contiguous()
            if (tuple(sum_r.shape) != (bs, c, h, w) or sum_w.numel() != bs * h * w
                    or max_w.numel() != bs * h * w):
                raise RuntimeError("running state has the wrong shape")
        new_r = th.empty_like(data)
        new_w = data.new_empty(bs, 1, h, w)
        new_m = data.new_empty(bs, 1, h, w)
        kmax = data.new_empty(bs, h, w)
        atap = th.empty(bs, h, w, dtype=th.int32, device=data.device)
        dev = data.device
        half = kernels.dtype == th.float16
        fwd = _lib.lib().sbmc_splat_update_fwd_f16 if half else _lib.lib().sbmc_splat_update_fwd_f32
        if gather:
            fwd = _lib.lib().sbmc_gather_update_fwd_f32
        ctx.gather = gather
        with th.cuda.device(dev), _timed("gather_update_fwd" if gather else "splat_update_fwd", dev):
            rc = fwd(
                _lib.ptr(data), _lib.ptr(kernels), _lib.ptr(sum_r), _lib.ptr(sum_w), _lib.ptr(max_w),
                _lib.ptr(new_r), _lib.ptr(new_w), _lib.ptr(new_m), _lib.ptr(kmax), _lib.ptr(atap),
                bs, c, h, w, k, _lib.current_stream(dev))
        _lib.check(rc, "splat_update_fwd")
        ctx.first = first
        ctx.k = k
        if first:
            ctx.save_for_backward(data, kernels, new_r, new_w, new_m, kmax, atap)
        else:
            ctx.save_for_backward(data, kernels, new_r, new_w, new_m, kmax, atap, sum_r, sum_w, max_w)
        return new_r, new_w, new_m

    @staticmethod
    def backward(ctx, d_r, d_w, d_m):
        saved = ctx.saved_tensors
        data, kernels, new_r, new_w, new_m, kmax, atap = saved[:7]
        sum_r = sum_w = max_w = None
        if not ctx.first:
            sum_r, sum_w, max_w = saved[7:]
        bs, c, h, w = data.shape
        d_r = th.zeros_like(new_r) if d_r is None else d_r.contiguous()
        d_w = th.zeros_like(new_w) if d_w is None else d_w.contiguous()
        d_m = th.zeros_like(new_m) if d_m is None else d_m.contiguous()
        d_data = th.empty_like(data)
        d_kernels = th.empty_like(kernels)
        nbytes = _lib.lib().sbmc_splat_update_bwd_scratch_bytes(bs, c, h, w, ctx.k)
        scratch = data.new_empty((nbytes + 3) // 4)
        d_sum_r = d_sum_w = d_max_w = None
        if not ctx.first:
            d_sum_r = th.empty_like(sum_r)
            d_sum_w = th.empty_like(sum_w)
            d_max_w = th.empty_like(max_w)
        dev = data.device
        half = kernels.dtype == th.float16
        bwd = _lib.lib().sbmc_splat_update_bwd_f16 if half else _lib.lib().sbmc_splat_update_bwd_f32
        if ctx.gather:
            bwd = _lib.lib().sbmc_gather_update_bwd_f32
        with th.cuda.device(dev), _timed("gather_update_bwd" if ctx.gather else "splat_update_bwd", dev):
            rc = bwd(
                _lib.ptr(data), _lib.ptr(kernels), _lib.ptr(sum_r), _lib.ptr(sum_w), _lib.ptr(max_w),
                _lib.ptr(new_r), _lib.ptr(new_w), _lib.ptr(new_m), _lib.ptr(kmax), _lib.ptr(atap),
                _lib.ptr(d_r), _lib.ptr(d_w), _lib.ptr(d_m),
                _lib.ptr(d_data), _lib.ptr(d_kernels),
                _lib.ptr(d_sum_r), _lib.ptr(d_sum_w), _lib.ptr(d_max_w), _lib.ptr(scratch),
                bs, c, h, w, ctx.k, _lib.current_stream(dev))
        _lib.check(rc, "splat_update_bwd")
        return d_data, d_kernels, d_sum_r, d_sum_w, d_max_w, None


class BiasAct(th.autograd.Function):
    """y <- act(y + bias[c]) in place on a planar [b, c, ...] tensor, one HBM pass; backward
    produces the input gradient and the bias gradient in one pass (csrc/bias_act.hip).

    act: 0 linear, 1 relu, 2 leaky_relu(slope).  Used around the batched-GEMM form of the
    per-sample 1x1 convolutions (modules._pointwise_gemm).
    """

    @staticmethod
    def supported(y):
        if y.dim() < 2 or y.numel() == 0:
            return False          # (the C ABI treats an empty batch as a no-op; torch handles it here)
        hw = y[0, 0].numel()
        return (y.is_cuda and y.dtype == th.float32 and y.is_contiguous() and hw % 4 == 0
                and y.data_ptr() % 16 == 0 and y.shape[0] <= 65535 and y.shape[1] <= 65535)

    @staticmethod
    def forward(ctx, y, bias, act, slope):
        _require_f32("BiasAct", y=y, bias=bias)
        b, c = y.shape[0], y.shape[1]
        hw = y[0, 0].numel()
        bias = bias.contiguous()
        dev = y.device
        with th.cuda.device(dev):
            rc = _lib.lib().sbmc_bias_act_fwd_f32(_lib.ptr(y), _lib.ptr(bias), b, c, hw, act, slope,
                                                  _lib.current_stream(dev))
        _lib.check(rc, "bias_act_fwd")
        ctx.mark_dirty(y)
        ctx.act, ctx.slope = act, slope
        if act != 0:
            ctx.save_for_backward(y)   # the linear case needs no activation mask
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        y = ctx.saved_tensors[0] if ctx.act != 0 else gy   # placeholder pointer, never read when linear
        b, c = gy.shape[0], gy.shape[1]
        hw = gy[0, 0].numel()
        gx = th.empty_like(gy)
        partial = gy.new_empty(b, c, _lib.lib().sbmc_bias_act_chunks(b, c, hw))
        dev = gy.device
        with th.cuda.device(dev):
            rc = _lib.lib().sbmc_bias_act_bwd_f32(_lib.ptr(gy), _lib.ptr(y), _lib.ptr(gx), _lib.ptr(partial),
                                                  b, c, hw, ctx.act, ctx.slope, _lib.current_stream(dev))
        _lib.check(rc, "bias_act_bwd")
        return gx, partial.sum((0, 2)), None, None


class CtxAct(th.autograd.Function):
    """y[b,s] <- act(y[b,s] + t[b] + bias) in place: the per-pixel context term of a chain's
    first layer added to the per-sample product (csrc/bias_act.hip, include/sbmc_hip.h).

    y [b*s, c, ...pixels] (modified in place), t [b, c, ...pixels] or [b, c, 1...] (constant over
    the image), bias [c]; act: 0 linear, 1 relu, 2 leaky_relu(slope).
    """

    @staticmethod
    def supported(y, t, s):
        if not BiasAct.supported(y) or y.shape[0] % s or not t.is_contiguous() or t.dtype != th.float32:
            return False
        b = y.shape[0] // s
        hw = y[0, 0].numel()
        return (t.shape[0] == b and t.shape[1] == y.shape[1] and t[0, 0].numel() in (1, hw)
                and t.data_ptr() % 16 == 0)

    @staticmethod
    def forward(ctx, y, t, bias, s, act, slope):
        _require_f32("CtxAct", y=y, t=t, bias=bias)
        c = y.shape[1]
        b = y.shape[0] // s
        hw = y[0, 0].numel()
        per_pixel = int(t[0, 0].numel() == hw and hw > 1)
        bias = bias.contiguous()
        dev = y.device
        with th.cuda.device(dev):
            rc = _lib.lib().sbmc_ctx_act_fwd_f32(_lib.ptr(y), _lib.ptr(t), _lib.ptr(bias), b, s, c, hw,
                                                 per_pixel, act, slope, _lib.current_stream(dev))
        _lib.check(rc, "ctx_act_fwd")
        ctx.mark_dirty(y)
        ctx.cfg = (s, act, slope, per_pixel, tuple(t.shape))
        if act != 0:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        s, act, slope, per_pixel, tshape = ctx.cfg
        gy = gy.contiguous()
        y = ctx.saved_tensors[0] if act != 0 else gy
        c = gy.shape[1]
        b = gy.shape[0] // s
        hw = gy[0, 0].numel()
        gx = th.empty_like(gy)
        gt = gy.new_empty(tshape) if per_pixel else None
        partial = gy.new_empty(b, c, _lib.lib().sbmc_bias_act_chunks(b, c, hw))
        dev = gy.device
        with th.cuda.device(dev):
            rc = _lib.lib().sbmc_ctx_act_bwd_f32(_lib.ptr(gy), _lib.ptr(y), _lib.ptr(gx), _lib.ptr(gt),
                                                 _lib.ptr(partial), b, s, c, hw, per_pixel, act, slope,
                                                 _lib.current_stream(dev))
        _lib.check(rc, "ctx_act_bwd")
        per_image = partial.sum(2)                 # [b, c]
        if not per_pixel:
            gt = per_image.view(tshape)
        return gx, gt, per_image.sum(0), None, None, None


def pointwise_supported(x, cout):
    """True when `PointwiseLayer` applies to the planar activations x [B, cin, ...pixels]."""
    if not (x.is_cuda and x.dtype == th.float32 and x.dim() >= 3 and x.numel() > 0):
        return False
    hw = x[0, 0].numel()
    return (x.data_ptr() % 16 == 0 and x.shape[0] <= 65535 and cout <= 65535
            and not th.is_autocast_enabled()      # reduced-precision activations take the library path
            and bool(_lib.lib().sbmc_pointwise_supported(x.shape[1], cout, hw)))


def pointwise_half_supported(x, cout):
    """True when the half-storage form of `PointwiseLayer` applies: under torch.autocast(float16) on the
    GPU (inference or training), x [B, cin, ...pixels] float32 or float16, dimensions the fused kernels take."""
    if not (x.is_cuda and x.dtype in (th.float32, th.float16) and x.dim() >= 3 and x.numel() > 0):
        return False
    if not th.is_autocast_enabled() or th.get_autocast_dtype("cuda") != th.float16:
        return False
    hw = x[0, 0].numel()
    return (x.data_ptr() % 16 == 0 and x.shape[0] <= 65535 and cout <= 65535
            and bool(_lib.lib().sbmc_pointwise_supported(x.shape[1], cout, hw)))


def pointwise_half(x, w, bias, t, s, act, slope):
    """`PointwiseLayer` forward with half-precision storage: x float32 or float16, output float16;
    w, bias and the context term t are float32, arithmetic is fp32.  No autograd here: training goes
    through `PointwiseLayer.apply(..., half=True)`."""
    _require_f32("pointwise_half", w=w, bias=bias, t=t)
    if not (x.is_cuda and x.dtype in (th.float32, th.float16)):
        raise TypeError("pointwise_half: x must be a float32 or float16 GPU tensor")
    x, w, bias = x.contiguous(), w.contiguous(), bias.contiguous()
    B, cin, hw = x.shape
    cout = w.shape[0]
    t_mode = 0
    if t is not None:
        t = t.contiguous()
        t_mode = 2 if (t.dim() == 3 and t.shape[2] == hw and hw > 1) else 1
    y = th.empty(B, cout, hw, dtype=th.float16, device=x.device)
    dev = x.device
    with th.cuda.device(dev), _timed("pointwise_fwd_f16 %dx%d" % (cout, cin), dev):
        rc = _lib.lib().sbmc_pointwise_fwd_f16(_lib.ptr(x), int(x.dtype == th.float16), _lib.ptr(w), _lib.ptr(bias),
                                               _lib.ptr(t) if t is not None else None, _lib.ptr(y),
                                               B, s, cin, cout, hw, t_mode, act, slope, _lib.current_stream(dev))
    _lib.check(rc, "pointwise_fwd_f16")
    return y


class PointwiseLayer(th.autograd.Function):
    """A whole 1x1-convolution layer in one pass: y[b] = act(w @ x[b] + bias (+ t[b // s])).

    x [B, cin, hw] (cin <= 128), w [cout, cin], bias [cout]; t = None, [B/s, cout] (context term
    constant over the image) or [B/s, cout, hw] (per pixel) -- the context half of a chain's first
    layer, see modules.pointwise_chain_with_context; act: 0 linear, 1 relu, 2 leaky_relu(slope).
    fp32 MFMA kernel csrc/pointwise.hip (exact fp32 products and sums; differs from a library GEMM
    by summation order).  Backward, cout <= 128: one fused pass as well (activation adjoint, gx = w^T gz,
    gw = sum_b gz x^T, bias and context gradients); wider layers: one pass for gz = gy * act'(y) and
    the bias / context gradients (csrc/bias_act.hip), then the two products as library GEMMs.
    """

    @staticmethod
    def forward(ctx, x, w, bias, t, s, act, slope, half=False):
        """half=True ("fp16 activations", torch.autocast(float16)): y is float16, x float16 or float32 (a
        chain's first layer); w, bias, t stay float32 and so does all arithmetic; backward likewise
        (sbmc_pointwise_{fwd,bwd}_f16)."""
        return _pointwise_forward(ctx, x, w, bias, t, s, act, slope, half)[0]

    @staticmethod
    def backward(ctx, gy):
        return _pointwise_backward(ctx, gy, None, 1) + (None, None, None, None)


class PointwiseLayerMean(th.autograd.Function):
    """`PointwiseLayer` that also returns the mean of y over groups of `mean_s` consecutive batch
    elements (the per-pixel mean over samples that feeds the U-net, reference sbmc/models.py:179):
    (y [B, cout, hw], ymean [B / mean_s, cout, hw]).  Both consumers' gradients arrive in ONE backward,
    whose kernel reads gy[b] + gmean[b / mean_s] / mean_s directly: the [B, cout, hw] broadcast-add
    pass autograd would otherwise run (7.5 GB at 720p x 8 spp) disappears."""

    @staticmethod
    def forward(ctx, x, w, bias, t, s, act, slope, mean_s, half=False):
        y, ymean = _pointwise_forward(ctx, x, w, bias, t, s, act, slope, half, mean_s=mean_s)
        ctx.mean_s = mean_s
        if ymean is None:                    # (half storage, wide layers, the fp32-MFMA kernel: one more pass over y)
            B, cout, hw = y.shape
            ymean = y.view(B // mean_s, mean_s, cout, hw).mean(1)
        return y, ymean

    @staticmethod
    def backward(ctx, gy, gmean):
        if gy is None:                       # only the mean was used
            gy = (gmean / ctx.mean_s).repeat_interleave(ctx.mean_s, 0)
            gmean = None
        return _pointwise_backward(ctx, gy, gmean, ctx.mean_s) + (None, None, None, None, None)


def _pointwise_forward(ctx, x, w, bias, t, s, act, slope, half=False, mean_s=0):
    """-> (y, ymean): ymean = the mean of y over groups of mean_s consecutive batch elements where the kernel
    computes it on the way (fp32 split-precision forward, cout <= 128), else None."""
    _require_f32("PointwiseLayer", w=w, bias=bias, t=t, x=None if half else x)
    if half and not (x.is_cuda and x.dtype in (th.float32, th.float16)):
        raise TypeError("PointwiseLayer(half): x must be a float32 or float16 GPU tensor")
    x = x.contiguous()
    w = w.contiguous()
    bias = bias.contiguous()
    B, cin, hw = x.shape
    cout = w.shape[0]
    t_mode = 0
    if t is not None:
        t = t.contiguous()
        t_mode = 2 if (t.dim() == 3 and t.shape[2] == hw and hw > 1) else 1
    y = th.empty(B, cout, hw, dtype=th.float16 if half else th.float32, device=x.device)
    dev = x.device
    L = _lib.lib()
    # an activated fp32 layer whose backward is the fused kernel keeps one SIGN BIT per output for it instead
    # of the output itself (written by the forward kernel: 1/32 of the bytes the adjoint has to read back)
    signs = None
    if (not half and act != 0 and any(ctx.needs_input_grad[:4]) and _pw_split_enabled()
            and L.sbmc_pointwise_bwd_supported(cin, cout, hw)):
        signs = th.empty(B, cout, (hw + 31) // 32, dtype=th.int32, device=dev)
    # magnitude words (ABI 6): the split-precision kernel leaves max |y| in a device word, and -- where the word of x
    # is known (`known_amax`: x came out of such a pass) -- multiplies in the 3 x 3 kernels' number format (two f16
    # planes under a power-of-two scale, three products) instead of three bf16 planes and six.  SBMC_HIP_PW_F16=0:
    # the three-plane form everywhere, no words.
    scaled = not half and _pw_split_enabled() and knob("SBMC_HIP_PW_F16") != 0
    xmax = known_amax(x) if scaled else None
    amax = amax_word(dev) if scaled else None
    ymean = None
    if (mean_s and mean_s >= 1 and not half and _pw_split_enabled() and cout <= 128 and B % mean_s == 0
            and (t_mode == 0 or s == mean_s) and knob("SBMC_PW_FUSED_MEAN") != 0):
        ymean = th.empty(B // mean_s, cout, hw, dtype=th.float32, device=dev)
    half_mean = (mean_s and mean_s >= 1 and half and x.dtype == th.float16 and cout <= 128 and B % mean_s == 0
                 and (t_mode == 0 or s == mean_s) and knob("SBMC_PW_FUSED_MEAN") != 0
                 and knob("SBMC_HIP_PW_F16MFMA") != 0)
    if half_mean:
        ymean = th.empty(B // mean_s, cout, hw, dtype=th.float16, device=dev)
    wide = (scaled and t is None and signs is None and ymean is None and knob("SBMC_PW_WIDE_FWD") != 0
            and bool(L.sbmc_pointwise_wide_fwd_supported(cin, cout, hw)))
    with th.cuda.device(dev), _timed("pointwise_fwd%s %dx%d" % ("_f16" if half else "", cout, cin), dev):
        if wide:
            # the 441-channel logits: the input tile staged once for all four row tiles (csrc/pointwise_chain.hip)
            rc = L.sbmc_pointwise_wide_fwd_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(y), _lib.ptr(amax),
                                               B, cin, cout, hw, act, slope, _lib.current_stream(dev))
        elif scaled:
            rc = L.sbmc_pointwise_fwd_scaled_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(t) if t is not None else None,
                                                 _lib.ptr(y), _lib.ptr(signs) if signs is not None else None,
                                                 _lib.ptr(ymean) if ymean is not None else None, mean_s if ymean is not None else 1,
                                                 _lib.ptr(xmax) if xmax is not None else None, _lib.ptr(amax),
                                                 B, s, cin, cout, hw, t_mode, act, slope, _lib.current_stream(dev))
        elif half_mean:
            rc = L.sbmc_pointwise_fwd_mean_f16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(t) if t is not None else None,
                                               _lib.ptr(y), _lib.ptr(ymean), mean_s, B, s, cin, cout, hw, t_mode, act, slope,
                                               _lib.current_stream(dev))
        elif ymean is not None:
            rc = L.sbmc_pointwise_fwd_mean_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias),
                                               _lib.ptr(t) if t is not None else None, _lib.ptr(y),
                                               _lib.ptr(signs) if signs is not None else None, _lib.ptr(ymean), mean_s,
                                               B, s, cin, cout, hw, t_mode, act, slope, _lib.current_stream(dev))
        elif half:
            rc = L.sbmc_pointwise_fwd_f16(_lib.ptr(x), int(x.dtype == th.float16), _lib.ptr(w), _lib.ptr(bias),
                                          _lib.ptr(t) if t is not None else None, _lib.ptr(y),
                                          B, s, cin, cout, hw, t_mode, act, slope, _lib.current_stream(dev))
        elif signs is not None:
            rc = L.sbmc_pointwise_fwd_signs_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias),
                                                _lib.ptr(t) if t is not None else None, _lib.ptr(y), _lib.ptr(signs),
                                                B, s, cin, cout, hw, t_mode, act, slope, _lib.current_stream(dev))
        else:
            rc = L.sbmc_pointwise_fwd_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias),
                                          _lib.ptr(t) if t is not None else None, _lib.ptr(y),
                                          B, s, cin, cout, hw, t_mode, act, slope, _lib.current_stream(dev))
    _lib.check(rc, "pointwise_fwd")
    ctx.cfg = (s, act, slope, t_mode, None if t is None else tuple(t.shape))
    ctx.half = half
    ctx.has_signs = signs is not None
    ctx.xmax = xmax                  # (the backward's scale of x: the same word, the same values)
    ctx.save_for_backward(x, w, signs if signs is not None else (y if act != 0 else None))
    if amax is not None:
        tag_amax(y, amax)
        if ymean is not None:
            tag_amax(ymean, amax)    # (a mean of values of y: the bound holds)
    return y, ymean


def _pw_split_enabled():
    """SBMC_HIP_PW_SPLIT=0 (development knob, read by the library as well) keeps the fp32-MFMA forward kernel,
    which writes no sign bits."""
    import os
    return knob("SBMC_HIP_PW_SPLIT") != 0


def _pointwise_backward(ctx, gy, gmean, mean_s):
    """-> (gx, gw, gbias, gt).  gmean: gradient of the group mean of y (or None)."""
    s, act, slope, t_mode, tshape = ctx.cfg
    x, w, y = ctx.saved_tensors
    half = ctx.half
    adt = th.float16 if half else th.float32          # storage type of the layer's output side
    gy = gy.to(adt).contiguous()
    B, cout, hw = gy.shape
    L = _lib.lib()
    dev = gy.device
    cin = x.shape[1]
    fused = bool(L.sbmc_pointwise_bwd_supported(cin, cout, hw))
    if gmean is not None and not (fused and t_mode != 2):
        gy = gy + (gmean / mean_s).repeat_interleave(mean_s, 0).to(adt)   # the kernels below take one gradient
        gmean = None
    if y is None:
        y = gy                                   # placeholder pointer, never read when linear
    if fused:
        # one pass: activation adjoint, both GEMMs, bias and context gradients
        if gmean is not None and t_mode == 0:
            groups = L.sbmc_pointwise_bwd_groups(B, mean_s, 1, hw)     # the walk groups the samples of a pixel
        else:
            groups = L.sbmc_pointwise_bwd_groups(B, s, t_mode, hw)
        nb = B // s if t_mode == 1 else 1
        gx = th.empty_like(x) if ctx.needs_input_grad[0] else None
        gwp = w.new_empty(groups, cout, cin)
        gbp = w.new_empty(groups, nb, cout)
        gt = w.new_empty(tshape) if t_mode == 2 else None
        if gmean is not None:
            gmean = gmean.to(adt).contiguous()
        # magnitude words: the two-f16-plane form wherever the words of gy (and gmean) and of x are known; the largest
        # |gx| goes out in a word for the backward of the layer before
        scaled = (not half and _pw_split_enabled() and knob("SBMC_HIP_PW_F16") != 0 and knob("SBMC_HIP_PW_GWS", 2) >= 2
                  and (act == 0 or getattr(ctx, "has_signs", False)))
        gmax = gmmax = xmax = gxmax = None
        if scaled:
            gmax, xmax = known_amax(gy), getattr(ctx, "xmax", None)
            # (the mean's gradient comes out of the U-net's first convolution, 1 / S of gy's bytes: where it carries
            # no word, the absmax pass over it costs less than the three-plane form of this layer)
            gmmax = ensure_amax(gmean) if (gmean is not None and gmean.is_contiguous()) else None
            if gmax is None or xmax is None or (gmean is not None and gmmax is None):
                gmax = gmmax = xmax = None
            # (the three-plane kernels with a context or mean gradient keep no running maximum: no room)
            if gx is not None and (gmax is not None or (t_mode != 2 and gmean is None)):
                gxmax = amax_word(dev)
        with th.cuda.device(dev), _timed("pointwise_bwd%s %dx%d%s" % ("_f16" if half else "", cout, cin,
                                                                      "" if gx is not None else " (no gx)"), dev):
            tail = (_lib.ptr(w), _lib.ptr(gx) if gx is not None else None, _lib.ptr(gwp),
                    _lib.ptr(gbp), _lib.ptr(gt) if gt is not None else None,
                    _lib.ptr(gmean) if gmean is not None else None, mean_s,
                    B, s, cin, cout, hw, t_mode, act, slope, _lib.current_stream(dev))
            if half:
                rc = L.sbmc_pointwise_bwd_f16(_lib.ptr(gy), _lib.ptr(y), _lib.ptr(x), int(x.dtype == th.float16), *tail)
            elif scaled:
                rc = L.sbmc_pointwise_bwd_scaled_f32(
                    _lib.ptr(gy), _lib.ptr(y) if act != 0 else None, _lib.ptr(x), *tail[:7],
                    _lib.ptr(gmax) if gmax is not None else None, _lib.ptr(gmmax) if gmmax is not None else None,
                    _lib.ptr(xmax) if xmax is not None else None, _lib.ptr(gxmax) if gxmax is not None else None, *tail[7:])
            elif getattr(ctx, "has_signs", False):
                rc = L.sbmc_pointwise_bwd_signs_f32(_lib.ptr(gy), _lib.ptr(y), _lib.ptr(x), *tail)    # y: the sign bits
            else:
                rc = L.sbmc_pointwise_bwd_f32(_lib.ptr(gy), _lib.ptr(y), _lib.ptr(x), *tail)
        _lib.check(rc, "pointwise_bwd")
        per_image = gbp.sum(0)                   # [nb, cout]
        if t_mode == 1:
            gt = per_image.view(tshape)
        if gxmax is not None:
            tag_amax(gx, gxmax)
        return gx, gwp.sum(0), per_image.sum(0), gt
    if (half and act == 0 and t_mode == 0 and x.dtype == th.float16 and knob("SBMC_HIP_PW_GW_WIDE") != 0
            and L.sbmc_pointwise_gw_wide_supported(cin, cout, hw)):
        # the 441-channel logits layer with half activations: weight + bias gradient in ONE pass over the half logit
        # gradient (pw_gw_wide_kernel; before: a cast of the 6.6 GB gradient to fp32 for the bias sums, a reduction
        # over the 13 GB copy and a half GEMM); the data gradient stays a half GEMM
        gx = gw = gbias = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            groups = L.sbmc_pointwise_gw_wide_groups(B, hw)
            gwp = th.empty(groups, cout, cin, dtype=th.float32, device=dev)
            gbp = th.empty(groups, cout, dtype=th.float32, device=dev)
            with th.cuda.device(dev), _timed("pointwise_gw_wide_f16 %dx%d" % (cout, cin), dev):
                _lib.check(L.sbmc_pointwise_gw_wide_f16(_lib.ptr(gy), _lib.ptr(x.contiguous()), _lib.ptr(gwp), _lib.ptr(gbp), B,
                                                        cin, cout, hw, _lib.current_stream(dev)), "pointwise_gw_wide_f16")
            gw, gbias = gwp.sum(0), gbp.sum(0)
        if ctx.needs_input_grad[0]:
            with th.autocast("cuda", enabled=False):
                gx = th.bmm(w.half().t().unsqueeze(0).expand(B, -1, -1), gy).to(x.dtype)
        return gx, gw, gbias, None
    if half:
        # wider than the fused backward takes (the 441-channel logits): activation adjoint and the sums in
        # torch, the two products as half GEMMs with fp32 accumulation (rocBLAS / hipBLASLt)
        gz = gy if act == 0 else th.where(y > 0, gy, gy * slope)
        gzf = gz.float()
        gbias = gzf.sum((0, 2))
        gt = None
        if t_mode == 1:
            gt = gzf.view(B // s, s, cout, hw).sum((1, 3)).view(tshape)
        elif t_mode == 2:
            gt = gzf.view(B // s, s, cout, hw).sum(1)
        gx = gw = None
        with th.autocast("cuda", enabled=False):
            if ctx.needs_input_grad[0]:
                gx = th.bmm(w.half().t().unsqueeze(0).expand(B, -1, -1), gz).to(x.dtype)
            if ctx.needs_input_grad[1]:
                gw = th.bmm(gz, x.half().transpose(1, 2)).float().sum(0)
        return gx, gw, gbias, gt
    if (act == 0 and t_mode == 0 and not half and gy.dtype == th.float32 and x.dtype == th.float32
            and knob("SBMC_HIP_PW_GW_WIDE") != 0 and L.sbmc_pointwise_gw_wide_supported(cin, cout, hw)):
        # the 441-channel logits layer (linear): weight and bias gradient in ONE pass over the logit gradient on the
        # bf16 matrix pipe at fp32 accuracy (csrc/pointwise.hip pw_gw_wide_kernel) instead of a read-only pass for the
        # bias sums + a library GEMM on the fp32 pipe; the data gradient (its reduction runs over the 441 channels: the
        # weights do not fit a CU in split form) stays a library GEMM
        gx = gw = gbias = None
        gmax, xmax = known_amax(gy), getattr(ctx, "xmax", None)
        if (ctx.needs_input_grad[0] and gmax is not None and xmax is not None and _pw_split_enabled()
                and knob("SBMC_HIP_PW_F16") != 0 and knob("SBMC_HIP_PW_WIDE_FUSED") != 0):
            # the whole backward in ONE pass over the logit gradient (pw_wide_bwd2_kernel: two f16 planes; the scale of gy
            # from the bound the splat's backward left, of x from the forward's word): no second 13 GB read, no library GEMM
            groups = L.sbmc_pointwise_gw_wide_groups(B, hw)
            gwp = w.new_empty(groups, cout, cin)
            gbp = w.new_empty(groups, cout)
            gx = th.empty_like(x)
            ws = th.empty(L.sbmc_pointwise_wide_bwd_ws_bytes(), dtype=th.uint8, device=dev)
            gxmax = amax_word(dev)
            with th.cuda.device(dev), _timed("pointwise_wide_bwd %dx%d" % (cout, cin), dev):
                _lib.check(L.sbmc_pointwise_wide_bwd_f32(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(w), _lib.ptr(gx), _lib.ptr(gwp),
                                                         _lib.ptr(gbp), _lib.ptr(ws), _lib.ptr(gmax), _lib.ptr(xmax),
                                                         _lib.ptr(gxmax), B, cin, cout, hw, _lib.current_stream(dev)),
                           "pointwise_wide_bwd")
            tag_amax(gx, gxmax)
            return gx, gwp.sum(0), gbp.sum(0), None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            groups = L.sbmc_pointwise_gw_wide_groups(B, hw)
            gwp = w.new_empty(groups, cout, cin)
            gbp = w.new_empty(groups, cout)
            with th.cuda.device(dev), _timed("pointwise_gw_wide %dx%d" % (cout, cin), dev):
                _lib.check(L.sbmc_pointwise_gw_wide_f32(_lib.ptr(gy), _lib.ptr(x.contiguous()), _lib.ptr(gwp), _lib.ptr(gbp), B,
                                                        cin, cout, hw, _lib.current_stream(dev)), "pointwise_gw_wide")
            gw, gbias = gwp.sum(0), gbp.sum(0)
        if ctx.needs_input_grad[0]:
            gx = th.bmm(w.t().unsqueeze(0).expand(B, -1, -1), gy)
        return gx, gw, gbias, None
    gz = gy if (act == 0 and t_mode == 0) else th.empty_like(gy)   # linear: gz is gy, only sums needed
    gt = None
    with th.cuda.device(dev):
        if t_mode == 0:
            partial = gy.new_empty(B, cout, L.sbmc_bias_act_chunks(B, cout, hw))
            rc = L.sbmc_bias_act_bwd_f32(_lib.ptr(gy), _lib.ptr(y), _lib.ptr(gz), _lib.ptr(partial),
                                         B, cout, hw, act, slope, _lib.current_stream(dev))
        else:
            b = B // s
            gt = gy.new_empty(tshape) if t_mode == 2 else None
            partial = gy.new_empty(b, cout, L.sbmc_bias_act_chunks(b, cout, hw))
            rc = L.sbmc_ctx_act_bwd_f32(_lib.ptr(gy), _lib.ptr(y), _lib.ptr(gz), _lib.ptr(gt),
                                        _lib.ptr(partial), b, s, cout, hw, int(t_mode == 2), act, slope,
                                        _lib.current_stream(dev))
    _lib.check(rc, "pointwise_bwd")
    per_image = partial.sum(2)
    gbias = per_image.sum(0)
    if t_mode == 1:
        gt = per_image.view(tshape)
    gx = gw = None
    if ctx.needs_input_grad[0]:
        gx = th.bmm(w.t().unsqueeze(0).expand(B, -1, -1), gz)
    if ctx.needs_input_grad[1]:
        gw = th.bmm(gz, x.transpose(1, 2)).sum(0)
    return gx, gw, gbias, gt


def pointwise_chain_supported(x, couts):
    """True when `PointwiseChain` applies: planar fp32 GPU activations x [B, cin, hw], 2 or 3 layers of at most 128
    channels each, no autocast (csrc/pointwise_chain.hip)."""
    import ctypes
    if not (x.is_cuda and x.dtype == th.float32 and x.dim() == 3 and x.numel() > 0 and x.is_contiguous()):
        return False
    if knob("SBMC_PW_CHAIN") == 0 or not _pw_split_enabled() or th.is_autocast_enabled():
        return False
    n = len(couts)
    arr = (ctypes.c_int * max(n, 1))(*couts)
    return (x.data_ptr() % 16 == 0 and x.shape[0] <= 65535
            and bool(_lib.lib().sbmc_pointwise_chain_supported(x.shape[1], n, arr, x.shape[2])))


def pointwise_chain_forward(x, t, s, layers, store_mid=False, want_signs=False, mean=False, want_amax=True):
    """The raw launch (no autograd): `layers` = [(w [cout, cin], bias [cout], act, slope), ...], 2 or 3 of them.
    -> (ys, signs, amaxes, ymean): ys[l] = the l-th layer's output [B, cout_l, hw] (None for an intermediate layer
    unless store_mid), signs[l] its sign words (want_signs, activated layers with a stored output), amaxes[l] the device
    word of max |y_l|, ymean the mean of the last output over groups of s images (mean=True)."""
    import ctypes
    L = _lib.lib()
    nl = len(layers)
    B, cin, hw = x.shape
    dev = x.device
    t_mode = 0
    if t is not None:
        t = t.contiguous()
        t_mode = 2 if (t.dim() == 3 and t.shape[2] == hw and hw > 1) else 1
    ws = [w.contiguous() for (w, _, _, _) in layers]
    bs = [b.contiguous() for (_, b, _, _) in layers]
    _require_f32("PointwiseChain", x=x, t=t, **{"w%d" % i: w for i, w in enumerate(ws)})
    couts = [w.shape[0] for w in ws]
    ys, signs, amaxes = [], [], []
    for l in range(nl):
        last = l + 1 == nl
        ys.append(th.empty(B, couts[l], hw, dtype=th.float32, device=dev) if (last or store_mid) else None)
        signs.append(th.empty(B, couts[l], (hw + 31) // 32, dtype=th.int32, device=dev)
                     if (want_signs and ys[l] is not None and layers[l][2] != 0) else None)
        amaxes.append(amax_word(dev) if want_amax else None)
    ymean = th.empty(B // s, couts[-1], hw, dtype=th.float32, device=dev) if mean else None
    # the context term's largest magnitude (it enters the bound the intermediate activations are scaled by): the producer's
    # word where it left one, else one pass over t (1 / S of an activation's bytes)
    tmax = ensure_amax(t) if t is not None else None

    def parr(ts):
        return (ctypes.c_void_p * nl)(*[None if q is None else q.data_ptr() for q in ts])
    with th.cuda.device(dev), _timed("pointwise_chain_fwd %s<-%d%s" % ("x".join(str(c) for c in couts), cin,
                                                                        "" if store_mid else " (inference)"), dev):
        rc = L.sbmc_pointwise_chain_fwd_f32(
            _lib.ptr(x), _lib.ptr(t) if t is not None else None, _lib.ptr(tmax) if t is not None else None,
            parr(ws), parr(bs), parr(ys), parr(signs), parr(amaxes),
            _lib.ptr(ymean) if ymean is not None else None, nl, (ctypes.c_int * nl)(*couts),
            (ctypes.c_int * nl)(*[a for (_, _, a, _) in layers]), (ctypes.c_float * nl)(*[float(sl) for (_, _, _, sl) in layers]),
            B, s, cin, hw, t_mode, _lib.current_stream(dev))
    _lib.check(rc, "pointwise_chain_fwd")
    return ys, signs, amaxes, ymean


class _LayerCtx(object):
    """What `_pointwise_backward` reads of a `PointwiseLayer` context, for one layer of a `PointwiseChain`."""

    def __init__(self, cfg, saved, has_signs, xmax, needs):
        self.cfg, self.saved_tensors, self.has_signs, self.xmax, self.needs_input_grad = cfg, saved, has_signs, xmax, needs
        self.half = False


class PointwiseChain(th.autograd.Function):
    """Two or three 1x1-convolution layers of at most 128 channels in ONE forward pass (csrc/pointwise_chain.hip):
    y = act_n(w_n .. act_1(w_1 x + b_1 + t) .. + b_n) on planar activations x [B, cin, hw] -- the reference's per-sample
    embeddings and the first two layers of its kernel regressor (sbmc/models.py:79-102, 147-153, 171-177, 196-199).  A
    tile's intermediate activations stay in LDS; they are written to HBM only when a backward will want them and never
    read back by the forward.  The backward runs layer by layer on the kernels of `PointwiseLayer`, from the tensors
    this forward left exactly as the separate layers would have.

    apply(x, t, s, mean, cfg, w_1, b_1, ..., w_n, b_n): t / s as `PointwiseLayer` (the FIRST layer's context term);
    mean: also return the mean of the output over groups of s images (-> (y, ymean)); cfg = ((act, slope), ...).
    """

    @staticmethod
    def forward(ctx, x, t, s, mean, cfg, *wb):
        nl = len(cfg)
        layers = [(wb[2 * l].contiguous(), wb[2 * l + 1].contiguous(), cfg[l][0], cfg[l][1]) for l in range(nl)]
        train = any(ctx.needs_input_grad)
        ctx.xmax = known_amax(x)
        x = x.contiguous()
        ys, signs, amaxes, ymean = pointwise_chain_forward(x, t, s, layers, store_mid=train, want_signs=train, mean=mean)
        B, cin, hw = x.shape
        t_mode = 0
        if t is not None:
            t_mode = 2 if (t.dim() == 3 and t.shape[2] == hw and hw > 1) else 1
        ctx.nl, ctx.s, ctx.mean, ctx.cfg_layers, ctx.t_mode = nl, s, mean, cfg, t_mode
        ctx.tshape = None if t is None else tuple(t.shape)
        ctx.amaxes = amaxes
        ctx.has_signs = [sg is not None for sg in signs]
        if train:
            # per layer: its input, its weight, its sign words (or -- a linear layer -- nothing)
            saved = [x]
            for l in range(nl):
                saved += [layers[l][0], signs[l]]
                if l + 1 < nl:
                    saved.append(ys[l])
            ctx.save_for_backward(*saved)
        tag_amax(ys[-1], amaxes[-1])
        if mean:
            tag_amax(ymean, amaxes[-1])          # (a mean of values of y: the bound holds)
            return ys[-1], ymean
        return ys[-1]

    @staticmethod
    def backward(ctx, gy, gmean=None):
        nl, s = ctx.nl, ctx.s
        saved = ctx.saved_tensors
        x = saved[0]
        inputs, weights, sgs = [x], [], []
        i = 1
        for l in range(nl):
            weights.append(saved[i])
            sgs.append(saved[i + 1])
            i += 2
            if l + 1 < nl:
                inputs.append(saved[i])
                i += 1
        if gy is None:                            # only the mean was used
            gy = (gmean / s).repeat_interleave(s, 0)
            gmean = None
        grads = [None] * (2 * nl)
        gt = None
        g = gy
        for l in range(nl - 1, -1, -1):
            if l == 1:
                pair = _chain_pair_backward(ctx, g if gmean is None or nl > 2 else None, inputs, weights, sgs)
                if pair is not None:
                    g, gt, grads[2], grads[3], grads[0], grads[1] = pair
                    break
            act, slope = ctx.cfg_layers[l]
            first = l == 0
            lctx = _LayerCtx((s if first else 1, act, slope, ctx.t_mode if first else 0, ctx.tshape if first else None),
                             (inputs[l], weights[l], sgs[l]), ctx.has_signs[l],
                             ctx.xmax if first else ctx.amaxes[l - 1],
                             (True if not first else ctx.needs_input_grad[0], True, True, True))
            if l + 1 < nl:
                tag = ctx.amaxes[l]
                if tag is not None and known_amax(inputs[l + 1]) is None:
                    tag_amax(inputs[l + 1], tag)
            gm = gmean if (l == nl - 1) else None
            gx, gw, gb, gtl = _pointwise_backward(lctx, g, gm, s if gm is not None else 1)
            grads[2 * l], grads[2 * l + 1] = gw, gb
            if first:
                gt = gtl
            g = gx
        return (g, gt, None, None, None) + tuple(grads)


def _chain_pair_backward(ctx, g, inputs, weights, sgs):
    """Layers 1 and 0 of a `PointwiseChain` in ONE backward pass (csrc/pointwise_chain_bwd.hip: the gradient of layer 0's
    output stays on the chip) -> (gx, gt, gw1, gb1, gw0, gb0), or None where the pass does not apply (then layer by layer):
    both layers 128 wide, the magnitude words of the incoming gradient and of both layers' inputs known, layer 1's sign words
    there (or it is linear)."""
    # OFF by default (SBMC_PW_CHAIN_BWD=1 switches it on): correct, but at 720p x 8 spp the pass measures 7.2 ms against the two
    # separate passes' 5.9 -- two layers' weight-gradient accumulators (64 registers) next to two layers' operand planes (64)
    # leave a wave of an 8-wave workgroup ~60 registers short, and the spilled registers travel through the same
    # vector-memory queue as the tile's requests and stores (profiles/HISTORY.md, round 6)
    if g is None or knob("SBMC_PW_CHAIN_BWD", 0) == 0 or knob("SBMC_HIP_PW_F16") == 0 or not _pw_split_enabled():
        return None
    L = _lib.lib()
    x, y0 = inputs[0], inputs[1]
    w0, w1 = weights[0], weights[1]
    B, cin, hw = x.shape
    (act0, slope0), (act1, slope1) = ctx.cfg_layers[0], ctx.cfg_layers[1]
    if not (w0.shape[0] == 128 and tuple(w1.shape) == (128, 128) and L.sbmc_pointwise_chain_bwd_supported(cin, hw)
            and g.dtype == th.float32 and (act1 == 0 or sgs[1] is not None)):
        return None
    g = g.contiguous()
    gmax, xbmax, xamax = known_amax(g), ctx.amaxes[0], ctx.xmax
    if gmax is None or xbmax is None:
        return None
    if xamax is None:
        if ctx.needs_input_grad[0]:
            return None                           # (an input that wants a gradient and carries no word: not a network input)
        xamax = ensure_amax(x)                    # a network input (the first embedding): one pass over it, tagged for the step
    s, t_mode, dev = ctx.s, ctx.t_mode, g.device
    dx = bool(ctx.needs_input_grad[0])
    groups = L.sbmc_pointwise_chain_bwd_groups(B, s, t_mode, hw)
    nb = B // s if t_mode == 1 else 1
    gx = th.empty_like(x) if dx else None
    gwp1, gbp1 = w1.new_empty(groups, 128, 128), w1.new_empty(groups, 128)
    gwp0, gbp0 = w0.new_empty(groups, 128, cin), w0.new_empty(groups, nb, 128)
    gt = w0.new_empty(ctx.tshape) if t_mode == 2 else None
    gxmax = amax_word(dev) if dx else None
    with th.cuda.device(dev), _timed("pointwise_chain_bwd 128x128<-%d%s" % (cin, "" if dx else " (no gx)"), dev):
        rc = L.sbmc_pointwise_chain_bwd_f32(
            _lib.ptr(g), _lib.ptr(sgs[1]) if act1 != 0 else None, _lib.ptr(y0), _lib.ptr(w1), _lib.ptr(x), _lib.ptr(w0),
            _lib.ptr(gx) if dx else None, _lib.ptr(gwp1), _lib.ptr(gbp1), _lib.ptr(gwp0), _lib.ptr(gbp0),
            _lib.ptr(gt) if gt is not None else None, _lib.ptr(gmax), _lib.ptr(xbmax), _lib.ptr(xamax),
            _lib.ptr(gxmax) if gxmax is not None else None, B, s, cin, hw, t_mode, act1, slope1, act0, slope0,
            _lib.current_stream(dev))
    _lib.check(rc, "pointwise_chain_bwd")
    per_image = gbp0.sum(0)                       # [nb, 128]
    if t_mode == 1:
        gt = per_image.view(ctx.tshape)
    if gxmax is not None:
        tag_amax(gx, gxmax)
    return gx, gt, gwp1.sum(0), gbp1.sum(0), gwp0.sum(0), per_image.sum(0)


def upsample_cat_supported(coarse, left, top=0, bot=0):
    """True when `UpsampleCat` applies: fp32 GPU tensors, `left` exactly twice the size of `coarse` (minus
    its `top` + `bot` halo rows in the row-slab form)."""
    return (coarse.is_cuda and left.is_cuda and coarse.dtype == th.float32 and left.dtype == th.float32
            and coarse.dim() == 4 and left.dim() == 4 and coarse.shape[0] == left.shape[0]
            and top in (0, 1) and bot in (0, 1)
            and left.shape[2] == 2 * (coarse.shape[2] - top - bot) and left.shape[3] == 2 * coarse.shape[3]
            and coarse.numel() > 0 and not th.is_autocast_enabled()
            # planar tensors (NHWC: UpsampleCatNHWC).  Row-cropped views of planar maps -- what every rank of
            # the planar sharded U-net holds -- qualify: the forward makes its operands contiguous itself
            and not _is_channels_last(coarse) and not _is_channels_last(left)
            and bool(_lib.lib().sbmc_upsample2x_cat_supported(coarse.shape[2] - top - bot, coarse.shape[3])))


class UpsampleCat(th.autograd.Function):
    """th.cat([F.interpolate(coarse, scale_factor=2, mode="bilinear", align_corners=False), left], 1)
    in one pass (csrc/resample.hip); backward: a gather for the upsampling adjoint (PyTorch scatters
    with atomics), the gradient of `left` is a channel slice of the incoming gradient.
    top, bot (0 or 1): row-slab form -- `coarse` carries that many halo rows of the neighbouring slabs
    above / below, the result covers this slab's own rows only (sbmc_amd/dist.py)."""

    @staticmethod
    def forward(ctx, coarse, left, top=0, bot=0):
        _require_f32("UpsampleCat", coarse=coarse, left=left)
        coarse = coarse.contiguous()
        left = left.contiguous()
        b, cu, hc, w = coarse.shape
        cl = left.shape[1]
        h = hc - top - bot
        out = coarse.new_empty(b, cu + cl, 2 * h, 2 * w)
        dev = coarse.device
        with th.cuda.device(dev):
            rc = _lib.lib().sbmc_upsample2x_cat_slab_fwd_f32(_lib.ptr(coarse), _lib.ptr(left), _lib.ptr(out),
                                                             b, cu, cl, hc, w, top, bot, _lib.current_stream(dev))
        _lib.check(rc, "upsample2x_cat_fwd")
        ctx.dims = (b, cu, cl, hc, w, top, bot)
        return out

    @staticmethod
    def backward(ctx, g):
        b, cu, cl, hc, w, top, bot = ctx.dims
        g = g.contiguous()
        gcoarse = None
        if ctx.needs_input_grad[0]:
            gcoarse = g.new_empty(b, cu, hc, w)
            dev = g.device
            with th.cuda.device(dev):
                rc = _lib.lib().sbmc_upsample2x_cat_slab_bwd_f32(_lib.ptr(g), _lib.ptr(gcoarse), b, cu, cl, hc, w,
                                                                 top, bot, _lib.current_stream(dev))
            _lib.check(rc, "upsample2x_cat_bwd")
        return gcoarse, (g[:, cu:] if ctx.needs_input_grad[1] else None), None, None


def _is_channels_last(t):
    """A 4-d tensor whose memory order is [b, h, w, c] (and not also plain contiguous in [b, c, h, w])."""
    return (t.dim() == 4 and t.is_contiguous(memory_format=th.channels_last)
            and (t.shape[1] > 1 and t.shape[2] * t.shape[3] > 1))


class ToChannelsLast(th.autograd.Function):
    """x [b, c, h, w] planar -> the same tensor in channels-last memory order (and back in the backward), by
    the LDS-tile transpose of csrc/nhwc_ops.hip instead of torch's generic strided copy."""

    @staticmethod
    def supported(x):
        """fp32, or fp16 (the U-nets under torch.autocast(float16); no magnitude word there: want_amax must be False)"""
        return (x.is_cuda and x.dtype in (th.float32, th.float16) and x.dim() == 4 and x.is_contiguous() and x.numel() > 0
                and x.shape[1] % 4 == 0 and (x.shape[2] * x.shape[3]) % 4 == 0 and x.data_ptr() % 16 == 0)

    @staticmethod
    def forward(ctx, x, want_amax=False):
        """want_amax: also returns the device word with max |x| that the pass finds on its way (`tag_amax`)."""
        b, c, h, w = x.shape
        out = th.empty(b, c, h, w, dtype=x.dtype, device=x.device, memory_format=th.channels_last)
        dev = x.device
        if x.dtype == th.float16:
            if want_amax:
                raise TypeError("ToChannelsLast: no magnitude word for half tensors")
            with th.cuda.device(dev):
                _lib.check(_lib.lib().sbmc_transpose2d_f16(_lib.ptr(x), _lib.ptr(out), b, c, h * w, _lib.current_stream(dev)),
                           "transpose2d_f16")
            return out
        amax = amax_word(dev) if want_amax else None
        with th.cuda.device(dev):
            if want_amax:
                rc = _lib.lib().sbmc_transpose2d_amax_f32(_lib.ptr(x), _lib.ptr(out), _lib.ptr(amax), b, c, h * w,
                                                          _lib.current_stream(dev))
            else:
                rc = _lib.lib().sbmc_transpose2d_f32(_lib.ptr(x), _lib.ptr(out), b, c, h * w, _lib.current_stream(dev))
        _lib.check(rc, "transpose2d")
        if want_amax:
            ctx.mark_non_differentiable(amax)
            ctx.set_materialize_grads(False)
            return out, amax
        return out

    @staticmethod
    def backward(ctx, g, *_):
        if g is None:
            return None, None
        return FromChannelsLast.apply(g.contiguous(memory_format=th.channels_last)), None


class FromChannelsLast(th.autograd.Function):
    """The inverse: a channels-last [b, c, h, w] tensor -> planar contiguous."""

    @staticmethod
    def supported(x):
        return (x.is_cuda and x.dtype in (th.float32, th.float16) and _is_channels_last(x) and x.numel() > 0
                and x.shape[1] % 4 == 0 and (x.shape[2] * x.shape[3]) % 4 == 0 and x.data_ptr() % 16 == 0)

    @staticmethod
    def forward(ctx, x):
        b, c, h, w = x.shape
        out = th.empty(b, c, h, w, dtype=x.dtype, device=x.device)
        dev = x.device
        if x.dtype == th.float32 and known_amax(x) is None:
            # nobody left a magnitude word on x (a U-net's output, the data gradient of its first convolution): this pass
            # finds it on its way -- the per-sample 1x1 layers behind it scale by it, and without a word each of them ran
            # an absmax pass over the tensor (6 of a training step's 8)
            amax = amax_word(dev)
            with th.cuda.device(dev):
                rc = _lib.lib().sbmc_transpose2d_amax_f32(_lib.ptr(x), _lib.ptr(out), _lib.ptr(amax), b, h * w, c,
                                                          _lib.current_stream(dev))
            _lib.check(rc, "transpose2d")
            tag_amax(out, amax)
            return out
        fn = _lib.lib().sbmc_transpose2d_f16 if x.dtype == th.float16 else _lib.lib().sbmc_transpose2d_f32
        with th.cuda.device(dev):
            rc = fn(_lib.ptr(x), _lib.ptr(out), b, h * w, c, _lib.current_stream(dev))
        _lib.check(rc, "transpose2d")
        return carry_amax(out, x)            # (the same values in another order: the word holds)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        return ToChannelsLast.apply(g) if ToChannelsLast.supported(g) else g.contiguous(memory_format=th.channels_last)


class PoolSkip(th.autograd.Function):
    """left [b, c, h, w] channels-last -> (max_pool2d(left, 2, 2), left): a U-net level's down path AND its skip
    connection as ONE node (reference sbmc/modules.py:300-320: `self.downsample(left)` ... `th.cat([up, left], 1)`), so that
    both gradients of `left` reach one backward: gx = g_skip + g_pooled routed to each window's first maximum, in one
    pass (csrc/nhwc_ops.hip) instead of max_pool_backward + a full-size accumulation pass.  Same values as
    torch's max_pool2d and its backward; no int64 index tensor (the arg-max is recomputed from `left`)."""

    @staticmethod
    def supported(left, pool):
        return (isinstance(pool, th.nn.MaxPool2d) and pool.kernel_size in (2, (2, 2)) and pool.stride in (2, (2, 2))
                and pool.padding in (0, (0, 0)) and pool.dilation in (1, (1, 1)) and not pool.ceil_mode
                and not pool.return_indices and left.is_cuda and left.dtype in (th.float32, th.float16) and left.dim() == 4
                and _is_channels_last(left) and left.shape[1] % 4 == 0 and left.shape[2] % 2 == 0 and left.shape[3] % 2 == 0
                and left.numel() > 0 and left.data_ptr() % 16 == 0
                and knob("SBMC_POOL_SKIP") != 0)

    @staticmethod
    def forward(ctx, left, adj_in=None):
        """adj_in: the `_AdjLink` of the convolution that produced `left`, whose only reader this node then is
        (Conv3x3BiasActNHWC.forward, want_link): the backward pass applies that layer's activation adjoint, sums its bias
        gradient and leaves the gradient's magnitude word -- the `bias_act_nhwc_bwd` pass over the map it has just written."""
        b, c, h, w = left.shape
        ctx.adj_in = None
        if (adj_in is not None and left.dtype == th.float32 and ctx.needs_input_grad[0] and not adj_in.taken
                and tuple(left.shape) == adj_in.shape and left.data_ptr() == adj_in.ptr
                and _lib.lib().sbmc_bias_act_nhwc_supported(c)):
            ctx.adj_in = adj_in
            adj_in.taken = True
        pooled = th.empty((b, c, h // 2, w // 2), dtype=left.dtype, device=left.device, memory_format=th.channels_last)
        dev = left.device
        with th.cuda.device(dev):
            _lib.check(_lib.lib().sbmc_maxpool2_nhwc_fwd(_lib.ptr(left), _lib.ptr(pooled), b, h // 2, w // 2, c,
                                                         left.element_size(), _lib.current_stream(dev)), "maxpool2_nhwc_fwd")
        ctx.save_for_backward(left)
        ctx.set_materialize_grads(False)
        # (a second handle on the same memory with the same strides: returning `left` itself would make autograd re-view
        # it, which loses the channels-last strides of a one-image batch)
        return pooled, left.detach()

    @staticmethod
    def backward(ctx, g_pooled, g_skip):
        (left,) = ctx.saved_tensors
        adj = ctx.adj_in
        if g_pooled is None and adj is None:
            return g_skip, None
        b, c, h, w = left.shape
        L = _lib.lib()
        if g_pooled is None:
            g_pooled = left.new_zeros((b, c, h // 2, w // 2)).contiguous(memory_format=th.channels_last)
        g_pooled = g_pooled.to(left.dtype).contiguous(memory_format=th.channels_last)
        if g_skip is not None:
            g_skip = g_skip.to(left.dtype).contiguous(memory_format=th.channels_last)
        gx = th.empty_like(left, memory_format=th.channels_last)
        dev = left.device
        with th.cuda.device(dev):
            if adj is not None:
                partial = left.new_empty(L.sbmc_bias_act_nhwc_chunks(b * (h // 2) * (w // 2), c), c)
                amax = amax_word(dev)
                _lib.check(L.sbmc_maxpool2_nhwc_bwd_add_adj_f32(_lib.ptr(left), _lib.ptr(g_pooled), _lib.ptr(g_skip), _lib.ptr(gx),
                                                                _lib.ptr(adj.signs), adj.slope, _lib.ptr(partial), _lib.ptr(amax),
                                                                b, h // 2, w // 2, c, _lib.current_stream(dev)),
                           "maxpool2_nhwc_bwd_add_adj")
                tag_amax(gx, amax)
                adj.done = (partial, amax, gx.data_ptr(), gx._version)
            else:
                _lib.check(L.sbmc_maxpool2_nhwc_bwd_add(_lib.ptr(left), _lib.ptr(g_pooled), _lib.ptr(g_skip), _lib.ptr(gx),
                                                        b, h // 2, w // 2, c, left.element_size(),
                                                        _lib.current_stream(dev)), "maxpool2_nhwc_bwd_add")
        return gx, None


class ContextProductNHWC(th.autograd.Function):
    """t[b] = w @ ctx[b] for a CHANNELS-LAST context map ctx [bs, cp, h, w] (memory [bs, h*w, cp]) and w [cout, cp]
    -> t [bs, cout, h*w] planar: the context half of a 1x1 chain's first layer (modules.pointwise_chain_with_context)
    reading the U-net's channels-last result in place.  Backward returns the context gradient channels-last too
    (the GEMM writes [h*w, cp] rows), which is what the U-net's backward wants: no layout copy either way."""

    @staticmethod
    def forward(ctx, context, w):
        bs, cp, h, wd = context.shape
        rows = context.permute(0, 2, 3, 1).reshape(bs, h * wd, cp)          # a view of the channels-last memory
        ctx.save_for_backward(rows, w)
        ctx.dims = (bs, cp, h, wd)
        return th.bmm(w.unsqueeze(0).expand(bs, -1, -1), rows.transpose(1, 2))

    @staticmethod
    def backward(ctx, gt):
        rows, w = ctx.saved_tensors
        bs, cp, h, wd = ctx.dims
        g_context = g_w = None
        if ctx.needs_input_grad[0]:
            g_rows = th.bmm(gt.transpose(1, 2), w.unsqueeze(0).expand(bs, -1, -1))   # [bs, hw, cp]: channels-last
            g_context = g_rows.view(bs, h, wd, cp).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            g_w = _context_weight_grad(gt, rows, w, (bs, cp, h, wd))
        return g_context, g_w


def _context_weight_grad(gt, rows, w, dims):
    """g_w = sum over pixels of gt[:, px] rows[px, :]: a 128 x 128 result over a 921 600-long reduction.  The GEMM
    library takes 1.46 ms for it at 720p (a 32 x 32 x 256 tile, profiles/r03_model_kernel_stats.csv); the fused
    1x1 backward in its weight-gradient-only form streams the same operands at the HBM rate once the context is
    planar: an LDS-tile transpose (0.25 ms) + 0.3 ms."""
    bs, cp, h, wd = dims
    cout, hw = w.shape[0], h * wd
    L = _lib.lib()
    if (gt.is_cuda and gt.dtype == th.float32 and rows.dtype == th.float32 and rows.is_contiguous() and cp % 4 == 0
            and hw % 4 == 0 and L.sbmc_pointwise_bwd_supported(cp, cout, hw)
            and os.environ.get("SBMC_CTX_WGRAD", "fused") == "fused"):
        gt = gt.contiguous()
        dev = gt.device
        planar = th.empty(bs, cp, hw, dtype=rows.dtype, device=dev)
        groups = L.sbmc_pointwise_bwd_groups(bs, 1, 0, hw)
        gwp = w.new_empty(groups, cout, cp)
        gbp = w.new_empty(groups, 1, cout)
        with th.cuda.device(dev):
            rc = L.sbmc_transpose2d_f32(_lib.ptr(rows), _lib.ptr(planar), bs, hw, cp, _lib.current_stream(dev))
            _lib.check(rc, "transpose2d")
            rc = L.sbmc_pointwise_bwd_f32(_lib.ptr(gt), _lib.ptr(gt), _lib.ptr(planar), _lib.ptr(w.contiguous()), None,
                                          _lib.ptr(gwp), _lib.ptr(gbp), None, None, 1, bs, 1, cp, cout, hw, 0, 0, 0.0,
                                          _lib.current_stream(dev))
        _lib.check(rc, "pointwise_bwd (context weight gradient)")
        return gwp.sum(0)
    return th.bmm(gt, rows).sum(0)


class BiasActNHWC(th.autograd.Function):
    """`BiasAct` for a channels-last activation (the U-nets' convolutions in NHWC, csrc/nhwc_ops.hip):
    y [b, c, h, w] with memory order [b, h, w, c], modified in place; backward in one pass as well."""

    @staticmethod
    def supported(y):
        return (y.is_cuda and y.dtype == th.float32 and y.numel() > 0 and _is_channels_last(y)
                and y.data_ptr() % 16 == 0 and bool(_lib.lib().sbmc_bias_act_nhwc_supported(int(y.shape[1]))))

    @staticmethod
    def forward(ctx, y, bias, act, slope, want_amax=False):
        """want_amax: also returns the device word with the bit pattern of max |y| (the pass finds it on the way; the
        3 x 3 convolution that reads y next scales by it: `tag_amax`)."""
        _require_f32("BiasActNHWC", y=y, bias=bias)
        b, c, h, w = y.shape
        bias = bias.contiguous()
        dev = y.device
        # with an activation and a backward to come: one sign bit per element for the adjoint (1/32 of the bytes
        # it would otherwise read back from y -- and y, which later passes overwrite in place, is not kept)
        signs = None
        if act != 0 and any(ctx.needs_input_grad[:2]):
            signs = th.empty((y.numel() + 31) // 32, dtype=th.int32, device=dev)
        amax = amax_word(dev) if want_amax else None
        with th.cuda.device(dev):
            if amax is not None:
                rc = _lib.lib().sbmc_bias_act_nhwc_fwd_amax_f32(_lib.ptr(y), _lib.ptr(bias), _lib.ptr(signs), _lib.ptr(amax),
                                                                b * h * w, c, act, slope, _lib.current_stream(dev))
            elif signs is not None:
                rc = _lib.lib().sbmc_bias_act_nhwc_fwd_signs_f32(_lib.ptr(y), _lib.ptr(bias), _lib.ptr(signs), b * h * w, c,
                                                                 act, slope, _lib.current_stream(dev))
            else:
                rc = _lib.lib().sbmc_bias_act_nhwc_fwd_f32(_lib.ptr(y), _lib.ptr(bias), b * h * w, c, act, slope,
                                                           _lib.current_stream(dev))
        _lib.check(rc, "bias_act_nhwc_fwd")
        ctx.mark_dirty(y)
        ctx.act, ctx.slope = act, slope
        if signs is not None:
            ctx.save_for_backward(signs)
        if amax is not None:
            ctx.mark_non_differentiable(amax)
            ctx.set_materialize_grads(False)
            return y, amax
        return y

    @staticmethod
    def backward(ctx, gy, *_):
        if gy is None:
            return None, None, None, None, None
        gy = gy.contiguous(memory_format=th.channels_last)
        b, c, h, w = gy.shape
        gx = th.empty_like(gy, memory_format=th.channels_last)
        L = _lib.lib()
        partial = gy.new_empty(L.sbmc_bias_act_nhwc_chunks(b * h * w, c), c)
        dev = gy.device
        # (the largest magnitude of gx, for the convolution's data / weight gradient kernels that read it next)
        amax = amax_word(dev)
        with th.cuda.device(dev):
            rc = L.sbmc_bias_act_nhwc_bwd_amax_f32(_lib.ptr(gy), _lib.ptr(ctx.saved_tensors[0]) if ctx.act != 0 else None,
                                                   _lib.ptr(gx), _lib.ptr(partial), _lib.ptr(amax), b * h * w, c, ctx.act,
                                                   ctx.slope, _lib.current_stream(dev))
        _lib.check(rc, "bias_act_nhwc_bwd")
        tag_amax(gx, amax)
        return gx, partial.sum(0), None, None, None


class _StreamK(object):
    """The stream-K workspace of csrc/conv3x3.hip (partial tiles of a launch whose tile count is no multiple of the CU
    count meet there), one per (device, stream): launches on one stream run one after the other.
    SBMC_CONV3X3_STREAMK=0: whole tiles round-robin, as in round 3."""

    def __init__(self):
        self._ws = {}

    def take(self, device):
        """-> workspace pointer, or None"""
        if knob("SBMC_CONV3X3_STREAMK") == 0:
            return None
        key = (device.index, th.cuda.current_stream(device).cuda_stream)
        ent = self._ws.get(key)
        if ent is None:
            with th.cuda.device(device):
                nbytes = _lib.lib().sbmc_conv3x3_workspace_bytes()
            ent = self._ws[key] = th.empty(nbytes, dtype=th.uint8, device=device)
        return _lib.ptr(ent)


_STREAMK = _StreamK()


class _AmaxArena(object):
    """Zeroed device words for the passes that RAISE a word to the largest magnitude of what they write (ABI 5: they no
    longer zero it themselves -- that was one memset launch per pass, ~230 per training step, none of which shrinks
    with the slab of a sharded frame).  One zero-filled block of words per (device, stream) serves the next SIZE
    requests; a word is a view of its block and keeps it alive."""
    SIZE = 2048

    def __init__(self):
        self._blocks = {}

    def take(self, device):
        key = (device.index, th.cuda.current_stream(device).cuda_stream)
        blk = self._blocks.get(key)
        if blk is None or blk[1] >= self.SIZE:
            blk = self._blocks[key] = [th.zeros(self.SIZE, dtype=th.int32, device=device), 0]
        i = blk[1]
        blk[1] = i + 1
        return blk[0][i:i + 1]


_AMAX_ARENA = _AmaxArena()


def amax_word(device):
    """A device word holding 0 (int32 [1]) for a `*_amax` pass to raise."""
    return _AMAX_ARENA.take(device)


def tag_amax(t, amax):
    """Remembers ON THE TENSOR OBJECT the device word `amax` (int32 [1]: bit pattern of a float >= max |t|) that
    the pass which produced `t` found on its way.  Valid only for this very object in this very state:
    `known_amax` checks the version counter (any in-place change bumps it) and the storage address, and whoever
    finds no tag runs csrc/conv3x3.hip's absmax pass instead -- a lost tag costs time, never accuracy."""
    t._sbmc_amax = (amax, t._version, t.data_ptr())
    return t


def known_amax(t):
    if knob("SBMC_AMAX_TAGS") == 0:
        return None
    tag = getattr(t, "_sbmc_amax", None)
    if tag is not None and tag[1] == t._version and tag[2] == t.data_ptr() and tag[0].device == t.device:
        return tag[0]
    return None


def ensure_amax(t):
    """The word of `t`'s largest magnitude: its tag, or -- none there -- the absmax pass of csrc/conv3x3.hip, tagged
    on `t` for whoever asks next."""
    a = known_amax(t)
    if a is None:
        a = Conv3x3NHWC._absmax(t)
        tag_amax(t, a)
    return a


def carry_amax(dst, src):
    """`dst` holds exactly `src`'s values (a view or reshape of the whole of it, a transposition): its word is src's."""
    a = known_amax(src)
    if a is not None and dst.numel() == src.numel():
        tag_amax(dst, a)
    return dst


class _TaggedView(th.autograd.Function):
    """x.view(shape) that hands the magnitude word on -- in BOTH directions: torch's own view nodes make new tensor
    objects, the tag of `tag_amax` rides on the object, and the 1 x 1 layers either side of a reshape (models._embed:
    [bs * spp, c, h, w] <-> [bs, spp, c, h, w] <-> [bs * spp, c, h * w]) take their scales from each other's words."""

    @staticmethod
    def forward(ctx, x, shape):
        ctx.shape = x.shape
        return carry_amax(x.view(shape), x)

    @staticmethod
    def backward(ctx, g):
        if not g.is_contiguous():
            return g.reshape(ctx.shape), None
        return carry_amax(g.view(ctx.shape), g), None


def tagged_view(x, *shape):
    """x.view(*shape); through `_TaggedView` where a magnitude word could travel (contiguous fp32 GPU tensors).
    The result is then a view made inside a custom Function: torch refuses IN-PLACE operations on it (an
    nn.ReLU(inplace=True) behind it raises).  Every caller in this package feeds it to one of its own Functions, which
    apply activations themselves; a new caller that wants to modify the result in place must use x.view() and lose the tag."""
    if x.is_cuda and x.dtype == th.float32 and x.is_contiguous() and knob("SBMC_AMAX_TAGS") != 0:
        return _TaggedView.apply(x, tuple(shape))
    return x.reshape(*shape)


def raise_amax(word, block):
    """word := max(word, largest magnitude of `block`), on the device: `block` a dense run of fp32 values (a few halo
    rows), `word` the magnitude word of the map they are attached to."""
    _lib.check(_lib.lib().sbmc_conv3x3_absmax_raise_f32(_lib.ptr(block), block.numel(), _lib.ptr(word),
                                                        _lib.current_stream(block.device)), "conv3x3_absmax_raise")


def wants_amax(t):
    """Will a 3 x 3 convolution of csrc/conv3x3.hip scale this tensor by its largest magnitude?  (fp32,
    channels-last, the kernels not switched off.)"""
    return (t.is_cuda and t.dtype == th.float32 and t.dim() == 4 and _is_channels_last(t) and t.data_ptr() % 16 == 0
            and knob("SBMC_CONV3X3") != 0
            and knob("SBMC_AMAX_TAGS") != 0)


def bound_amax(*amaxes):
    """An upper bound of the largest magnitude of a tensor built from others without growing any value (max
    pooling, bilinear interpolation, concatenation): the largest of theirs.  None if any is unknown."""
    if not amaxes or any(a is None for a in amaxes):
        return None
    out = amaxes[0]
    for a in amaxes[1:]:
        out = th.maximum(out, a)            # (bit patterns of non-negative floats order like integers)
    return out


class Conv3x3NHWC(th.autograd.Function):
    """3 x 3 convolution (stride 1, zero padding 1, no bias) of a channels-last fp32 activation: the U-nets'
    convolutions (reference sbmc/modules.py:195-320 through ttools' ConvChain -> cuDNN) on csrc/conv3x3.hip --
    fp32 accuracy from three f16 matrix products per term, ~2.8x MIOpen's fp32 solver.  The data gradient is the
    same kernel on the mirrored, transposed weights; the weight gradient a kernel of its own (a GEMM that reduces
    over the pixels).

    SBMC_CONV3X3=0 keeps every convolution on MIOpen."""

    @staticmethod
    def supported(x, conv):
        if knob("SBMC_CONV3X3") == 0:
            return False
        if not (isinstance(conv, th.nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
                and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
                and conv.padding_mode == "zeros"):
            return False
        if not (x.is_cuda and x.dtype == th.float32 and x.numel() > 0 and _is_channels_last(x)
                and x.data_ptr() % 16 == 0 and x.shape[1] == conv.in_channels):
            return False
        b, c, h, w = x.shape
        L = _lib.lib()
        # (the adjoint runs the kernel with the channel counts exchanged)
        return bool(L.sbmc_conv3x3_supported(b, h, w, c, conv.out_channels)
                    and L.sbmc_conv3x3_supported(b, h, w, conv.out_channels, c))

    @staticmethod
    def _prepare(w, flip):
        """The weights' two f16 planes in the kernel's stage order (and their scale), on the device.  A weight that
        comes out of a `wbank.WeightBank` carries both orientations already."""
        wp = getattr(w, "_sbmc_wp", None)
        if wp is not None:
            return wp[1 if flip else 0]
        L = _lib.lib()
        cout, cin = (w.shape[1], w.shape[0]) if flip else (w.shape[0], w.shape[1])
        if w.data_ptr() % 16 or not (w.is_contiguous() or w.is_contiguous(memory_format=th.channels_last)):
            w = w.contiguous()
        s = w.stride()
        wp = th.empty(L.sbmc_conv3x3_weights_bytes(cin, cout), dtype=th.uint8, device=w.device)
        s_co, s_ci = (s[1], s[0]) if flip else (s[0], s[1])
        _lib.check(L.sbmc_conv3x3_prepare_weights_f32(_lib.ptr(w), s_co, s_ci, s[2], s[3], w.numel(), cin, cout,
                                                      1 if flip else 0, _lib.ptr(wp), _lib.current_stream(w.device)),
                   "conv3x3_prepare_weights")
        return wp

    @staticmethod
    def _absmax(x):
        """Bit pattern of the largest magnitude of x, on the device (the kernels derive their power-of-two scale
        from it: no host synchronisation)."""
        out = th.empty(1, dtype=th.int32, device=x.device)
        _lib.check(_lib.lib().sbmc_conv3x3_absmax_f32(_lib.ptr(x), x.numel(), _lib.ptr(out), _lib.current_stream(x.device)),
                   "conv3x3_absmax")
        return out

    @staticmethod
    def _conv(x, xmax, wp, cout):
        """x [b, cin, h, w] in channels-last memory order -> [b, cout, h, w], the same order."""
        b, cin, h, w = x.shape
        y = th.empty((b, cout, h, w), dtype=th.float32, device=x.device, memory_format=th.channels_last)
        ws = _STREAMK.take(x.device)
        _lib.check(_lib.lib().sbmc_conv3x3_nhwc_f32(_lib.ptr(x), _lib.ptr(xmax), _lib.ptr(wp), _lib.ptr(y), b, h, w, cin,
                                                    cout, ws, _lib.current_stream(x.device)), "conv3x3_nhwc")
        return y

    @staticmethod
    def forward(ctx, x, w):
        _require_f32("Conv3x3NHWC", x=x, w=w)
        with th.cuda.device(x.device), _timed("conv3x3_fwd %dx%d@%dx%dx%d" % (w.shape[0], w.shape[1], x.shape[0], x.shape[2], x.shape[3]),
                                              x.device):
            xmax = known_amax(x)
            if xmax is None:
                xmax = Conv3x3NHWC._absmax(x)
            y = Conv3x3NHWC._conv(x, xmax, Conv3x3NHWC._prepare(w, False), w.shape[0])
        ctx.save_for_backward(x, w, xmax)
        ctx.wp = getattr(w, "_sbmc_wp", None)          # (a weight bank's prepared forms: the adjoint's is the second)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, xmax = ctx.saved_tensors
        return Conv3x3NHWC._backward(x, w, xmax, gy.contiguous(memory_format=th.channels_last),
                                     ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.wp)

    @staticmethod
    def _backward(x, w, xmax, gy, want_gx, want_gw, wp=None, bias_partial=None, adj=None):
        """bias_partial: the bias gradient's per-workgroup partial sums [chunks, cout] (bias_act_nhwc_bwd); where the
        weight-gradient kernel runs, its reduction launch adds them up as well and a third value, the bias gradient,
        is returned (else None: the caller sums).
        adj (an `_AdjLink`): x is the activated output of the chain's previous layer and nothing else reads it -- the
        data gradient comes out with that layer's activation adjoint applied (csrc ADJ), and the link tells it so."""
        b, cin, h, wd = x.shape
        cout = w.shape[0]
        gx = gw = gbias = None
        L = _lib.lib()
        dev = gy.device
        with th.cuda.device(dev):
            gmax = known_amax(gy)
            if gmax is None:
                gmax = Conv3x3NHWC._absmax(gy)
            if want_gx and adj is not None:
                with _timed("conv3x3_bwd_data_adj %dx%d@%dx%dx%d" % (cout, cin, b, h, wd), dev):
                    gx = th.empty((b, cin, h, wd), dtype=th.float32, device=dev, memory_format=th.channels_last)
                    apartial = th.zeros((L.sbmc_conv3x3_adj_partial_rows(), cin), dtype=th.float32, device=dev)
                    amax = amax_word(dev)
                    _lib.check(L.sbmc_conv3x3_adj_nhwc_f32(
                        _lib.ptr(gy), _lib.ptr(gmax), _lib.ptr(wp[1] if wp is not None else Conv3x3NHWC._prepare(w, True)),
                        _lib.ptr(adj.signs), adj.slope, _lib.ptr(gx), _lib.ptr(apartial), _lib.ptr(amax), b, h, wd, cout, cin,
                        _STREAMK.take(dev), _lib.current_stream(dev)), "conv3x3_adj_nhwc")
                    tag_amax(gx, amax)
                    adj.done = (apartial, amax, gx.data_ptr(), gx._version)
            elif want_gx:
                with _timed("conv3x3_bwd_data %dx%d@%dx%dx%d" % (cout, cin, b, h, wd), dev):
                    gx = Conv3x3NHWC._conv(gy, gmax, wp[1] if wp is not None else Conv3x3NHWC._prepare(w, True), cin)
            if want_gw:
                with _timed("conv3x3_bwd_weight %dx%d@%dx%dx%d" % (cout, cin, b, h, wd), dev):
                    if (knob("SBMC_CONV3X3_WGRAD") != 0
                            and L.sbmc_conv3x3_wgrad_supported(b, h, wd, cin, cout)):
                        gw = th.empty((cout, cin, 3, 3), dtype=th.float32, device=dev, memory_format=th.channels_last)
                        scratch = th.empty(L.sbmc_conv3x3_wgrad_scratch_bytes(b, h, wd, cin, cout), dtype=th.uint8, device=dev)
                        s = gw.stride()
                        if bias_partial is not None:
                            gbias = th.empty(cout, dtype=th.float32, device=dev)
                        _lib.check(L.sbmc_conv3x3_wgrad_bias_f32(
                            _lib.ptr(gy), _lib.ptr(gmax), _lib.ptr(x), _lib.ptr(xmax), _lib.ptr(gw), s[0], s[1], s[2], s[3],
                            _lib.ptr(scratch), b, h, wd, cin, cout, _lib.ptr(bias_partial),
                            bias_partial.shape[0] if bias_partial is not None else 0, cout, _lib.ptr(gbias),
                            _lib.current_stream(dev)), "conv3x3_wgrad")
                    else:
                        wcl = w.contiguous(memory_format=th.channels_last)      # (MIOpen's NHWC solver)
                        gw = th.ops.aten.convolution_backward(gy, x, wcl, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                              [False, True, False])[1]
        if bias_partial is not None:
            return gx, gw, gbias
        return gx, gw


class Conv3x3BiasActNHWC(th.autograd.Function):
    """`Conv3x3NHWC` with the bias + ReLU / LeakyReLU pass behind it (reference sbmc/modules.py:154-175) in the
    kernel's epilogue: the activated output is written once instead of written, read and rewritten; the epilogue
    also leaves the sign bits the adjoint needs and the largest magnitude the next convolution scales by (returned as
    second output: `tag_amax`).  Backward: the activation's adjoint + bias gradient pass (`bias_act_nhwc_bwd`), then
    `Conv3x3NHWC`'s two gradient kernels."""

    @staticmethod
    def supported(conv):
        """(on top of Conv3x3NHWC.supported) the adjoint of the bias + activation pass takes these output channels"""
        return conv.bias is not None and bool(_lib.lib().sbmc_bias_act_nhwc_supported(int(conv.out_channels)))

    @staticmethod
    def adj_link_for(y):
        """The link `forward(..., want_link=True)` left on its output (or None): hand it to the ONE layer that
        consumes that output as `adj_in`."""
        return getattr(y, "_sbmc_adj_link", None)

    @staticmethod
    def forward(ctx, x, w, bias, act, slope, adj_in=None, want_link=False):
        """adj_in / want_link (ABI 7, csrc ADJ): two layers of a CHAIN -- `x` is the activated output of another
        Conv3x3BiasActNHWC that NOTHING ELSE reads (the caller's guarantee: modules.ConvChain._run between its own
        layers) -- share an `_AdjLink`: the producer's forward leaves it on its output (want_link), the consumer takes
        it (adj_in) and its backward runs the data gradient with the producer's activation adjoint + bias sums in the
        epilogue; the producer's backward then finds its gradient finished and skips its `bias_act_nhwc_bwd` pass (one
        read and one write of the gradient less per layer)."""
        _require_f32("Conv3x3BiasActNHWC", x=x, w=w, bias=bias)
        L = _lib.lib()
        b, cin, h, wd = x.shape
        cout = w.shape[0]
        dev = x.device
        bias = bias.contiguous()
        need_grad = any(ctx.needs_input_grad[:3])
        with th.cuda.device(dev), _timed("conv3x3_fwd %dx%d@%dx%dx%d" % (cout, cin, b, h, wd), dev):
            xmax = known_amax(x)
            if xmax is None:
                xmax = Conv3x3NHWC._absmax(x)
            wp = Conv3x3NHWC._prepare(w, False)
            y = th.empty((b, cout, h, wd), dtype=th.float32, device=dev, memory_format=th.channels_last)
            signs = th.empty((y.numel() + 31) // 32, dtype=th.int32, device=dev) if (act != 0 and need_grad) else None
            amax = amax_word(dev)
            ws = _STREAMK.take(dev)
            _lib.check(L.sbmc_conv3x3_bias_act_nhwc_f32(_lib.ptr(x), _lib.ptr(xmax), _lib.ptr(wp), _lib.ptr(bias), _lib.ptr(y),
                                                        _lib.ptr(signs), _lib.ptr(amax), b, h, wd, cin, cout, act, slope,
                                                        ws, _lib.current_stream(dev)), "conv3x3_bias_act_nhwc")
        ctx.act, ctx.slope = act, slope
        ctx.wp = getattr(w, "_sbmc_wp", None)
        ctx.save_for_backward(x, w, xmax, signs if signs is not None else xmax)
        ctx.mark_non_differentiable(amax)
        ctx.set_materialize_grads(False)         # (else autograd zero-fills a "gradient" for the amax word every step)
        # the chain: what this layer's data gradient does for the layer before it (adj_in), and what the layer behind it
        # will do for this one (link_out)
        ctx.adj_in = adj_in if (adj_in is not None and ctx.needs_input_grad[0] and adj_in.fits(x, L, cout)) else None
        if ctx.adj_in is not None:
            ctx.adj_in.taken = True
        ctx.link_out = None
        if want_link and signs is not None and act in (1, 2) and knob("SBMC_CONV3X3_ADJ") != 0:
            ctx.link_out = y._sbmc_adj_link = _AdjLink(signs, 0.0 if act == 1 else float(slope), y)
        return y, amax

    @staticmethod
    def backward(ctx, gy, _gamax):
        if gy is None:
            return None, None, None, None, None, None, None
        x, w, xmax, signs = ctx.saved_tensors
        gy = gy.contiguous(memory_format=th.channels_last)
        b, cout, h, wd = gy.shape
        L = _lib.lib()
        dev = gy.device
        done = ctx.link_out.collect(gy) if ctx.link_out is not None else None
        if done is not None:
            # the layer behind this one has applied the adjoint in its data gradient's epilogue: gy IS gz
            gz, partial, gmax = gy, done[0], done[1]
        else:
            gz = th.empty_like(gy, memory_format=th.channels_last)
            partial = gy.new_empty(L.sbmc_bias_act_nhwc_chunks(b * h * wd, cout), cout)
            gmax = amax_word(dev)
            with th.cuda.device(dev):
                _lib.check(L.sbmc_bias_act_nhwc_bwd_amax_f32(_lib.ptr(gy), _lib.ptr(signs) if ctx.act != 0 else None, _lib.ptr(gz),
                                                             _lib.ptr(partial), _lib.ptr(gmax), b * h * wd, cout, ctx.act, ctx.slope,
                                                             _lib.current_stream(dev)), "bias_act_nhwc_bwd")
        tag_amax(gz, gmax)
        want_bias = ctx.needs_input_grad[2]
        res = Conv3x3NHWC._backward(x, w, xmax, gz, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.wp,
                                    partial if want_bias else None, ctx.adj_in)
        gbias = res[2] if want_bias else None
        if want_bias and gbias is None:
            gbias = partial.sum(0)               # (the weight-gradient kernel did not run: its reduction adds them up)
        return res[0], res[1], gbias, None, None, None, None


class _AdjLink(object):
    """What two consecutive layers of a convolution chain share (Conv3x3BiasActNHWC.forward, adj_in / want_link): the
    producer's sign words and slope for the consumer's data-gradient epilogue, and -- in the backward pass -- the
    consumer's word that the producer's gradient arrives finished, with the bias gradient's partial sums and the
    gradient's magnitude word."""

    def __init__(self, signs, slope, y):
        self.signs, self.slope = signs, slope
        self.shape, self.ptr = tuple(y.shape), y.data_ptr()
        self.taken = False            # a consumer's forward has taken the link
        self.done = None              # (partial, gmax, gz pointer, gz version) between the consumer's backward and the producer's

    def fits(self, x, L, cout_consumer):
        """The consumer's input IS the producer's output, once, and the ADJ kernel takes the shape."""
        b, c, h, w = x.shape
        return (not self.taken and tuple(x.shape) == self.shape and x.data_ptr() == self.ptr
                and bool(L.sbmc_conv3x3_adj_supported(b, h, w, cout_consumer, c)))

    def collect(self, gy):
        """Producer's backward: (partial, gmax) if `gy` is the finished gradient the consumer's backward left."""
        done, self.done = self.done, None
        if done is None:
            return None
        # (the pointer alone is not enough: autograd may add a second reader's gradient into that very buffer in place --
        # the version counter then moves)
        if done[2] != gy.data_ptr() or done[3] != gy._version:
            raise RuntimeError("Conv3x3BiasActNHWC: the gradient of a chained layer's output was replaced on its way "
                               "(another consumer of that output, or a hook): the caller's guarantee does not hold")
        return done[0], done[1]


class Conv3x3BiasActHalfNHWC(th.autograd.Function):
    """The U-nets' 3 x 3 convolution + bias + ReLU / LeakyReLU on HALF activations ("fp16 activations", BASELINE
    configs[4]; reference sbmc/modules.py:154-175 under torch.autocast(float16) semantics: half inputs, the fp32 weight
    rounded to half once, fp32 accumulation, half output) on csrc/conv3x3.hip's kernels in their one-plane form: ONE f16
    matrix product per term where the fp32 form issues three, no scales, no absmax.  x [b, cin, h, w] float16
    channels-last, w fp32 (a weight bank's weight carries its prepared forms), bias fp32 -> y float16 channels-last.
    Backward: activation adjoint + bias partial sums in one pass over the half gradient, data gradient by the same
    kernel on the mirrored weights, weight gradient (fp32) by the one-plane weight-gradient kernel."""

    @staticmethod
    def supported(x, conv):
        if knob("SBMC_CONV3X3") == 0 or knob("SBMC_CONV3X3_HALF") == 0:
            return False
        if not (isinstance(conv, th.nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
                and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
                and conv.padding_mode == "zeros" and conv.bias is not None):
            return False
        if not (x.is_cuda and x.dtype == th.float16 and x.numel() > 0 and _is_channels_last(x)
                and x.data_ptr() % 16 == 0 and x.shape[1] == conv.in_channels):
            return False
        b, c, h, w = x.shape
        L = _lib.lib()
        return bool(L.sbmc_conv3x3_supported(b, h, w, c, conv.out_channels)
                    and L.sbmc_conv3x3_supported(b, h, w, conv.out_channels, c)
                    and L.sbmc_bias_act_nhwc_supported(int(conv.out_channels)))

    @staticmethod
    def forward(ctx, x, w, bias, act, slope):
        L = _lib.lib()
        b, cin, h, wd = x.shape
        cout = w.shape[0]
        dev = x.device
        wf = w if w.dtype == th.float32 else w.float()
        bias = bias.float().contiguous()
        need_grad = any(ctx.needs_input_grad[:3])
        with th.cuda.device(dev), _timed("conv3x3_f16_fwd %dx%d@%dx%dx%d" % (cout, cin, b, h, wd), dev):
            wp = Conv3x3NHWC._prepare(wf, False)
            y = th.empty((b, cout, h, wd), dtype=th.float16, device=dev, memory_format=th.channels_last)
            signs = th.empty((y.numel() + 31) // 32, dtype=th.int32, device=dev) if (act != 0 and need_grad) else None
            ws = _STREAMK.take(dev)
            _lib.check(L.sbmc_conv3x3_bias_act_nhwc_f16(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(bias), _lib.ptr(y), _lib.ptr(signs),
                                                        b, h, wd, cin, cout, act, slope, ws, _lib.current_stream(dev)),
                       "conv3x3_bias_act_nhwc_f16")
        ctx.act, ctx.slope = act, slope
        ctx.wp = getattr(wf, "_sbmc_wp", None)
        ctx.save_for_backward(x, wf, signs if signs is not None else x)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, signs = ctx.saved_tensors
        if gy.dtype != th.float16:
            gy = gy.half()
        gy = gy.contiguous(memory_format=th.channels_last)
        b, cout, h, wd = gy.shape
        cin = x.shape[1]
        L = _lib.lib()
        dev = gy.device
        want_gx, want_gw, want_gb = ctx.needs_input_grad[:3]
        gx = gw = gbias = None
        with th.cuda.device(dev):
            st = _lib.current_stream(dev)
            gz = th.empty_like(gy, memory_format=th.channels_last)
            partial = th.empty(L.sbmc_bias_act_nhwc_chunks(b * h * wd, cout), cout, dtype=th.float32, device=dev)
            _lib.check(L.sbmc_bias_act_nhwc_bwd_signs_f16(_lib.ptr(gy), _lib.ptr(signs) if ctx.act != 0 else None, _lib.ptr(gz),
                                                          _lib.ptr(partial), b * h * wd, cout, ctx.act, ctx.slope, st),
                       "bias_act_nhwc_bwd_f16")
            if want_gx:
                with _timed("conv3x3_f16_bwd_data %dx%d@%dx%dx%d" % (cout, cin, b, h, wd), dev):
                    wpb = ctx.wp[1] if ctx.wp is not None else Conv3x3NHWC._prepare(w, True)
                    gx = th.empty((b, cin, h, wd), dtype=th.float16, device=dev, memory_format=th.channels_last)
                    ws = _STREAMK.take(dev)
                    _lib.check(L.sbmc_conv3x3_nhwc_f16(_lib.ptr(gz), _lib.ptr(wpb), _lib.ptr(gx), b, h, wd, cout, cin, ws, st),
                               "conv3x3_nhwc_f16 (data gradient)")
            if want_gw:
                with _timed("conv3x3_f16_bwd_weight %dx%d@%dx%dx%d" % (cout, cin, b, h, wd), dev):
                    if L.sbmc_conv3x3_wgrad_supported(b, h, wd, cin, cout):
                        gw = th.empty((cout, cin, 3, 3), dtype=th.float32, device=dev, memory_format=th.channels_last)
                        scratch = th.empty(L.sbmc_conv3x3_wgrad_scratch_bytes(b, h, wd, cin, cout), dtype=th.uint8, device=dev)
                        s = gw.stride()
                        if want_gb:
                            gbias = th.empty(cout, dtype=th.float32, device=dev)
                        _lib.check(L.sbmc_conv3x3_wgrad_bias_f16(
                            _lib.ptr(gz), _lib.ptr(x), _lib.ptr(gw), s[0], s[1], s[2], s[3], _lib.ptr(scratch), b, h, wd, cin,
                            cout, _lib.ptr(partial) if want_gb else None, partial.shape[0] if want_gb else 0, cout,
                            _lib.ptr(gbias), st), "conv3x3_wgrad_f16")
                    else:
                        wcl = w.half().contiguous(memory_format=th.channels_last)      # (MIOpen's half NHWC solver)
                        gw = th.ops.aten.convolution_backward(gz, x, wcl, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                              [False, True, False])[1].float()
        if want_gb and gbias is None:
            gbias = partial.sum(0)
        return gx, gw, gbias, None, None


def upsample_cat_nhwc_supported(coarse, left, top=0, bot=0):
    """fp32, or fp16 (the U-nets under torch.autocast(float16)) channels-last tensors of one dtype."""
    return (coarse.is_cuda and left.is_cuda and coarse.dtype in (th.float32, th.float16) and left.dtype == coarse.dtype
            and coarse.dim() == 4 and left.dim() == 4 and coarse.shape[0] == left.shape[0]
            and top in (0, 1) and bot in (0, 1)
            and left.shape[2] == 2 * (coarse.shape[2] - top - bot) and left.shape[3] == 2 * coarse.shape[3]
            and coarse.numel() > 0
            and _is_channels_last(coarse) and _is_channels_last(left)
            and coarse.data_ptr() % 16 == 0 and left.data_ptr() % 16 == 0
            and bool(_lib.lib().sbmc_upsample2x_cat_nhwc_supported(coarse.shape[1], left.shape[1],
                                                                   coarse.shape[2] - top - bot, coarse.shape[3])))


class UpsampleCatNHWC(th.autograd.Function):
    """`UpsampleCat` on channels-last tensors (result channels-last too), float32 or float16 storage: one pass
    per direction, the backward also delivers the skip connection's gradient as a contiguous tensor.
    top, bot: row-slab form, as for `UpsampleCat`."""

    @staticmethod
    def forward(ctx, coarse, left, top=0, bot=0, adj_in=None):
        """adj_in: the `_AdjLink` of the convolution that produced `coarse`, whose only reader this node then is: the
        backward pass applies that layer's activation adjoint to the coarse map's gradient as it writes it (whole frames,
        fp32)."""
        if not (coarse.is_cuda and coarse.dtype in (th.float32, th.float16) and left.dtype == coarse.dtype):
            raise TypeError("UpsampleCatNHWC: float32 or float16 GPU tensors of one dtype expected")
        b, cu, hc, w = coarse.shape
        ctx.adj_in = None
        if (adj_in is not None and coarse.dtype == th.float32 and top == 0 and bot == 0 and ctx.needs_input_grad[0]
                and not adj_in.taken and tuple(coarse.shape) == adj_in.shape and coarse.data_ptr() == adj_in.ptr
                and _lib.lib().sbmc_bias_act_nhwc_supported(cu)):
            ctx.adj_in = adj_in
            adj_in.taken = True
        cl = left.shape[1]
        h = hc - top - bot
        out = th.empty(b, cu + cl, 2 * h, 2 * w, dtype=coarse.dtype, device=coarse.device,
                       memory_format=th.channels_last)
        dev = coarse.device
        L = _lib.lib()
        fwd = L.sbmc_upsample2x_cat_nhwc_slab_fwd_f16 if coarse.dtype == th.float16 else L.sbmc_upsample2x_cat_nhwc_slab_fwd_f32
        with th.cuda.device(dev):
            rc = fwd(_lib.ptr(coarse), _lib.ptr(left), _lib.ptr(out), b, cu, cl, hc, w, top, bot,
                     _lib.current_stream(dev))
        _lib.check(rc, "upsample2x_cat_nhwc_fwd")
        ctx.dims = (b, cu, cl, hc, w, top, bot)
        return out

    @staticmethod
    def backward(ctx, g):
        b, cu, cl, hc, w, top, bot = ctx.dims
        h = hc - top - bot
        g = g.contiguous(memory_format=th.channels_last)
        gcoarse = gleft = None
        if ctx.needs_input_grad[0]:
            gcoarse = th.empty(b, cu, hc, w, dtype=g.dtype, device=g.device, memory_format=th.channels_last)
        if ctx.needs_input_grad[1]:
            gleft = th.empty(b, cl, 2 * h, 2 * w, dtype=g.dtype, device=g.device, memory_format=th.channels_last)
        if gcoarse is not None or gleft is not None:
            dev = g.device
            L = _lib.lib()
            bwd = L.sbmc_upsample2x_cat_nhwc_slab_bwd_f16 if g.dtype == th.float16 else L.sbmc_upsample2x_cat_nhwc_slab_bwd_f32
            adj = ctx.adj_in if (gcoarse is not None and g.dtype == th.float32) else None
            with th.cuda.device(dev):
                if adj is not None:
                    partial = g.new_empty(L.sbmc_bias_act_nhwc_chunks(b * hc * w, cu), cu)
                    amax = amax_word(dev)
                    rc = L.sbmc_upsample2x_cat_nhwc_bwd_adj_f32(_lib.ptr(g), _lib.ptr(gcoarse), _lib.ptr(gleft), _lib.ptr(adj.signs),
                                                                adj.slope, _lib.ptr(partial), _lib.ptr(amax), b, cu, cl, hc, w,
                                                                _lib.current_stream(dev))
                    tag_amax(gcoarse, amax)
                    adj.done = (partial, amax, gcoarse.data_ptr(), gcoarse._version)
                else:
                    rc = bwd(_lib.ptr(g), _lib.ptr(gcoarse), _lib.ptr(gleft), b, cu, cl, hc, w, top, bot,
                             _lib.current_stream(dev))
            _lib.check(rc, "upsample2x_cat_nhwc_bwd")
        return gcoarse, gleft, None, None, None


def gather_update_supported(data, kernels):
    """True when the fused gather-kernel update (`SplatUpdate(..., gather=True)`) applies."""
    if not (data.is_cuda and kernels.is_cuda) or data.dtype != th.float32 or kernels.dtype != th.float32:
        return False
    k2 = kernels.shape[1]
    k = int(round(k2 ** 0.5))
    if k * k != k2:
        return False
    return bool(_lib.lib().sbmc_gather_update_supported(int(data.shape[1]), k, int(kernels.shape[-2]),
                                                        int(kernels.shape[-1])))


def splat_all_supported_dims(c, k, h, w):
    """`SplatAll` applies to fp32 radiance with c channels, kernel size k, on an h x w frame."""
    return bool(_lib.lib().sbmc_splat_all_supported(int(c), int(k), int(h), int(w)))


def splat_all_supported(data, kernels):
    """True when `SplatAll` can take data [bs, S, c, h, w] / kernels [bs, S, k*k, h, w]."""
    if not (data.is_cuda and kernels.is_cuda) or data.dim() != 5 or kernels.dim() != 5:
        return False
    if data.dtype != th.float32 or kernels.dtype not in (th.float32, th.float16):
        return False
    k2 = kernels.shape[2]
    k = int(round(k2 ** 0.5))
    if k * k != k2:
        return False
    h, w = kernels.shape[-2:]
    if kernels.dtype == th.float16:
        return _half_ok(data.shape[2], k, h, w)
    return bool(_lib.lib().sbmc_splat_all_supported(int(data.shape[2]), k, int(h), int(w)))


def splat_slab_supported(data, kernels, top, bot):
    """True when `SplatAll` takes data [bs, S, c, h, w] / kernels [bs, S, k*k, h, w] in its row-slab
    form with `top` / `bot` overhang rows (the strip kernels)."""
    if not splat_all_supported(data, kernels):
        return False
    k = int(round(kernels.shape[2] ** 0.5))
    h, w = kernels.shape[-2:]
    if kernels.dtype == th.float16 and not _half_ok(data.shape[2], k, top + h + bot, w):
        return False
    return bool(_lib.lib().sbmc_splat_slab_supported(int(data.shape[2]), k, int(h), int(w), int(top), int(bot)))


class SplatAll(th.autograd.Function):
    """All S progressive splat updates of a frame in three launches per direction.

    Equivalent (up to fp32 rounding) to
        state = None, None, None
        for s in range(S): state = ProgressiveKernelApply(splat=True)(data[:, s], kernels[:, s], *state)
    i.e. to the sample loop of the reference's Multisteps.forward (sbmc/models.py:195-209): the
    running state is a log-sum-exp monoid, so every sample is reduced independently by ONE
    launch of the fused forward kernel over bs*S images and a per-pixel kernel folds the
    partial states in sample order; backward likewise (include/sbmc_hip.h).

    Args:
      data(th.Tensor)[bs, S, c, h, w]: sample radiance.
      kernels(th.Tensor)[bs, S, k*k, h, w]: per-sample splat kernel logits.
      top, bot(int), zero_top, zero_bot(bool): row-slab form (one frame sharded along H,
        sbmc_amd/dist.py; include/sbmc_hip.h "Row-slab form"): the tensors hold one slab's own h
        rows, the state comes out on top + h + bot destination rows; an edge with zero_* = False is
        shared with a neighbouring slab, whose sources contribute nothing here.  Defaults = the
        whole frame.
    Returns:
      sum_r[bs, c, hd, w], sum_w[bs, 1, hd, w], max_w[bs, 1, hd, w], hd = top + h + bot.
    """

    @staticmethod
    def forward(ctx, data, kernels, top=0, bot=0, zero_top=True, zero_bot=True):
        bs, S, k2, h, w = kernels.shape
        k = int(round(k2 ** 0.5))
        c = data.shape[2]
        if tuple(data.shape) != (bs, S, c, h, w):
            raise RuntimeError("data should be [bs, S, c, h, w] matching kernels [bs, S, k*k, h, w]")
        _require_f32("SplatAll", data=data, kernels=None if kernels.dtype == th.float16 else kernels)
        slab = bool(top or bot or not zero_top or not zero_bot)
        hd = top + h + bot
        if kernels.dtype == th.float16 and (not kernels.is_cuda or not _half_ok(c, k, hd, w)):
            raise TypeError("SplatAll: half logits are only taken by the k=21 strip kernels")
        data = data.contiguous()
        kernels = kernels.contiguous()
        dev = data.device
        part_r = data.new_empty(bs, S, c, hd, w)
        part_w = data.new_empty(bs, S, hd, w)
        part_m = data.new_empty(bs, S, hd, w)
        atap = th.empty(bs, S, hd, w, dtype=th.int32, device=dev)
        lib = _lib.lib()
        with th.cuda.device(dev):
            half = kernels.dtype == th.float16
            with _timed("splat_update_fwd_all_f16" if half else "splat_update_fwd_all", dev):
                if slab:
                    rc = (lib.sbmc_splat_slab_fwd_f16 if half else lib.sbmc_splat_slab_fwd_f32)(
                        _lib.ptr(data), _lib.ptr(kernels),
                        _lib.ptr(part_r), _lib.ptr(part_w), _lib.ptr(part_m), _lib.ptr(atap),
                        bs * S, c, h, w, k, int(top), int(bot), int(bool(zero_top)), int(bool(zero_bot)),
                        _lib.current_stream(dev))
                else:
                    kmax = data.new_empty(bs, S, h, w)
                    rc = (lib.sbmc_splat_update_fwd_f16 if half else lib.sbmc_splat_update_fwd_f32)(
                        _lib.ptr(data), _lib.ptr(kernels), None, None, None,
                        _lib.ptr(part_r), _lib.ptr(part_w), _lib.ptr(part_m), _lib.ptr(kmax), _lib.ptr(atap),
                        bs * S, c, h, w, k, _lib.current_stream(dev))
            _lib.check(rc, "splat_update_fwd (all samples)")
            sum_r = data.new_empty(bs, c, hd, w)
            sum_w = data.new_empty(bs, 1, hd, w)
            max_w = data.new_empty(bs, 1, hd, w)
            run_r = th.empty_like(part_r)
            run_w = data.new_empty(bs, S, hd, w)
            run_m = data.new_empty(bs, S, hd, w)
            rc = lib.sbmc_splat_merge_fwd_f32(
                _lib.ptr(part_r), _lib.ptr(part_w), _lib.ptr(part_m),
                _lib.ptr(sum_r), _lib.ptr(sum_w), _lib.ptr(max_w),
                _lib.ptr(run_r), _lib.ptr(run_w), _lib.ptr(run_m),
                bs, S, c, hd, w, _lib.current_stream(dev))
            _lib.check(rc, "splat_merge_fwd")
        ctx.k, ctx.slab = k, (int(top), int(bot))
        ctx.save_for_backward(data, kernels, part_m, atap, run_r, run_w, run_m)
        return sum_r, sum_w, max_w

    @staticmethod
    def backward(ctx, d_r, d_w, d_m):
        data, kernels, part_m, atap, run_r, run_w, run_m = ctx.saved_tensors
        bs, S, c, h, w = data.shape
        top, bot = ctx.slab
        hd = top + h + bot
        dev = data.device
        d_r = data.new_zeros(bs, c, hd, w) if d_r is None else d_r.contiguous()
        d_w = data.new_zeros(bs, 1, hd, w) if d_w is None else d_w.contiguous()
        d_m = data.new_zeros(bs, 1, hd, w) if d_m is None else d_m.contiguous()
        d_data = th.empty_like(data)
        d_kernels = th.empty_like(kernels)
        nbytes = _lib.lib().sbmc_splat_update_bwd_scratch_bytes(bs * S, c, hd, w, ctx.k)
        scratch = data.new_empty((nbytes + 3) // 4)
        half = kernels.dtype == th.float16
        L = _lib.lib()
        args = (_lib.ptr(data), _lib.ptr(kernels), _lib.ptr(part_m), _lib.ptr(atap),
                _lib.ptr(run_r), _lib.ptr(run_w), _lib.ptr(run_m),
                _lib.ptr(d_r), _lib.ptr(d_w), _lib.ptr(d_m),
                _lib.ptr(d_data), _lib.ptr(d_kernels), _lib.ptr(scratch), bs, S, c, h, w, ctx.k)
        # fp32: the per-pixel chain also leaves an upper bound of |d_kernels| in a device word (the scale of the 441-channel
        # layer's backward, which reads d_kernels next: no pass over 13 GB for it); it needs max |data|
        bound = None
        if (not half and knob("SBMC_HIP_PW_F16") != 0 and knob("SBMC_AMAX_TAGS") != 0
                and (known_amax(data) is not None or data.data_ptr() % 16 == 0)):    # (the absmax pass takes 16-byte-aligned pointers)
            # (max |data|: its tag if a pass left one -- never tagged here: the radiance is a network INPUT, a new
            # tensor every step of a real run, and a bench that reuses its batch must not skip the pass)
            dmax, bound = known_amax(data), amax_word(dev)
            if dmax is None:
                dmax = Conv3x3NHWC._absmax(data)
        with th.cuda.device(dev), _timed("splat_update_bwd_all_f16" if half else "splat_update_bwd_all", dev):
            if bound is not None:
                rc = L.sbmc_splat_all_bwd_bound_f32(*args[:13], _lib.ptr(dmax), _lib.ptr(bound), *args[13:], top, bot,
                                                    _lib.current_stream(dev))
            elif top or bot:
                rc = (L.sbmc_splat_slab_bwd_f16 if half else L.sbmc_splat_slab_bwd_f32)(
                    *args, top, bot, _lib.current_stream(dev))
            else:
                rc = (L.sbmc_splat_all_bwd_f16 if half else L.sbmc_splat_all_bwd_f32)(
                    *args, _lib.current_stream(dev))
        _lib.check(rc, "splat_all_bwd")
        if bound is not None:
            tag_amax(d_kernels, bound)
        return d_data, d_kernels, None, None, None, None
