"""Building blocks of the SBMC denoiser on MI355X.

Public API, constructor signatures, sub-module names and therefore state-dict
keys are those of the reference's ``sbmc/modules.py`` (``__all__`` at :29), so a
reference checkpoint loads unchanged:

* ``ConvChain`` (:34-192) and ``Autoencoder`` (:195-320): the kernel-predicting
  backbone.  Plain ``torch.nn`` convolutions -- on ROCm these run on MIOpen
  (MFMA); nothing here is hand-written.
* ``KernelApply`` (:323-361) and ``ProgressiveKernelApply`` (:364-473): the splat
  path.  On GPU tensors ``ProgressiveKernelApply(splat=True)`` is ONE fused HIP
  operator (``functions.SplatUpdate``); every other case composes the
  boundary-level operators ``Scatter2Gather`` / ``KernelWeighting`` exactly like
  the reference does.
"""
import logging
import warnings

import numpy as np
import torch as th
import torch.nn as nn
import torch.nn.functional as F

from . import functions as funcs
from .utils import knob

__all__ = ["ConvChain", "Autoencoder", "KernelApply", "ProgressiveKernelApply"]

LOG = logging.getLogger(__name__)

_ACTIVATIONS = {
    "relu": nn.ReLU,
    "leaky_relu": nn.LeakyReLU,
    "tanh": nn.Tanh,
    "elu": nn.ELU,
}


def _weight_norm(conv):
    # Old-style weight norm on purpose: it registers `weight_g` / `weight_v`, the
    # parameter names found in reference checkpoints.
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", FutureWarning)
        return nn.utils.weight_norm(conv)


def _init_conv(conv, nonlinearity):
    # Reference behaviour (modules.py:85-94,178-188): zero bias, Xavier-uniform on
    # `conv.weight`.  With weight norm `conv.weight` is a derived tensor, so the
    # Xavier call does not touch the trainable weight_g / weight_v -- kept as is so
    # that a seeded init reproduces the reference's parameters bit for bit.
    conv.bias.data.zero_()
    gain_of = "relu" if nonlinearity in ("elu", "softplus") else nonlinearity
    nn.init.xavier_uniform_(conv.weight.data, nn.init.calculate_gain(gain_of))


def conv_weight(conv):
    """The convolution's weight: out of the step's weight bank when one is installed (sbmc_amd/wbank.py: the weight
    norm of many layers per launch, 3 x 3 weights prepared for csrc/conv3x3.hip on the way), else torch's weight norm
    (reference sbmc/modules.py:85-94: w = g v / ||v||), else the plain parameter."""
    w = conv.__dict__.get("_sbmc_bank_w")
    if w is not None:
        return w
    if hasattr(conv, "weight_g"):   # old-style weight norm: w = g * v / ||v||, norm over dims 1..3
        return th._weight_norm(conv.weight_v, conv.weight_g, 0)
    return conv.weight


def _is_pointwise(conv):
    return (isinstance(conv, nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is not None)


def _pointwise_gemm(conv, x, activation=None, mean_s=0, mean_out=None):
    """conv(x) [+ activation] for a 1x1 convolution as y[b] = W @ x[b] on [B, Cin, H*W] (no
    layout change), with bias and ReLU / LeakyReLU fused (functions.PointwiseLayer).
    Returns (y, activation_was_applied).  mean_s > 0 (only honoured when nothing is left to apply
    after the fused layer): the mean of y over groups of mean_s batch elements is appended to
    `mean_out` (functions.PointwiseLayerMean)."""
    w = conv_weight(conv)
    b, c, h, wd = x.shape
    wmat = w.view(1, w.shape[0], c).expand(b, -1, -1)
    x3 = funcs.tagged_view(x, b, c, h * wd)         # (magnitude words travel with the activations: functions.tag_amax)
    act, slope = 0, 0.0
    if isinstance(activation, nn.ReLU):
        act = 1
    elif isinstance(activation, nn.LeakyReLU):
        act, slope = 2, float(activation.negative_slope)
    half = funcs.pointwise_half_supported(x3, w.shape[0])   # fp16 activations (torch.autocast(float16))
    if half or funcs.pointwise_supported(x3, w.shape[0]):
        wm, bias = w.view(w.shape[0], c).float(), conv.bias.float()
        if mean_s and mean_out is not None and b % mean_s == 0 and (activation is None or act != 0):
            y, m = funcs.PointwiseLayerMean.apply(x3, wm, bias, None, 1, act, slope, mean_s, half)
            mean_out.append(funcs.tagged_view(m, b // mean_s, w.shape[0], h, wd))
        else:
            y = funcs.PointwiseLayer.apply(x3, wm, bias, None, 1, act, slope, half)
        return funcs.tagged_view(y, b, w.shape[0], h, wd), act != 0
    y = th.bmm(wmat, x3)
    if funcs.BiasAct.supported(y):
        y = funcs.BiasAct.apply(y, conv.bias, act, slope)
        return y.view(b, w.shape[0], h, wd), act != 0
    y = y + conv.bias.view(1, -1, 1)
    return y.view(b, w.shape[0], h, wd), False


def pointwise_chain_with_context(chain, per_sample, context, mean_out=None):
    """`chain(cat([per_sample[:, s], context], 1))` for every sample s of a 1x1 ConvChain, without
    building the concatenation: the first layer is linear, so its context half W_c @ context is
    computed once per pixel and added (with bias and activation) to the per-sample half
    W_f @ per_sample[:, s] by one fused pass (functions.CtxAct).  Same arithmetic as the
    reference's th.cat([f, propagated], 1) -> conv (sbmc/models.py:147-153,171-177,196-199) up to
    fp32 summation order; the context product is done once instead of once per sample.

    per_sample [bs, S, cs, h, w], context [bs, cp, h, w] or [bs, cp, 1, 1] -> [bs*S, cout, h, w];
    returns None when the fused path does not apply (the caller then concatenates).
    mean_out: a list; when the chain's last layer can deliver it, the mean of the output over the S
    samples of every pixel ([bs, cout, h, w]) is appended (see functions.PointwiseLayerMean).
    """
    mods = list(chain.children())
    first = mods[0]
    if isinstance(first, ConvChain._ConvBNRelu):
        if len(first.layer) != 2:
            return None
        conv, act_mod = first.layer[0], first.layer[1]
    elif isinstance(first, nn.Conv2d):
        conv, act_mod = first, (mods[1] if len(mods) > 1 else None)
    else:
        return None
    if not (_is_pointwise(conv) and per_sample.is_cuda and per_sample.dtype in (th.float32, th.float16)):
        return None
    if per_sample.dtype == th.float16 and not funcs.pointwise_half_supported(per_sample.reshape(
            per_sample.shape[0] * per_sample.shape[1], per_sample.shape[2], -1), conv.out_channels):
        return None
    act = 0, 0.0
    if isinstance(act_mod, nn.ReLU):
        act = 1, 0.0
    elif isinstance(act_mod, nn.LeakyReLU):
        act = 2, float(act_mod.negative_slope)
    elif act_mod is not None and isinstance(first, ConvChain._ConvBNRelu):
        return None                      # tanh / elu chains keep the generic path
    bs, S, cs, h, w = per_sample.shape
    cp = context.shape[1]
    if conv.in_channels != cs + cp:
        return None
    wt = conv_weight(conv)
    wt = wt.view(wt.shape[0], cs + cp)
    cout = wt.shape[0]
    xs = funcs.tagged_view(per_sample, bs * S, cs, h * w)
    nhwc_ctx = funcs._is_channels_last(context) and context.dtype in (th.float32, th.float16) and h * w > 1
    ctx3 = None if nhwc_ctx else context.reshape(bs, cp, -1)

    def context_term(wc):
        # the U-net may hand its result over channels-last: the product then reads it in place and returns
        # the context gradient channels-last as well (functions.ContextProductNHWC).  A half context (fp16
        # activations) is widened in its own memory order first: an elementwise pass instead of a strided copy
        if nhwc_ctx:
            return funcs.ContextProductNHWC.apply(context.float(), wc.float().contiguous())
        return th.bmm(wc.unsqueeze(0).expand(bs, -1, -1), ctx3.to(wc.dtype))
    if funcs.pointwise_half_supported(xs, cout):             # fp16 activations, inference
        with th.autocast("cuda", enabled=False):
            t = context_term(wt[:, cs:].float()).contiguous()
        tt = t if t.shape[2] == h * w and h * w > 1 else t.reshape(bs, cout)
        y = funcs.PointwiseLayer.apply(xs, wt[:, :cs].float(), conv.bias.float(), tt, S, act[0], act[1], True)
        y = y.view(bs * S, cout, h, w)
        consumed = 1 if isinstance(first, ConvChain._ConvBNRelu) else (2 if act[0] != 0 else 1)
        rest = mods[consumed:]
        return chain._run(rest, y, mean_s=S, mean_out=mean_out) if rest else y
    t = context_term(wt[:, cs:]).contiguous()
    if t.dtype == th.float32 and funcs.pointwise_supported(xs, cout):
        tt = t if t.shape[2] == h * w and h * w > 1 else t.reshape(bs, cout)
        # the context term's magnitude word without a pass over it: the fused chain wants it only for the BOUND it scales
        # its intermediate activations by, and max |context| x the largest absolute row sum of the context weights is one
        # (a library GEMM made t: nobody left a word on it, and each chain ran an absmax pass over it -- 3 per step)
        ca = funcs.known_amax(context) if tt is t else None
        if ca is not None and funcs.known_amax(tt) is None:
            with th.no_grad():
                bound = ca.view(th.float32) * wt[:, cs:].detach().abs().sum(1).max()
            funcs.tag_amax(tt, bound.view(th.int32))
        # two or three layers of the chain in ONE pass where they fit (functions.PointwiseChain: a tile's intermediate
        # activations stay on the chip); what is behind them -- the regressor's 441-channel layer -- runs as before
        plan, used = _pointwise_plan(mods)
        if len(plan) >= 2 and funcs.pointwise_chain_supported(xs, [c.out_channels for c, _, _ in plan]):
            wb = [wt[:, :cs], conv.bias]
            for c, _, _ in plan[1:]:
                wc = conv_weight(c)
                wb += [wc.view(wc.shape[0], wc.shape[1]), c.bias]
            rest = mods[used:]
            mean = bool(mean_out is not None and not rest)
            out = funcs.PointwiseChain.apply(xs, tt, S, mean, tuple((a, sl) for _, a, sl in plan), *wb)
            y, m = out if mean else (out, None)
            cl = plan[-1][0].out_channels
            if mean:
                mean_out.append(funcs.tagged_view(m, bs, cl, h, w))
            y = funcs.tagged_view(y, bs * S, cl, h, w)
            return chain._run(rest, y, mean_s=S, mean_out=mean_out) if rest else y
        y = funcs.PointwiseLayer.apply(xs, wt[:, :cs], conv.bias, tt, S, act[0], act[1])
        y = funcs.tagged_view(y, bs * S, cout, h, w)
    else:
        y = th.bmm(wt[:, :cs].unsqueeze(0).expand(bs * S, -1, -1), xs)
        if not funcs.CtxAct.supported(y, t, S):
            return None
        y = funcs.CtxAct.apply(y, t, conv.bias, S, act[0], act[1]).view(bs * S, cout, h, w)
    # the rest of the chain, minus what has been consumed
    consumed = 1 if isinstance(first, ConvChain._ConvBNRelu) else (2 if act[0] != 0 else 1)
    rest = mods[consumed:]
    return chain._run(rest, y, mean_s=S, mean_out=mean_out) if rest else y


def _pointwise_plan(mods, limit=3):
    """The leading layers of a ConvChain's module list that `functions.PointwiseChain` can take in one pass: 1x1
    convolutions of at most 128 output channels, each followed by ReLU / LeakyReLU or -- the last one of the list --
    by nothing.  -> ([(conv, act, slope), ...], number of modules of `mods` they cover)."""
    plan, i = [], 0
    while i < len(mods) and len(plan) < limit:
        m = mods[i]
        if isinstance(m, ConvChain._ConvBNRelu):
            if len(m.layer) != 2:
                break
            conv, act_mod, step = m.layer[0], m.layer[1], 1
        elif isinstance(m, nn.Conv2d):
            conv, act_mod, step = m, (mods[i + 1] if i + 1 < len(mods) else None), 2
        else:
            break
        if not _is_pointwise(conv) or conv.out_channels > 128 or (plan and conv.in_channels != plan[-1][0].out_channels):
            break
        if isinstance(act_mod, nn.ReLU):
            act = (1, 0.0)
        elif isinstance(act_mod, nn.LeakyReLU):
            act = (2, float(act_mod.negative_slope))
        elif act_mod is None and isinstance(m, nn.Conv2d):
            act, step = (0, 0.0), 1
        else:
            break
        plan.append((conv, act[0], act[1]))
        i += step
    return plan, i


_LAYOUT_DECISIONS = {}


def _convs_on_own_kernel(net, x):
    """True if all spatial convolutions of `net` are 3 x 3 / stride 1 / padding 1 with channel counts the
    split-precision kernel takes (functions.Conv3x3NHWC), at x's resolution or coarser."""
    import os
    if knob("SBMC_CONV3X3") == 0 or x.shape[2] * x.shape[3] < 2:
        return False
    convs = [m for m in net.modules() if isinstance(m, nn.Conv2d) and m.kernel_size != (1, 1)]
    if not convs:
        return False
    L = funcs._lib.lib()
    for m in convs:
        if not (m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1) and m.dilation == (1, 1)
                and m.groups == 1 and m.padding_mode == "zeros"
                and L.sbmc_conv3x3_supported(int(x.shape[0]), int(x.shape[2]), int(x.shape[3]), m.in_channels, m.out_channels)
                and L.sbmc_conv3x3_supported(int(x.shape[0]), int(x.shape[2]), int(x.shape[3]), m.out_channels, m.in_channels)):
            return False
    return True


def unet_channels_last(net, x, rows=None):
    """Should this U-net run channels-last on this input?  MEASURED, once per (channels, height, width,
    training?) and process: one 3x3 convolution of the net's width at the input's resolution, forward (and
    backward when gradients are on), in both layouts.  With the find records of `sbmc_amd.miopen_db` MIOpen
    picks its NHWC implicit-GEMM solvers and channels-last wins by the NCHW<->NHWC transposes it no longer
    needs; without a matching record (another GPU, MIOpen build or frame size) its heuristic choice for
    channels-last fp32 can be many times slower, and the U-net stays planar.
    rows: measure at this height instead of x's (a row slab is convolved together with its halo rows).
    SBMC_UNET_LAYOUT = nchw | nhwc overrides the measurement."""
    import os
    mode = os.environ.get("SBMC_UNET_LAYOUT", "auto").lower()
    half = th.is_autocast_enabled() and th.get_autocast_dtype("cuda") == th.float16
    if (mode == "nchw" or not x.is_cuda or x.dim() != 4 or x.shape[1] % 4 or x.numel() == 0
            or x.dtype not in (th.float32, th.float16) or (th.is_autocast_enabled() and not half)):
        return False
    if mode == "nhwc":
        return True
    if x.dtype == th.float32 and not half and _convs_on_own_kernel(net, x):
        # every 3 x 3 convolution of the net runs on csrc/conv3x3.hip, which is channels-last by construction (and
        # ~2-3x MIOpen's fp32 solvers in either layout): nothing to measure, no dependence on MIOpen's find-db
        _LAYOUT_DECISIONS.setdefault((x.device.index, "own 3x3 kernel"), True)
        return True
    if x.dtype == th.float16 and _convs_on_own_kernel(net, x) and knob("SBMC_CONV3X3_HALF") != 0:
        # half activations: the same kernels in their one-plane form (functions.Conv3x3BiasActHalfNHWC)
        _LAYOUT_DECISIONS.setdefault((x.device.index, "own 3x3 kernel"), True)
        return True
    grad = th.is_grad_enabled() and any(q.requires_grad for q in net.parameters())
    shape = (x.shape[0], x.shape[1], int(rows) if rows else x.shape[2], x.shape[3])
    dtype = th.float16 if (half or x.dtype == th.float16) else th.float32     # what MIOpen will convolve in
    first = next((m for m in net.modules() if isinstance(m, nn.Conv2d) and m.kernel_size[0] > 1), None)
    cout = first.out_channels if first is not None and first.in_channels == x.shape[1] else x.shape[1]
    key = (x.device.index,) + shape + (cout, grad, dtype)
    if key not in _LAYOUT_DECISIONS:
        _LAYOUT_DECISIONS[key] = _measure_layouts(shape, x.device, grad, dtype, cout)
    return _LAYOUT_DECISIONS[key]


def _measure_layouts(shape, device, grad, dtype=th.float32, cout=None):
    """One 3x3 convolution of the net's first layer's shape (c -> cout channels), forward and -- with `grad`
    -- backward, in both layouts: the median of five runs each; channels-last must win by 5 % (the decision
    must not flip on timing noise: the two layouts differ in fp32 rounding).  Logged at INFO."""
    b, c, h, w = shape
    cout = cout or c
    times = {}
    with th.enable_grad(), th.autocast("cuda", enabled=False):
        for cl in (False, True):
            xin = th.zeros(b, c, h, w, device=device, dtype=dtype)
            wt = th.zeros(cout, c, 3, 3, device=device, dtype=dtype)
            if cl:
                xin = xin.contiguous(memory_format=th.channels_last)
                wt = wt.contiguous(memory_format=th.channels_last)
            xin.requires_grad_(grad)
            wt.requires_grad_(grad)

            def run():
                y = F.conv2d(xin, wt, None, 1, 1)
                if grad:
                    y.backward(y.detach())
                    xin.grad = wt.grad = None
            run()                                   # solver selection / kernel load
            th.cuda.synchronize(device)
            marks = [th.cuda.Event(enable_timing=True) for _ in range(6)]
            for i in range(5):
                marks[i].record()
                run()
            marks[5].record()
            th.cuda.synchronize(device)
            times[cl] = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(5))[2]
            del xin, wt
    pick = times[True] < 0.95 * times[False]
    LOG.info("U-net layout at %s (%d -> %d channels, %s%s): planar %.2f ms, channels-last %.2f ms -> %s",
             tuple(shape), c, cout, str(dtype).replace("torch.", ""), ", with backward" if grad else "",
             times[False], times[True], "channels-last" if pick else "planar")
    return pick


class ConvChain(nn.Module):
    """A stack of ``depth`` convolutions: (depth-1) x [conv, (norm), activation] + conv.

    Args (same as the reference):
        ninputs(int), noutputs(int): input / output channels.
        ksize(int): size of all the convolution kernels.
        width(int): channels of the intermediate layers.
        depth(int): number of conv layers (> 0).
        stride(int): stride of the intermediate convolutions.
        pad(bool): zero-pad to keep the resolution, else 'valid' convolutions.
        normalize(bool), normalization_type(str): optional batch / instance norm.
        output_type(str): linear, relu, leaky_relu, sigmoid, tanh, elu, softplus.
        activation(str): relu, leaky_relu, tanh, elu.
        weight_norm(bool): weight-normalised convolutions.
    """

    def __init__(self, ninputs, noutputs, ksize=3, width=64, depth=3, stride=1,
                 pad=True, normalize=False, normalization_type="batch",
                 output_type="linear", activation="relu", weight_norm=True):
        super(ConvChain, self).__init__()
        if depth <= 0:
            LOG.error("ConvChain should have non-negative depth.")
            raise ValueError("negative network depth.")
        padding = ksize // 2 if pad else 0

        nin = ninputs
        for d in range(depth - 1):
            self.add_module("layer_{}".format(d), ConvChain._ConvBNRelu(
                nin, ksize, width, normalize=normalize,
                normalization_type=normalization_type, padding=padding,
                stride=stride, activation=activation, weight_norm=weight_norm))
            nin = width

        head = nn.Conv2d(nin, noutputs, ksize, bias=True, padding=padding)
        if weight_norm:
            head = _weight_norm(head)
        if output_type not in ("linear", "relu", "leaky_relu", "sigmoid", "tanh",
                               "elu", "softplus"):
            raise ValueError("Unknon output type '{}'".format(output_type))
        _init_conv(head, output_type)
        self.add_module("prediction", head)

        out_act = {
            "relu": lambda: nn.ReLU(inplace=True),
            "leaky_relu": lambda: nn.LeakyReLU(inplace=True),
            "sigmoid": nn.Sigmoid,
            "tanh": nn.Tanh,
            "elu": nn.ELU,
            "softplus": nn.Softplus,
        }.get(output_type)
        if out_act is not None:
            self.add_module("output_activation", out_act())

    #: 1x1 / stride-1 convolutions as plain batched GEMMs (rocBLAS / hipBLASLt) on the planar
    #: NCHW activations: y[b] = W @ x[b].  Same arithmetic as the convolution; set per instance
    #: by Multisteps for its per-sample chains.
    pointwise_as_gemm = False
    #: spatial convolutions without bias + ONE fused in-place bias / activation pass per direction
    #: (functions.BiasAct) instead of torch's separate bias add, activation, activation backward and
    #: bias-gradient reduction; set per instance by Multisteps for its U-nets.
    fuse_bias_act = False

    def forward(self, x, want_link=False):
        """want_link: the caller is the ONLY reader of the result and takes the chain's last layer's `_AdjLink`
        (functions.Conv3x3BiasActNHWC.adj_link_for) -- a U-net level whose pooling + skip node applies that layer's
        activation adjoint in its own backward pass."""
        return self._run(list(self.children()), x, want_link=want_link)

    @staticmethod
    def _conv_bias_act(conv, x, activation, adj_in=None, want_link=False):
        """conv(x) without bias, then bias + activation by the fused pass.  Returns (y, activation
        was applied), or (None, False) when the fused pass does not apply.
        adj_in / want_link: see functions.Conv3x3BiasActNHWC.forward -- `_run` passes them between the layers of ONE
        chain, whose intermediate maps nothing else reads."""
        if conv.bias is None or conv.padding_mode != "zeros" or not isinstance(conv.padding, tuple):
            return None, False
        act, slope = 0, 0.0
        if isinstance(activation, nn.ReLU):
            act = 1
        elif isinstance(activation, nn.LeakyReLU):
            act, slope = 2, float(activation.negative_slope)
        if x.dtype == th.float16:
            # fp16 activations (torch.autocast(float16)): the one-plane form of csrc/conv3x3.hip, bias + activation in
            # its epilogue; anything it does not take runs the module (MIOpen's half solvers) as before
            if (activation is None or act != 0) and funcs.Conv3x3BiasActHalfNHWC.supported(x, conv):
                return funcs.Conv3x3BiasActHalfNHWC.apply(x, conv_weight(conv), conv.bias, act, slope), act != 0
            return None, False
        if th.is_autocast_enabled():
            return None, False
        w = conv_weight(conv)
        if funcs._is_channels_last(x):
            # the U-net runs channels-last (Autoencoder.forward): MIOpen's NHWC solvers without any layout
            # change around them, bias + activation by the NHWC pass
            if funcs.Conv3x3NHWC.supported(x, conv) and funcs.Conv3x3BiasActNHWC.supported(conv):
                # csrc/conv3x3.hip: fp32 values on the f16 matrix pipe, bias + activation in the kernel's epilogue
                y, amax = funcs.Conv3x3BiasActNHWC.apply(x, w, conv.bias, act, slope, adj_in, want_link)
                return funcs.tag_amax(y, amax), act != 0
            elif funcs.Conv3x3NHWC.supported(x, conv):
                y = funcs.Conv3x3NHWC.apply(x, w)             # (a channel count the bias / activation adjoint does not take)
            else:
                w = w.contiguous(memory_format=th.channels_last)
                y = th.nn.functional.conv2d(x, w, None, conv.stride, conv.padding, conv.dilation, conv.groups)
            if funcs.BiasActNHWC.supported(y):
                # (the pass also finds max |y|: the next 3 x 3 convolution scales by it)
                y, amax = funcs.BiasActNHWC.apply(y, conv.bias, act, slope, True)
                return funcs.tag_amax(y, amax), act != 0
            return y + conv.bias.view(1, -1, 1, 1), False
        y = th.nn.functional.conv2d(x, w, None, conv.stride, conv.padding, conv.dilation, conv.groups)
        if not funcs.BiasAct.supported(y):
            # (odd plane sizes, channels_last results, empty batches) the convolution is done: finish
            # with torch's bias add instead of running the module -- and the convolution -- again
            return y + conv.bias.view(1, -1, 1, 1), False
        return funcs.BiasAct.apply(y, conv.bias, act, slope), act != 0

    def _run(self, mods, x, mean_s=0, mean_out=None, halo=None, want_link=False):
        """halo: the chain runs on a row slab of a frame sharded over several GPUs (sbmc_amd.dist) -- a tuple
        (pad, crop[, refresh]): before every padded k x k convolution `pad(x, k // 2)` attaches the neighbouring
        slabs' k // 2 edge rows, and `crop(y, k // 2)` drops the rows of its result that saw the artificial zero
        padding beyond them (lazily: just before the next exchange, or at the end, so that in-place activations
        never meet a view).  `refresh(y, k // 2)`, where given and applicable (it returns None otherwise), does
        crop + pad without copying the slab: the stale outer rows of `y` are overwritten by the neighbours'."""
        gemm = self.pointwise_as_gemm and x.is_cuda
        pending = 0                                                # halo rows still to be dropped
        i = 0
        # (two consecutive fused 3 x 3 layers of the chain on a whole frame: the second one's data gradient applies the
        # first one's activation adjoint -- functions._AdjLink; a sharded frame's halo rows travel between them)
        link = None
        fusable = lambda q: halo is None and (
            (isinstance(q, ConvChain._ConvBNRelu) and len(q.layer) == 2 and isinstance(q.layer[0], nn.Conv2d)
             and q.layer[0].kernel_size == (3, 3)) or (isinstance(q, nn.Conv2d) and q.kernel_size == (3, 3)))
        while i < len(mods):
            m = mods[i]
            i += 1
            link_in, link = link, None
            if halo is not None:
                conv = m.layer[0] if isinstance(m, ConvChain._ConvBNRelu) else m
                if isinstance(conv, nn.Conv2d) and conv.kernel_size[0] > 1:
                    need = conv.kernel_size[0] // 2
                    fresh = halo[2](x, need) if (len(halo) > 2 and pending == need) else None
                    if fresh is not None:
                        x = fresh                                 # the stale halo rows renewed in place
                    else:
                        if pending:
                            x = halo[1](x, pending)
                        x = halo[0](x, need)
                    pending = need
            if gemm and isinstance(m, ConvChain._ConvBNRelu) and _is_pointwise(m.layer[0]):
                rest = list(m.layer.children())[1:]
                x, fused = _pointwise_gemm(m.layer[0], x, rest[0] if len(rest) == 1 else None)
                for sub in (rest[1:] if fused else rest):
                    x = sub(x)
            elif gemm and isinstance(m, nn.Conv2d) and _is_pointwise(m):
                nxt = mods[i] if i < len(mods) else None          # the chain's output activation
                last = i + (1 if nxt is not None else 0) >= len(mods)
                x, fused = _pointwise_gemm(m, x, nxt, mean_s if last else 0, mean_out)
                if fused:
                    i += 1
                elif nxt is not None and mean_out:
                    mean_out.clear()                              # an activation is still to come
            elif (self.fuse_bias_act and x.is_cuda and x.dtype in (th.float32, th.float16)
                  and isinstance(m, ConvChain._ConvBNRelu) and len(m.layer) == 2
                  and isinstance(m.layer[0], nn.Conv2d)):
                y, fused = ConvChain._conv_bias_act(m.layer[0], x, m.layer[1], link_in,
                                                    fusable(m) and i < len(mods) and fusable(mods[i]))
                x = m(x) if y is None else (y if fused else m.layer[1](y))
                link = funcs.Conv3x3BiasActNHWC.adj_link_for(x) if (y is not None and fused) else None
            elif self.fuse_bias_act and x.is_cuda and x.dtype in (th.float32, th.float16) and isinstance(m, nn.Conv2d):
                nxt = mods[i] if i < len(mods) else None
                # (the chain's last layer: what is behind it is the caller's business)
                y, fused = ConvChain._conv_bias_act(m, x, nxt, link_in, want_link and halo is None
                                                    and i + (1 if nxt is not None else 0) >= len(mods))
                if y is None:
                    x = m(x)
                else:
                    x = y
                    if fused:
                        i += 1
            else:
                x = m(x)
        if halo is not None and pending:
            x = halo[1](x, pending)
        return x

    class _ConvBNRelu(nn.Module):
        """conv -> (norm) -> activation, stored as ``self.layer`` (an nn.Sequential)."""

        def __init__(self, ninputs, ksize, noutputs, normalize=False,
                     normalization_type="batch", stride=1, padding=0,
                     activation="relu", weight_norm=True):
            super(ConvChain._ConvBNRelu, self).__init__()
            if activation not in _ACTIVATIONS:
                LOG.error("Incorrect activation %s", activation)
                raise ValueError("activation should be one of: relu, leaky_relu, tanh, elu")
            act = _ACTIVATIONS[activation]
            if normalize:
                conv = nn.Conv2d(ninputs, noutputs, ksize, stride=stride,
                                 padding=padding, bias=False)
                if normalization_type == "batch":
                    nrm = nn.BatchNorm2d(noutputs)
                elif normalization_type == "instance":
                    nrm = nn.InstanceNorm2d(noutputs, affine=True)
                else:
                    LOG.error("Incorrect normalization %s", normalization_type)
                    raise ValueError("Unkown normalization type {}".format(normalization_type))
                nrm.bias.data.zero_()
                nrm.weight.data.fill_(1.0)
                self.layer = nn.Sequential(conv, nrm, act())
                gain_of = "relu" if activation == "elu" else activation
                nn.init.xavier_uniform_(conv.weight.data, nn.init.calculate_gain(gain_of))
            else:
                conv = nn.Conv2d(ninputs, noutputs, ksize, stride=stride, padding=padding)
                if weight_norm:
                    conv = _weight_norm(conv)
                self.layer = nn.Sequential(conv, act())
                _init_conv(conv, activation)

        def forward(self, x):
            return self.layer(x)


class Autoencoder(nn.Module):
    """U-net: per level [left ConvChain] -> pool -> coarser level -> bilinear up ->
    concat(skip) -> [right ConvChain].  Arguments as in the reference (:195-245)."""

    def __init__(self, ninputs, noutputs, ksize=3, width=64, num_levels=3,
                 num_convs=2, max_width=512, increase_factor=1.0,
                 normalize=False, normalization_type="batch",
                 output_type="linear", activation="relu", pooling="max"):
        super(Autoencoder, self).__init__()

        def level_width(lvl):
            return min(int(width * increase_factor ** lvl), max_width)

        coarser = None
        for lvl in reversed(range(num_levels)):
            n_in, n_out, o_type = level_width(lvl - 1), level_width(lvl), activation
            n_us = level_width(lvl + 1)
            if lvl == 0:
                n_in, n_out, o_type = ninputs, noutputs, output_type
            elif lvl == num_levels - 1:
                n_us = None
            coarser = Autoencoder._Level(
                n_in, n_out, next_level=coarser, num_us=n_us, ksize=ksize,
                width=level_width(lvl), num_convs=num_convs, output_type=o_type,
                normalize=normalize, normalization_type=normalization_type,
                activation=activation, pooling=pooling)
        self.add_module("net", coarser)

    #: hand the result back in channels-last memory order when the U-net ran that way (Multisteps sets it: its
    #: consumer, the context product of the next 1x1 chain, reads either layout without a copy)
    keep_channels_last = False

    def forward(self, x):
        if unet_channels_last(self, x):
            # channels-last between the convolutions: MIOpen's NHWC-native solvers need no transposes then
            if funcs.ToChannelsLast.supported(x) and x.dtype == th.float32:
                xin, amax = funcs.ToChannelsLast.apply(x, True)       # (the first convolution's scale, found on the way)
                funcs.tag_amax(xin, amax)
            elif funcs.ToChannelsLast.supported(x):
                xin = funcs.ToChannelsLast.apply(x)                   # (half activations: no scales)
            else:
                xin = x.contiguous(memory_format=th.channels_last)
            y = self.net(xin)
            if self.keep_channels_last:
                return y
            return funcs.FromChannelsLast.apply(y) if funcs.FromChannelsLast.supported(y) else y.contiguous()
        return self.net(x)

    class _Level(nn.Module):
        """One resolution of the U-net (reference :247-320)."""

        def __init__(self, num_inputs, num_outputs, next_level=None, num_us=None,
                     ksize=3, width=64, num_convs=2, output_type="linear",
                     normalize=True, normalization_type="batch", pooling="max",
                     activation="relu"):
            super(Autoencoder._Level, self).__init__()
            self.is_last = next_level is None
            common = dict(ksize=ksize, width=width, depth=num_convs, stride=1, pad=True,
                          normalize=normalize, normalization_type=normalization_type)
            if self.is_last:
                # (the reference does not forward `activation` here either)
                self.left = ConvChain(num_inputs, num_outputs, output_type=output_type, **common)
                return
            assert num_us is not None
            self.left = ConvChain(num_inputs, width, output_type=activation,
                                  activation=activation, **common)
            if pooling == "max":
                self.downsample = nn.MaxPool2d(2, 2)
            elif pooling == "average":
                self.downsample = nn.AvgPool2d(2, 2)
            elif pooling == "conv":
                self.downsample = nn.Conv2d(width, width, 2, stride=2)
            else:
                raise ValueError("unknown pooling'{}'".format(pooling))
            self.next_level = next_level
            self.right = ConvChain(num_us + width, num_outputs, output_type=output_type, **common)

        def forward(self, x, want_link=False):
            """want_link: the caller (the level above) is the only reader of the result and takes the `_AdjLink` of this
            level's last convolution (ConvChain.forward)."""
            if self.is_last:
                return self.left(x, want_link=want_link)
            left = self.left(x, want_link=isinstance(self.downsample, nn.MaxPool2d))
            if funcs.PoolSkip.supported(left, self.downsample):
                # down path and skip connection as one autograd node: their gradients meet in ONE pass -- which is also
                # the adjoint pass of the chain's last activation where that layer left a link
                known = funcs.known_amax(left)
                pooled, left = funcs.PoolSkip.apply(left, funcs.Conv3x3BiasActNHWC.adj_link_for(left))
                if known is not None:
                    funcs.tag_amax(left, known)
            else:
                pooled = self.downsample(left)
            if isinstance(self.downsample, (nn.MaxPool2d, nn.AvgPool2d)) and funcs.known_amax(left) is not None:
                funcs.tag_amax(pooled, funcs.known_amax(left))        # pooling grows no magnitude
            coarse = self.next_level(pooled, want_link=True)
            if funcs.upsample_cat_nhwc_supported(coarse, left):
                # (the upsampling's backward pass is also the adjoint pass of the coarser level's last activation)
                cat = funcs.UpsampleCatNHWC.apply(coarse, left, 0, 0, funcs.Conv3x3BiasActNHWC.adj_link_for(coarse))
                bound = funcs.bound_amax(funcs.known_amax(coarse), funcs.known_amax(left))
                return self.right(cat if bound is None else funcs.tag_amax(cat, bound), want_link=want_link)
            if funcs.upsample_cat_supported(coarse, left):
                return self.right(funcs.UpsampleCat.apply(coarse, left), want_link=want_link)   # one pass, same values
            up = F.interpolate(coarse, size=left.shape[-2:], mode="bilinear",
                               align_corners=False)
            return self.right(th.cat([up, left], 1), want_link=want_link)


def _ksize_of(kernels):
    k2 = kernels.shape[1]
    return int(np.sqrt(k2))


class KernelApply(nn.Module):
    """Applies kernel-based averaging to the input (reference :323-361).

    Args:
        softmax(bool): softmax-normalise the kernels over the taps of each output pixel.
        splat(bool): kernels are sample-centred (splat); they are transposed to the
            gather layout first.
    """

    def __init__(self, softmax=True, splat=True):
        super(KernelApply, self).__init__()
        self.softmax = softmax
        self.splat = splat

    def forward(self, data, kernels):
        """data [bs, c, h, w], kernels [bs, k*k, h, w] -> (output [bs, c, h, w], sum_w [bs, 1, h, w])."""
        bs, k2, h, w = kernels.shape
        k = _ksize_of(kernels)
        if self.softmax and kernels.is_cuda:
            # softmax over the taps followed by the weighted sum IS one initialisation call of the
            # fused progressive update, normalised: exp(g - max) / sum exp(g - max) -- one pass
            # over the logits instead of softmax's three plus KernelWeighting's one
            ok = (funcs.splat_update_supported(data, kernels) if self.splat
                  else funcs.gather_update_supported(data, kernels))
            if ok and kernels.dtype == th.float32:
                sum_r, sum_w, _ = funcs.SplatUpdate.apply(data, kernels, None, None, None, not self.splat)
                # the softmax weights sum to one: value 1, gradient exactly 0, still part of the graph
                return sum_r / sum_w, sum_w / sum_w
        # the boundary-level operators are fp32 (as the reference's): half logits are up-cast
        kernels = kernels.float().view(bs, k, k, h, w)
        data = data.float()
        if self.splat:
            kernels = funcs.Scatter2Gather.apply(kernels)
        if self.softmax:
            kernels = F.softmax(kernels.view(bs, k * k, h, w), dim=1).view(bs, k, k, h, w)
        output, sum_w = funcs.KernelWeighting.apply(data, kernels)
        return output, sum_w.unsqueeze(1)


class ProgressiveKernelApply(nn.Module):
    """Accumulates one sample's kernel-weighted contribution into running sums
    with a numerically-stable running softmax (reference :364-473).

    ``sum_r / sum_w`` is the normalised reconstruction after any number of calls:
        sum_r = sum_i sum_taps exp(kernels_i - max_w) * data_i
        sum_w = sum_i sum_taps exp(kernels_i - max_w)
    with ``max_w`` the running per-pixel maximum of the (gather-layout) logits.

    Args:
        splat(bool): kernels are sample-centred (splat) rather than gather kernels.
        fused(bool): allow the single-kernel HIP path when the operands qualify
            (ROCm tensors, fp32; splat kernels, or gather kernels where the strip kernels apply).  ``False`` forces the reference
            composition of the boundary-level operators (used by the tests to check
            one against the other on the GPU).
    """

    def __init__(self, splat=False, fused=True):
        super(ProgressiveKernelApply, self).__init__()
        self.splat = splat
        self.fused = fused

    def forward(self, data, kernels, sum_r, sum_w, max_w):
        """
        Args:
            data [bs, c, h, w]; kernels [bs, k*k, h, w];
            sum_r [bs, c, h, w] | sum_w [bs, 1, h, w] | max_w [bs, 1, h, w], or all None
            for the initialisation call.
        Returns:
            (sum_r, sum_w, max_w) updated.
        """
        if sum_r is None and (sum_w is not None or max_w is not None):
            LOG.error("sum_r is None, this is the initialization step: "
                      "sum_w and max_w should be None as well.")
            raise RuntimeError("all of sum_r, sum_w, max_w should be none")

        if self.splat and self.fused and funcs.splat_update_supported(data, kernels):
            return funcs.SplatUpdate.apply(data, kernels, sum_r, sum_w, max_w)
        if not self.splat and self.fused and funcs.gather_update_supported(data, kernels):
            return funcs.SplatUpdate.apply(data, kernels, sum_r, sum_w, max_w, True)
        return self._composed(data, kernels, sum_r, sum_w, max_w)

    def _composed(self, data, kernels, sum_r, sum_w, max_w):
        # Reference sequence, out of place (the reference mutates the gather tensor
        # with sub_/exp_; values and gradients are identical).
        bs, k2, h, w = kernels.shape
        k = _ksize_of(kernels)
        # the boundary-level operators are fp32 (as the reference's): half logits are up-cast
        kernels = kernels.float().view(bs, k, k, h, w)
        data = data.float()
        if self.splat:
            kernels = funcs.Scatter2Gather.apply(kernels)
        kmax = kernels.reshape(bs, k * k, h, w).max(1, keepdim=True)[0]

        if sum_r is None:
            max_w = kmax
            weights = th.exp(kernels - max_w.unsqueeze(1))
            sum_r, sum_w = funcs.KernelWeighting.apply(data.contiguous(), weights.contiguous())
            return sum_r, sum_w.unsqueeze(1), max_w

        new_max = th.max(kmax, max_w)
        scaler = th.exp(max_w - new_max)
        weights = th.exp(kernels - new_max.unsqueeze(1))
        new_r, new_w = funcs.KernelWeighting.apply(data.contiguous(), weights.contiguous())
        return sum_r * scaler + new_r, sum_w * scaler + new_w.unsqueeze(1), new_max
