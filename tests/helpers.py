import os
import sys

import numpy as np
import torch as th

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)      # make_golden.py: the seeded inputs the generator and the tests share


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def t(a, device="cpu"):
    return th.from_numpy(np.asarray(a)).to(device)


def close(a, b, rtol=1e-5, what=""):
    """|a - b| <= rtol * max|b| + rtol * |b|  (north_star: 1e-5 relative fp32, abs-scaled for
    mixed-sign sums, SURVEY.md section 7 hard part 2)."""
    a = a.detach().cpu().double()
    b = b.detach().cpu().double() if isinstance(b, th.Tensor) else th.from_numpy(np.asarray(b)).double()
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    if b.numel() == 0:
        return
    scale = b.abs().max().item()
    err = (a - b).abs()
    bound = rtol * scale + rtol * b.abs()
    bad = err > bound
    assert not bad.any(), "%s: max err %.3e (scale %.3e), %d/%d bad" % (
        what, err.max().item(), scale, int(bad.sum()), b.numel())


def run_progressive(mod_fn, data_list, kern_list, grads, device):
    """S progressive updates, then backward with upstream grads on all three outputs."""
    datas = [d.detach().to(device).requires_grad_() for d in data_list]
    kerns = [k.detach().to(device).requires_grad_() for k in kern_list]
    sr = sw = mw = None
    for d, k in zip(datas, kerns):
        sr, sw, mw = mod_fn(d, k, sr, sw, mw)
    th.autograd.backward([sr, sw, mw], [g.to(device) for g in grads])
    return (sr, sw, mw), [d.grad for d in datas], [k.grad for k in kerns]


def multisteps_from_golden(device="cpu"):
    from sbmc_amd import Multisteps
    g = golden("multisteps.npz")
    nf, ngf, width, ew, ks, nsteps = [int(v) for v in g["meta"]]
    model = Multisteps(nf, ngf, width=width, embedding_width=ew, ksize=ks, nsteps=nsteps)
    sd = {k[3:]: t(g[k]) for k in g.files if k.startswith("sd.")}
    model.load_state_dict(sd, strict=True)
    batch = {k[3:]: t(g[k], device) for k in g.files if k.startswith("in.")}
    return g, model.to(device), batch


def gather_logits_fp64(kernels):
    """Scatter2Gather (reference src/scatter2gather.cpp:34-47) restated with torch slicing, any
    dtype (used in float64): kernels [bs, k*k, h, w] sample-centred -> [bs, k*k, h, w] gather layout,
    out[dy*k+dx, Y, X] = in[(2p-dy)*k + (2p-dx), Y+dy-p, X+dx-p], 0 outside the image."""
    bs, k2, h, w = kernels.shape
    k = int(round(k2 ** 0.5))
    p = (k - 1) // 2
    padded = th.nn.functional.pad(kernels.view(bs, k, k, h, w), (p, p, p, p))
    rows = []
    for dy in range(k):
        for dx in range(k):
            rows.append(padded[:, 2 * p - dy, 2 * p - dx, dy:dy + h, dx:dx + w])
    return th.stack(rows, 1)


def progressive_fp64(datas, kerns, grads=None, splat=True):
    """The reference's ProgressiveKernelApply chain (sbmc/modules.py:422-471) evaluated in float64 by the
    oracle's double instantiation of its operators (oracle/sbmc_oracle_ops.inc, OpenMP: fast enough for
    full-width bands).  Same contract as `progressive_fp64_torch`, against which it is checked on the CPU."""
    from oracle import sbmc_oracle as orc
    orc.lib()
    datas = [d.detach().double().requires_grad_() for d in datas]
    kerns = [k.detach().double().requires_grad_() for k in kerns]
    st = (None, None, None)
    for d, kk in zip(datas, kerns):
        st = orc.progressive_kernel_apply(d, kk, *st, splat=splat)
    if grads is None:
        return st, None, None
    th.autograd.backward(list(st), [g.double() for g in grads])
    return st, [d.grad for d in datas], [k.grad for k in kerns]


def progressive_fp64_torch(datas, kerns, grads=None, splat=True):
    """The reference's ProgressiveKernelApply chain (sbmc/modules.py:422-471) in float64 torch ops only
    (no native code at all; slow: small frames):
    the "truth" two fp32 implementations are both measured against where their own 1e-5 agreement is
    limited by cancellation (the routed arg-max element of d_kernels).
    Returns (state, d_datas, d_kerns) like helpers.run_progressive (gradients None without grads)."""
    datas = [d.detach().double().requires_grad_() for d in datas]
    kerns = [k.detach().double().requires_grad_() for k in kerns]
    sr = sw = mw = None
    for d, kk in zip(datas, kerns):
        bs, k2, h, w = kk.shape
        k = int(round(k2 ** 0.5))
        p = (k - 1) // 2
        g = gather_logits_fp64(kk) if splat else kk
        kmax = g.max(1, keepdim=True)[0]
        new_max = kmax if sr is None else th.max(kmax, mw)
        wts = th.exp(g - new_max)
        dpad = th.nn.functional.pad(d, (p, p, p, p))
        new_r = th.zeros_like(d)
        for dy in range(k):
            for dx in range(k):
                new_r = new_r + wts[:, dy * k + dx:dy * k + dx + 1] * dpad[:, :, dy:dy + h, dx:dx + w]
        new_w = wts.sum(1, keepdim=True)
        if sr is None:
            sr, sw, mw = new_r, new_w, new_max
        else:
            sc = th.exp(mw - new_max)
            sr, sw, mw = sr * sc + new_r, sw * sc + new_w, new_max
    if grads is None:
        return (sr, sw, mw), None, None
    th.autograd.backward([sr, sw, mw], [g.double() for g in grads])
    return (sr, sw, mw), [d.grad for d in datas], [k.grad for k in kerns]


class ProgressiveFP64(th.nn.Module):
    """ProgressiveKernelApply (reference sbmc/modules.py:422-471; splat=False: the gather ablation, no transposition) in
    torch ops of any dtype."""

    def __init__(self, splat=True):
        super(ProgressiveFP64, self).__init__()
        self.splat = splat

    def forward(self, data, kernels, sum_r, sum_w, max_w):
        bs, k2, h, w = kernels.shape
        k = int(round(k2 ** 0.5))
        p = (k - 1) // 2
        g = gather_logits_fp64(kernels) if self.splat else kernels
        kmax = g.max(1, keepdim=True)[0]
        new_max = kmax if sum_r is None else th.max(kmax, max_w)
        wts = th.exp(g - new_max)
        dpad = th.nn.functional.pad(data, (p, p, p, p))
        new_r = th.zeros_like(data)
        for dy in range(k):
            for dx in range(k):
                new_r = new_r + wts[:, dy * k + dx:dy * k + dx + 1] * dpad[:, :, dy:dy + h, dx:dx + w]
        new_w = wts.sum(1, keepdim=True)
        if sum_r is None:
            return new_r, new_w, new_max
        sc = th.exp(max_w - new_max)
        return sum_r * sc + new_r, sum_w * sc + new_w, new_max


def multisteps_fp64(model, ctor_args, ctor_kwargs):
    """A float64 twin of `model` (a Multisteps with splat kernels) on the CPU: same weights, torch ops only
    (the per-sample reference structure: no fused 1x1 kernels, no batched samples, the progressive update in
    torch float64).  The yardstick two fp32 evaluations are measured against; never "the reference"."""
    from sbmc_amd import Multisteps
    m64 = Multisteps(*ctor_args, pointwise_gemm=False, batch_samples=False, **ctor_kwargs)
    m64.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    m64.double()
    m64.kernel_update = ProgressiveFP64(splat=m64.splat)
    return m64


class KernelApplyFP64(th.nn.Module):
    """KernelApply(softmax=True, splat=False) (reference sbmc/modules.py:338-361) in torch ops of any dtype."""

    def forward(self, data, kernels):
        bs, k2, h, w = kernels.shape
        k = int(round(k2 ** 0.5))
        p = (k - 1) // 2
        wts = th.softmax(kernels, 1)
        dpad = th.nn.functional.pad(data, (p, p, p, p))
        out = th.zeros_like(data)
        for dy in range(k):
            for dx in range(k):
                out = out + wts[:, dy * k + dx:dy * k + dx + 1] * dpad[:, :, dy:dy + h, dx:dx + w]
        return out, th.ones_like(out[:, :1])


def float64_twin(net, factory):
    """A float64 twin of a module on the CPU: `factory()` builds the same module anew (weight-normalised modules do not
    deepcopy), the weights are `net`'s."""
    twin = factory()
    twin.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
    return twin.double()


def module_scales(named_grads):
    """max |gradient| per MODULE ({parameter name: scale}): `weight_g` and `weight_v` of a weight-normalised
    convolution are two projections of one quantity, dL/dw, and weight_g's gradient is a cancellation
    residual of it -- its rounding error has the size of dL/dw's, not of its own value."""
    per = {}
    for k, g in named_grads.items():
        mod = k.rsplit(".", 1)[0]
        per[mod] = max(per.get(mod, 0.0), g.detach().abs().max().item())
    return {k: per[k.rsplit(".", 1)[0]] for k in named_grads}


def no_worse_than(a, ref32, truth, rtol=1e-5, slack=2.0, what="", scale=None, terms=None):
    """|a - truth| <= rtol-bound, OR no worse than `slack` x the error the reference-order fp32
    computation (`ref32`, the oracle) itself makes against the float64 truth, element by element max.
    For quantities whose fp32 value is a difference of two long sums: both implementations round, so
    demanding 1e-5 of their mutual difference would test rounding luck, not correctness."""
    a = a.detach().cpu().double()
    ref32 = ref32.detach().cpu().double()
    truth = truth.detach().cpu().double()
    scale = truth.abs().max().item() if scale is None else scale
    err = (a - truth).abs().max().item()
    ref_err = (ref32 - truth).abs().max().item()
    bound = max(rtol * scale, slack * ref_err)
    if terms is not None:                      # `a` is an fp32 sum whose terms' magnitudes add up to `terms` (bias_term_sums)
        bound = max(bound, SUM_ROUNDING * terms)
    assert err <= bound, "%s: err vs fp64 %.3e > max(%.1e * scale = %.3e, %.1f x oracle's own fp32 error %.3e%s)" % (
        what, err, rtol, rtol * scale, slack, ref_err, "" if terms is None else ", 2^-21 x sum |terms| = %.3e" % (SUM_ROUNDING * terms))


SUM_ROUNDING = 2.0 ** -21


def bias_term_sums(model64):
    """Hooks on every biased convolution of a float64 model: after its backward, {"<module>.bias": the largest over the output
    channels of sum |d loss / d output|} -- the sum of the MAGNITUDES of the terms a bias gradient adds up.  An fp32 sum
    of n terms of mixed sign is good to a few ulp of THAT (SUM_ROUNDING = 8 ulp: partial sums per thread, a tree over
    them), not of its own value: where a bias gradient cancels to a hundredth of its terms' magnitudes, two correct fp32
    evaluations differ by 1e-5 of it and land on either side of any such bound by rounding luck."""
    sums = {}

    def forward_hook(name):
        # (a hook on the output TENSOR, set in the forward: a module backward hook does not go with the in-place
        # activations behind the convolutions; a tensor hook sees the gradient of the value it was set on)
        def on_grad(g):
            sums[name + ".bias"] = sums.get(name + ".bias", 0.0) + g.detach().abs().sum((0, 2, 3)).max().item()

        def fn(mod, inp, out):
            if out.requires_grad:
                out.register_hook(on_grad)
        return fn
    for name, mod in model64.named_modules():
        if isinstance(mod, th.nn.Conv2d) and mod.bias is not None:
            mod.register_forward_hook(forward_hook(name))
    return sums


def close_sum(a, truth, abs_terms, rtol=1e-5, what="", extra=0.0):
    """`a`: fp32 sums of many terms of mixed sign (a weight or bias gradient: a sum over every pixel), `truth` the float64
    sums, `abs_terms` the sums of the terms' MAGNITUDES (same shape).  Each element is held to `close`'s bound (rtol of the
    tensor's scale + rtol of itself) or to 8 ulp of its terms' magnitudes (SUM_ROUNDING), whichever is larger: what an fp32
    accumulation warrants where the terms cancel.  extra: an absolute allowance on top (half-storage kernels: a few terms whose
    rounding to half fell the other way)."""
    a = a.detach().cpu().double()
    b = truth.detach().cpu().double()
    t_ = abs_terms.detach().cpu().double()
    assert a.shape == b.shape == t_.shape, "%s: shapes %s %s %s" % (what, tuple(a.shape), tuple(b.shape), tuple(t_.shape))
    scale = b.abs().max().item()
    err = (a - b).abs()
    bound = th.maximum(rtol * scale + rtol * b.abs(), SUM_ROUNDING * t_) + extra
    bad = err > bound
    assert not bad.any(), "%s: max err %.3e (scale %.3e), %d/%d beyond max(%.0e of scale, 2^-21 of sum |terms|)" % (
        what, err.max().item(), scale, int(bad.sum()), b.numel(), rtol)


def close_or_yardstick(a, ref32, truth_fn, rtol=1e-5, slack=2.0, what=""):
    """`a` within rtol of the reference-order fp32 result `ref32` -- or, where that fails because BOTH are
    rounding noise of a cancelling sum (the d_kernels element that receives the routed gradient of the running
    max), no further from the float64 evaluation `truth_fn()` than `slack` x `ref32` is (no_worse_than).
    Returns (error of a, error of ref32) against the truth when the yardstick was needed, else None."""
    try:
        close(a, ref32, rtol=rtol, what=what)
        return None
    except AssertionError:
        truth = truth_fn()
        no_worse_than(a, ref32, truth, rtol=rtol, slack=slack, what=what + " (vs float64)")
        t = truth.detach().cpu().double()
        return ((a.detach().cpu().double() - t).abs().max().item(), (ref32.detach().cpu().double() - t).abs().max().item())


def multisteps_wide(case, device="cpu"):
    """The production-width fixture (tests/golden/make_golden.py:gen_multisteps_wide): Multisteps(93, 3, width 128,
    embedding 128, 3 steps) under the fixture's seed, the seeded batch, and the check that BOTH reproduce what the
    reference saw (exact checksums of every parameter's and input's bit patterns) -- so that a failing comparison below means a
    wrong kernel, never a changed initialisation.  -> (fixture, model, batch, target)"""
    from make_golden import WIDE_CASES, bits_checksum, wide_inputs
    from sbmc_amd import Multisteps
    c = WIDE_CASES[case]
    # ("k21c", round 6: the case whose activations' pre-activations all lie well clear of zero -- the biases that do that
    # travel in the fixture, make_golden.clear_the_kinks)
    g = golden("multisteps_clear.npz" if c.get("clear") else "multisteps_wide.npz")
    th.manual_seed(c["seed"])
    model = Multisteps(93, 3, width=128, embedding_width=128, ksize=c["ksize"], nsteps=3)
    sd = model.state_dict()
    for k in g.files:
        if k.startswith(case + ".bias."):
            sd[k[len(case) + 6:]].copy_(th.from_numpy(np.asarray(g[k])))
    keys = [k[len(case) + 7:] for k in g.files if k.startswith(case + ".sdsum.")]
    assert sorted(keys) == sorted(sd.keys())
    for k in keys:
        assert np.array_equal(bits_checksum(sd[k]), g["%s.sdsum.%s" % (case, k)]), "seeded init differs from the reference's: " + k
    batch, target = wide_inputs(case)
    for k, v in batch.items():
        assert np.array_equal(bits_checksum(v), g["%s.insum.%s" % (case, k)]), k
    assert np.array_equal(bits_checksum(target), g[case + ".insum.target_image"])
    return g, model.to(device), {k: v.to(device) for k, v in batch.items()}, target.to(device)


def rel_close(a, b, rtol=1e-5, what=""):
    """Plain elementwise relative bound |a - b| <= rtol * |b| (+ 1e-30): for quantities that are sums of
    non-negative terms (splat outputs, weight sums, running maxima of logits are NOT -- they are differences only
    through their arguments), where no cancellation excuses an absolute scale."""
    a = a.detach().cpu().double()
    b = b.detach().cpu().double() if isinstance(b, th.Tensor) else th.from_numpy(np.asarray(b)).double()
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    err = (a - b).abs()
    bad = err > rtol * b.abs() + 1e-30
    assert not bad.any(), "%s: %d/%d beyond %.0e relative, worst %.3e at |b| = %.3e" % (
        what, int(bad.sum()), b.numel(), rtol, (err / b.abs().clamp_min(1e-300))[bad].max().item(),
        b.abs()[bad].min().item())


def state_close(out, ref, what="", rtol=1e-5, truth=None):
    """The splat's running state (sum_r, sum_w, max_w) for NON-NEGATIVE radiance: sum_r and sum_w are sums of
    non-negative terms -- plain elementwise 1e-5 relative, no absolute scale; max_w is a selection among the input
    logits -- equal to the bit.
    truth: a callable -> the same state in float64 (progressive_fp64).  `ref` is itself an fp32 evaluation -- a
    sequential sum of up to 3 x 441 terms per sample, whose own rounding reaches 1e-5 of an element now and then (a
    few elements in 10^4: measured) -- so an element that differs from it by more than rtol is held to the float64
    value instead: within rtol of it, or no further from it than twice `ref` is."""
    t64 = None
    for i, (a, b, n) in enumerate(zip(out, ref, ("sum_r", "sum_w", "max_w"))):
        b = b if isinstance(b, th.Tensor) else th.from_numpy(np.asarray(b))
        if n == "max_w":
            assert th.equal(a.detach().cpu().float(), b.detach().cpu().float()), "%s max_w: not the same logit selected" % what
            continue
        assert (b >= 0).all(), "state_close is for non-negative radiance"
        if truth is None:
            rel_close(a, b, rtol=rtol, what="%s %s" % (what, n))
            continue
        a64, b64 = a.detach().cpu().double(), b.detach().cpu().double()
        bad = (a64 - b64).abs() > rtol * b64.abs() + 1e-30
        if bad.any():
            if t64 is None:
                t64 = truth()
            t = t64[i].detach().cpu().double()
            ea, eb = (a64 - t).abs()[bad], (b64 - t).abs()[bad]
            ok = (ea <= rtol * t.abs()[bad] + 1e-30) | (ea <= 2.0 * eb)
            assert ok.all(), "%s %s: %d elements beyond %.0e of float64 AND beyond twice the fp32 oracle's own error" % (
                what, n, int((~ok).sum()), rtol)
