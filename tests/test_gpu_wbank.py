"""The weight bank (sbmc_amd/wbank.py, csrc/conv3x3.hip `sbmc_wbank_*`): the weight norm of many layers per launch
(reference sbmc/modules.py:85-94, 178-188: torch's old-style weight norm on every convolution), the 3 x 3 layers'
prepared weights on the way, against torch's own `_weight_norm` and its autograd; the zeroed amax words the passes
raise (ABI 5); and the words travelling with halo rows (sbmc_amd/dist.py)."""
import os

import pytest
import torch as th
import torch.nn as nn
import torch.nn.functional as F

from helpers import bias_term_sums, close, module_scales, multisteps_fp64, no_worse_than

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _wn_conv(cin, cout, k, seed):
    th.manual_seed(seed)
    conv = nn.utils.weight_norm(nn.Conv2d(cin, cout, k, padding=k // 2))
    with th.no_grad():
        conv.weight_g.mul_(th.empty_like(conv.weight_g).uniform_(-2.0, 2.0))       # signs and magnitudes vary
    return conv.to(DEV)


def test_bank_matches_torch_weight_norm_forward_and_backward():
    """w, and (gv, gg) for gradients in three layouts: contiguous, channels-last (what conv3_wgrad writes), and
    none at all (a layer the loss does not depend on: zeros)."""
    from sbmc_amd.wbank import WeightBank
    shapes = [(128, 128, 3), (96, 128, 1), (256, 128, 3), (384, 128, 3), (128, 441, 1), (32, 8, 5)]
    convs = [_wn_conv(ci, co, k, 10 + i) for i, (ci, co, k) in enumerate(shapes)]
    for c in convs:
        assert WeightBank.takes(c)
    ws = WeightBank(convs).weights()
    refs = [th._weight_norm(c.weight_v, c.weight_g, 0) for c in convs]
    for w, r in zip(ws, refs):
        assert w.shape == r.shape
        close(w, r, 1e-6, "w")
    g = th.Generator().manual_seed(5)
    gws = []
    for i, r in enumerate(refs):
        gw = th.randn(r.shape, generator=g).to(DEV)
        if i % 3 == 1:
            gw = gw.contiguous(memory_format=th.channels_last)
        gws.append(gw)
    skip = 4                                             # this one gets no gradient
    th.autograd.backward([w for i, w in enumerate(ws) if i != skip], [gw for i, gw in enumerate(gws) if i != skip])
    got = [(c.weight_v.grad.clone(), c.weight_g.grad.clone()) for c in convs]
    for c in convs:
        c.weight_v.grad = c.weight_g.grad = None
    th.autograd.backward([r for i, r in enumerate(refs) if i != skip], [gw for i, gw in enumerate(gws) if i != skip])
    for i, c in enumerate(convs):
        if i == skip:
            assert got[i][0].abs().max().item() == 0 and got[i][1].abs().max().item() == 0
            continue
        close(got[i][0], c.weight_v.grad, 1e-5, "gv %d" % i)
        close(got[i][1], c.weight_g.grad, 1e-5, "gg %d" % i)


def test_bank_splits_more_layers_than_one_launch_takes():
    from sbmc_amd import _lib
    from sbmc_amd.wbank import WeightBank
    convs = [_wn_conv(16, 8 + i, 1, i) for i in range(_lib.WBANK_MAX + 5)]
    ws = WeightBank(convs).weights()
    for c, w in zip(convs, ws):
        close(w, th._weight_norm(c.weight_v, c.weight_g, 0), 1e-6, "w")
    sum(w.sum() for w in ws).backward()
    for c in convs:
        r = th._weight_norm(c.weight_v.detach().requires_grad_(), c.weight_g.detach().requires_grad_(), 0)
        assert c.weight_v.grad is not None and th.isfinite(c.weight_v.grad).all()
        assert r.shape == c.weight_v.shape


@pytest.mark.parametrize("cin,cout", [(128, 128), (384, 128), (256, 512)])
def test_bank_prepared_weights_convolve_like_float64(cin, cout):
    """The prepared forms that ride on a bank's weight: the convolution through them (forward with the fused
    epilogue, data and weight gradient) against float64, at the bound of tests/test_gpu_conv3x3.py."""
    from sbmc_amd import functions as funcs
    from sbmc_amd.wbank import WeightBank
    conv = _wn_conv(cin, cout, 3, 3)
    (w,) = WeightBank([conv]).weights()
    assert getattr(w, "_sbmc_wp", None) is not None
    th.manual_seed(1)
    x = th.randn(1, cin, 37, 50, device=DEV).contiguous(memory_format=th.channels_last).requires_grad_()
    gy = th.randn(1, cout, 37, 50, device=DEV).contiguous(memory_format=th.channels_last)
    calls = []
    orig = funcs.Conv3x3NHWC._absmax
    funcs.Conv3x3NHWC._absmax = staticmethod(lambda t: (calls.append(t.shape), orig(t))[1])
    try:
        y, _ = funcs.Conv3x3BiasActNHWC.apply(x, w, conv.bias, 2, 0.01)
        y.backward(gy)
    finally:
        funcs.Conv3x3NHWC._absmax = staticmethod(orig)
    assert len(calls) == 1, calls                        # x only (untagged here): none for the weights
    xd = x.detach().double().requires_grad_()
    vd, gd = conv.weight_v.detach().double().requires_grad_(), conv.weight_g.detach().double().requires_grad_()
    bd = conv.bias.detach().double()
    wd = th._weight_norm(vd, gd, 0)
    yd = F.leaky_relu(F.conv2d(xd, wd, bd, padding=1), 0.01)
    yd.backward(gy.double())
    for a, r, what in ((y, yd, "y"), (x.grad, xd.grad, "gx"), (conv.weight_v.grad, vd.grad, "gv"),
                       (conv.weight_g.grad, gd.grad, "gg")):
        e = (a.double() - r).abs().max().item() / r.abs().max().item()
        assert e <= 1e-5, (what, e)


def test_model_with_and_without_banks_agrees(monkeypatch):
    """A small Multisteps training step: losses equal, every parameter gradient within 1e-5 of its scale whether the
    weight norm runs layer by layer in torch or in the banks."""
    from sbmc_amd import Multisteps, losses
    from sbmc_amd.utils import crop_like
    th.manual_seed(0)
    ctor = ((12, 3), dict(width=128, embedding_width=128, ksize=5, nsteps=2))
    model = Multisteps(*ctor[0], **ctor[1]).to(DEV)
    batch = {"radiance": th.rand(1, 2, 3, 32, 48, device=DEV), "features": th.rand(1, 2, 12, 32, 48, device=DEV),
             "global_features": th.rand(1, 3, 1, 1, device=DEV)}
    target = th.rand(1, 3, 32, 48, device=DEV)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SBMC_WBANK", mode)
        model.zero_grad(set_to_none=True)
        out = model(batch)["radiance"]
        loss = losses.TonemappedRelativeMSE()(out, crop_like(target, out))
        loss.backward()
        res[mode] = (loss.item(), {k: p.grad.clone() for k, p in model.named_parameters()})
        for m in model.modules():
            assert "_sbmc_bank_w" not in m.__dict__          # nothing stale left on the modules
    assert abs(res["0"][0] - res["1"][0]) <= 1e-6 * abs(res["0"][0])
    # against a float64 evaluation of the model: either way of taking the weight norm within 1e-5 of the module's gradient
    # scale, or no further from it than twice the other (a bias gradient also to 8 ulp of its terms' magnitudes)
    m64 = multisteps_fp64(model, *ctor).train(True)
    terms = bias_term_sums(m64)
    o64 = m64({k: v.cpu().double() for k, v in batch.items()})["radiance"]
    losses.TonemappedRelativeMSE()(o64, crop_like(target.cpu().double(), o64)).backward()
    g64 = {k: q.grad for k, q in m64.named_parameters()}
    scales = module_scales(g64)
    for k in res["0"][1]:
        tk = terms.get(k) if k.endswith(".bias") else None
        no_worse_than(res["1"][1][k], res["0"][1][k], g64[k], what=k + " (banks)", scale=scales[k], terms=tk)
        no_worse_than(res["0"][1][k], res["1"][1][k], g64[k], what=k + " (torch)", scale=scales[k], terms=tk)


def test_amax_words_come_zeroed_and_are_raised():
    from sbmc_amd import functions as funcs
    dev = th.device(DEV, 0)
    words = [funcs.amax_word(dev) for _ in range(funcs._AmaxArena.SIZE + 3)]      # crosses into a second block
    assert all(int(w.item()) == 0 for w in words[::257])
    x = th.randn(2, 8, 12, 20, device=dev)
    xin, amax = funcs.ToChannelsLast.apply(x, True)
    assert amax.view(th.float32).item() == x.abs().max().item()
    y = xin.clone(memory_format=th.channels_last)
    out, a2 = funcs.BiasActNHWC.apply(y, th.zeros(8, device=dev), 0, 0.0, True)
    assert a2.view(th.float32).item() == x.abs().max().item()


def test_amax_word_travels_with_the_halo_rows():
    """`halo_pad` / `halo_refresh` of a tagged channels-last map in loop-back: the padded map is tagged with a word
    that bounds it, nobody runs an absmax pass, and a neighbour's larger word raises this rank's."""
    from sbmc_amd import dist as sdist, functions as funcs
    from sbmc_amd.halo import HaloChannel, rows_run
    from test_gpu_halo import _loop
    part = _loop()
    dev = th.device(DEV, 0)
    x = th.randn(1, 32, 12, 20, device=dev).contiguous(memory_format=th.channels_last)
    word = th.tensor([7.5], device=dev).view(th.int32)              # a bound above max |x|
    funcs.tag_amax(x, word)
    y = sdist.halo_pad(x, 1, part, True)
    assert funcs.known_amax(y) is not None and funcs.known_amax(y).view(th.float32).item() == 7.5
    z = sdist.halo_refresh(y, 1, part, True)
    assert funcs.known_amax(z) is not None and funcs.known_amax(z).view(th.float32).item() == 7.5
    # a larger word arrives with the rows from "above": the receiver's word is raised to it
    ch = part.channel
    big = th.tensor([100.0], device=dev).view(th.int32)
    mine = th.tensor([3.0], device=dev).view(th.int32)
    ch.put(rows_run(x, 0, 1, True), rows_run(x, 11, 12, True), amax=big)
    into = th.empty(1, 32, 2, 20, device=dev).contiguous(memory_format=th.channels_last)
    ch.get(rows_run(into, 0, 1, True), rows_run(into, 1, 2, True), amax=mine)
    assert mine.view(th.float32).item() == 100.0
    # untagged map: one absmax pass by halo_pad itself, none after
    u = th.randn(1, 32, 12, 20, device=dev).contiguous(memory_format=th.channels_last)
    yu = sdist.halo_pad(u, 1, part, True)
    assert funcs.known_amax(yu).view(th.float32).item() == u.abs().max().item()
    ch.check()
