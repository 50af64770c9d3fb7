import os

import numpy as np
import torch as th

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


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
