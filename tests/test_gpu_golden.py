"""GPU: the HIP path against the fixtures captured from the reference package, and
Multisteps end to end (HIP fused splat + MIOpen convs) against the same fixtures."""
import os

import pytest
import torch as th

from helpers import (close, golden, module_scales, multisteps_fp64, multisteps_from_golden, no_worse_than,
                     progressive_fp64, run_progressive, state_close, t)

pytestmark = pytest.mark.gpu


def test_ops_match_reference_fixtures():
    from sbmc_amd import functions as F
    g = golden("ops.npz")
    for tag in ("a", "b", "c"):
        data = t(g[tag + ".data"], "cuda").requires_grad_()
        wts = t(g[tag + ".weights"], "cuda").requires_grad_()
        o, s = F.KernelWeighting.apply(data, wts)
        th.autograd.backward([o, s], [t(g[tag + ".d_output"], "cuda"), t(g[tag + ".d_sum_w"], "cuda")])
        close(o, g[tag + ".output"]); close(s, g[tag + ".sum_w"])
        close(data.grad, g[tag + ".d_data"]); close(wts.grad, g[tag + ".d_weights"])
        x = t(g[tag + ".s2g_in"], "cuda").requires_grad_()
        y = F.Scatter2Gather.apply(x)
        y.backward(t(g[tag + ".s2g_gout"], "cuda"))
        assert th.equal(y.detach().cpu(), t(g[tag + ".s2g_out"]))
        assert th.equal(x.grad.cpu(), t(g[tag + ".s2g_gin"]))


def test_kernel_apply_matches_reference_fixtures():
    from sbmc_amd import modules
    g = golden("modules.npz")
    for softmax in (False, True):
        for splat in (False, True):
            tag = "ka.sm%d.sp%d." % (softmax, splat)
            d = t(g["ka.data"], "cuda").requires_grad_()
            kk = t(g["ka.kernels"], "cuda").requires_grad_()
            o, s = modules.KernelApply(softmax=softmax, splat=splat)(d, kk)
            th.autograd.backward([o, s], [t(g[tag + "g_output"], "cuda"), t(g[tag + "g_sum_w"], "cuda")])
            close(o, g[tag + "output"]); close(s, g[tag + "sum_w"])
            close(d.grad, g[tag + "d_data"]); close(kk.grad, g[tag + "d_kernels"])


@pytest.mark.parametrize("case,spp", [("p5", 3), ("p21", 2)])
@pytest.mark.parametrize("splat", [True, False])
@pytest.mark.parametrize("fused", [True, False])
def test_progressive_matches_reference_fixtures(case, spp, splat, fused):
    from sbmc_amd import modules
    g = golden("modules.npz")
    tag = "%s.sp%d." % (case, splat)
    datas = [t(g[tag + "data%d" % i]) for i in range(spp)]
    kerns = [t(g[tag + "kernels%d" % i]) for i in range(spp)]
    grads = [t(g[tag + "g%d" % i]) for i in range(3)]
    mod = modules.ProgressiveKernelApply(splat=splat, fused=fused)
    out, dd, dk = run_progressive(mod, datas, kerns, grads, "cuda")
    state_close(out, [t(g[tag + n]) for n in ("sum_r", "sum_w", "max_w")], what=tag,
                truth=lambda: progressive_fp64(datas, kerns, splat=splat)[0])
    for i in range(spp):
        close(dd[i], g[tag + "d_data%d" % i], what="d_data")
        close(dk[i], g[tag + "d_kernels%d" % i], what="d_kernels")


def test_reference_kats_on_gpu():
    """reference tests/test_modules.py:63-140 and tests/test_functions.py:72-103 on ROCm tensors"""
    from sbmc_amd import functions as F, modules
    bs, c, h, w, k = 4, 5, 16, 16, 3
    y, x, val = h // 2, w // 2, 1.43
    data = th.zeros(bs, c, h, w, device="cuda")
    weights = th.zeros(bs, k * k, h, w, device="cuda")
    data[0, 0, y, x] = val
    weights[0, :, y, x] = 1.0
    for splat in (True, False):
        out, sum_w = modules.KernelApply(softmax=False, splat=splat)(data, weights)
        assert out[0, 0, y, x].item() == pytest.approx(val, abs=1e-4)
        pout, psw, pmw = modules.ProgressiveKernelApply(splat=splat)(data, weights, None, None, None)
        assert pout[0, 0, y, x].item() == pytest.approx(val, abs=1e-4)
        if splat:
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    assert out[0, 0, y + dy, x + dx].item() == pytest.approx(val, abs=1e-4)
                    assert sum_w[0, 0, y + dy, x + dx].item() == pytest.approx(1, abs=1e-4)
                    assert pout[0, 0, y + dy, x + dx].item() == pytest.approx(val, abs=1e-4)
        else:
            assert sum_w[0, 0, y, x].item() == pytest.approx(k * k, abs=1e-4)
            assert psw[0, 0, y, x].item() == pytest.approx(k * k, abs=1e-4)
    for ksize in (3, 5, 7):
        d = th.full((3, 5, 16, 16), 7.0, device="cuda", requires_grad=True)
        wt = th.ones(3, ksize, ksize, 16, 16, device="cuda", requires_grad=True)
        o, s = F.KernelWeighting.apply(d, wt)
        og = th.zeros_like(o)
        og[1, 2, 8, 8] = 1.1
        o.backward(og)
        gd = d.grad.clone()
        p = ksize // 2
        assert th.allclose(gd[1, 2, 8 - p:8 + p + 1, 8 - p:8 + p + 1], th.full((ksize, ksize), 1.1, device="cuda"))
        gd[1, 2, 8 - p:8 + p + 1, 8 - p:8 + p + 1] = 0
        assert gd.abs().max().item() == 0.0
        assert wt.grad[1, ksize - 1, ksize - 1, 8, 8].item() == pytest.approx(7.7, abs=1e-3)


def test_gradcheck_on_gpu():
    """reference tests/test_functions.py:105-144,187-208 with the reference tolerances"""
    from torch.autograd import gradcheck
    from sbmc_amd import functions as F
    th.manual_seed(0)
    data = (2 * th.randn(2, 3, 16, 16)).cuda()
    wts = th.randn(2, 3, 3, 16, 16).cuda()
    assert gradcheck(F.KernelWeighting.apply, (data.clone().requires_grad_(), wts), eps=1e-4, atol=5e-2, rtol=5e-4)
    assert gradcheck(F.KernelWeighting.apply, (data, wts.clone().requires_grad_()), eps=1e-4, atol=5e-2, rtol=5e-4)
    x = th.randn(2, 3, 3, 32, 32).cuda().requires_grad_()
    assert gradcheck(F.Scatter2Gather.apply, (x,), eps=1e-4, atol=5e-2, rtol=5e-4)


def test_multisteps_on_gpu_matches_reference_fixture():
    from sbmc_amd import losses
    from sbmc_amd.utils import crop_like
    g, model, batch = multisteps_from_golden("cuda")
    target = batch.pop("target_image")
    model.train(False)
    with th.no_grad():
        out = model(batch)["radiance"]
    close(out, g["eval.radiance"], rtol=1e-5, what="eval output")   # (width 8: MIOpen convolutions, generic 1x1 path;
    model.train(True)                                               #  the production widths: the test below)
    res = model(batch)["radiance"]
    close(res, g["train.radiance"], rtol=1e-5, what="train output")
    loss = losses.TonemappedRelativeMSE()(res, crop_like(target, res))
    close(loss, g["train.loss"], rtol=1e-5, what="loss")
    loss.backward()
    # parameter gradients: sums over every pixel and sample whose fp32 value depends on the order of addition
    # (MIOpen's weight-gradient kernels, the 1x1 kernels' per-workgroup partial sums): within 1e-5 of a float64
    # evaluation of the model, or no further from it than twice the reference fixture is (scales per module)
    nf, ngf, width, ew, ks, nsteps = [int(v) for v in g["meta"]]
    m64 = multisteps_fp64(model, (nf, ngf), dict(width=width, embedding_width=ew, ksize=ks, nsteps=nsteps)).train(True)
    o64 = m64({k: v.cpu().double() for k, v in batch.items()})["radiance"]
    losses.TonemappedRelativeMSE()(o64, crop_like(target.cpu().double(), o64)).backward()
    g64 = {k: q.grad for k, q in m64.named_parameters()}
    scales = module_scales(g64)
    for k, p in model.named_parameters():
        no_worse_than(p.grad, t(g["grad." + k]), g64[k], what="grad " + k, scale=scales[k])


@pytest.mark.parametrize("case", ["k5", "k21", "k21c"])
def test_multisteps_production_width_on_gpu_matches_reference_fixture(case):
    """The kernels that carry the timed step -- split-precision 3x3 convolutions (csrc/conv3x3.hip), split 1x1 layers
    and the wide 441-channel gradient (csrc/pointwise.hip), the fused splat -- inside the reference's
    Multisteps(93, 3, width 128, k = 5 / 21) against what the REFERENCE computed for the same seeded model and batch
    (tests/golden/multisteps_wide.npz): outputs and loss at 1e-5, gradients against the float64 yardstick as in
    test_host_golden.wide_fixture_checks.  Asserts that those kernels, not a library path, ran."""
    from test_host_golden import wide_fixture_checks
    from sbmc_amd import functions as F
    calls = []
    F.enable_kernel_timing(calls)
    try:
        report = wide_fixture_checks(case, "cuda")
    finally:
        F.enable_kernel_timing(None)
    names = [c[0] for c in calls]

    def count(prefix):
        return sum(1 for n in names if n.startswith(prefix))
    # per forward pass: 3 U-nets x 15 convolutions; the three embeddings' 1x1 chains in ONE fused pass each (round 6:
    # csrc/pointwise_chain.hip) and the regressor's -- all three layers at k = 5 (25 logits), its first two + the
    # 441-channel layer's own kernel at k = 21; the fixture check runs an eval and a train forward and one backward (layer
    # by layer: 12 launches)
    assert count("conv3x3_fwd") == 2 * 45, count("conv3x3_fwd")
    assert count("conv3x3_bwd_weight") == 45 and count("conv3x3_bwd_data") == 45
    assert count("pointwise_chain_fwd 128x128x128<-") == 2 * 3, names
    assert count("pointwise_chain_fwd 128x128x25<-" if case == "k5" else "pointwise_chain_fwd 128x128<-") == 2, names
    assert count("pointwise_fwd ") == (0 if case == "k5" else 2), count("pointwise_fwd ")
    assert count("pointwise_bwd ") == 11 + (1 if case == "k5" else 0), names
    # k = 21: the 441-channel layer's backward is the ONE-PASS kernel (round 5), reached through the splat's bound word
    assert count("pointwise_wide_bwd") == (0 if case == "k5" else 1) and count("pointwise_gw_wide") == 0, names
    assert any(n.startswith("splat") for n in names), set(names)
    worst = max(report.items(), key=lambda kv: kv[1][0])
    print("%s: worst gradient %s at %.2e of its scale from float64 (reference %.2e)" % (case, worst[0], *worst[1]))


def test_native_library_is_loaded():
    """The GPU tests must run on the hand-written kernels, not on a fallback."""
    from sbmc_amd import _lib
    _lib.lib()
    assert "libsbmc_hip.so" in open("/proc/self/maps").read()


def test_product_never_imports_the_oracle():
    """Nothing under sbmc_amd/ may import or reference oracle/ (it is test infrastructure)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in glob.glob(os.path.join(root, "sbmc_amd", "**", "*.py"), recursive=True):
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), path
        assert "sbmc_oracle" not in src, path


def test_kpcn_on_gpu_matches_reference_fixture():
    from test_host_golden import _kpcn_from_golden
    g, model, data = _kpcn_from_golden("cuda")
    res = model(data)
    # 1e-5 of a float64 evaluation of the same model, or no further from it than twice the reference's own fixture
    from helpers import KernelApplyFP64
    _, m64, d64 = _kpcn_from_golden("cpu")
    m64.double()
    m64.kernel_apply = KernelApplyFP64()
    with th.no_grad():
        r64 = m64({k: v.double() for k, v in d64.items()})
    for k in ("radiance", "diffuse", "specular"):
        no_worse_than(res[k], t(g["out." + k]), r64[k], what=k)


@pytest.mark.parametrize("tag", ["splat", "gather", "pixel"])
def test_multisteps_odd_sizes_on_gpu(tag):
    from test_host_golden import _multisteps_odd
    g, model, batch = _multisteps_odd(tag, "cuda")
    with th.no_grad():
        out = model(batch)["radiance"]
        # 1e-5 of a float64 evaluation of the same model, or no further from it than twice the reference's own fixture
        m64 = multisteps_fp64(model, (5, 3), dict(width=4, embedding_width=4, ksize=3, nsteps=3, splat=(tag != "gather"),
                                                  pixel=(tag == "pixel"))).train(False)
        o64 = m64({k: v.cpu().double() for k, v in batch.items()})["radiance"]
    no_worse_than(out, t(g[tag + ".eval.radiance"]), o64, what=tag)
