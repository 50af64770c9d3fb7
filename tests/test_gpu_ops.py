"""GPU parity: the gfx950 kernels (called through the C ABI) vs the CPU oracle.

Tolerance: north_star asks for 1e-5 relative fp32.  Sums of up to 441 mixed-sign
products are compared with an abs-scaled bound: |a - b| <= 1e-5 * max|b| + 1e-5 * |b|
(hard part 2).
"""
import pytest
import torch as th

from helpers import close_sum, float64_twin, no_worse_than, progressive_fp64, state_close

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def close(a, b, rtol=RTOL, what=""):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    assert a.shape == b.shape, what
    scale = b.abs().max().item() if b.numel() else 0.0
    err = (a - b).abs()
    bound = rtol * scale + rtol * b.abs()
    bad = err > bound
    assert not bad.any(), "%s: max err %.3e (scale %.3e), %d bad" % (
        what, err.max().item(), scale, int(bad.sum()))


SHAPES = [
    # bs, c, h, w, k
    (1, 3, 16, 16, 3),
    (2, 3, 19, 70, 5),      # ragged: w % 64 != 0, h % 4 != 0
    (1, 5, 9, 130, 7),
    (2, 1, 33, 65, 9),
    (1, 3, 40, 150, 21),    # the tuned K=21 instantiation, border + interior tiles
    (1, 9, 12, 20, 3),      # more channels than one kernel pass takes (groups of 8)
    (1, 3, 5, 7, 21),       # image smaller than the kernel
]


@pytest.mark.parametrize("bs,c,h,w,k", SHAPES)
def test_scatter2gather(oracle, bs, c, h, w, k):
    from sbmc_amd import functions as F
    th.manual_seed(1)
    x = th.randn(bs, k, k, h, w)
    ref = oracle.Scatter2Gather.apply(x)
    out = F.Scatter2Gather.apply(x.cuda())
    assert th.equal(out.cpu(), ref)  # pure permutation: bit exact


def test_scatter2gather_rectangular(oracle):
    from sbmc_amd import functions as F
    th.manual_seed(2)
    x = th.randn(2, 3, 7, 11, 80)
    assert th.equal(F.Scatter2Gather.apply(x.cuda()).cpu(), oracle.Scatter2Gather.apply(x))
    x = th.randn(1, 4, 6, 11, 30)  # even sizes: pad = (k-1)//2
    assert th.equal(F.Scatter2Gather.apply(x.cuda()).cpu(), oracle.Scatter2Gather.apply(x))


@pytest.mark.parametrize("bs,c,h,w,k", SHAPES)
def test_kernel_weighting_fwd_bwd(oracle, bs, c, h, w, k):
    from sbmc_amd import functions as F
    th.manual_seed(3)
    data = (2 * th.randn(bs, c, h, w)).requires_grad_()
    wts = th.randn(bs, k, k, h, w).requires_grad_()
    go = th.randn(bs, c, h, w)
    gs = th.randn(bs, h, w)
    o_ref, s_ref = oracle.KernelWeighting.apply(data, wts)
    th.autograd.backward([o_ref, s_ref], [go, gs])
    dg, wg = data.grad.clone(), wts.grad.clone()

    data_g = data.detach().cuda().requires_grad_()
    wts_g = wts.detach().cuda().requires_grad_()
    o, s = F.KernelWeighting.apply(data_g, wts_g)
    th.autograd.backward([o, s], [go.cuda(), gs.cuda()])
    close(o, o_ref, what="output")
    close(s, s_ref, what="sum_w")
    close(data_g.grad, dg, what="d_data")
    close(wts_g.grad, wg, what="d_weights")


def test_kernel_weighting_rectangular(oracle):
    from sbmc_amd import functions as F
    th.manual_seed(4)
    data = th.randn(1, 3, 20, 90)
    wts = th.randn(1, 3, 5, 20, 90)
    o_ref, s_ref = oracle.KernelWeighting.apply(data, wts)
    o, s = F.KernelWeighting.apply(data.cuda(), wts.cuda())
    close(o, o_ref)
    close(s, s_ref)


def _progressive(mod_fn, data_list, kern_list, grads, device):
    """Runs S progressive updates, then backward with upstream grads on all 3 outputs."""
    datas = [d.detach().to(device).requires_grad_() for d in data_list]
    kerns = [k.detach().to(device).requires_grad_() for k in kern_list]
    sr = sw = mw = None
    for d, k in zip(datas, kerns):
        sr, sw, mw = mod_fn(d, k, sr, sw, mw)
    th.autograd.backward([sr, sw, mw], [g.to(device) for g in grads])
    return (sr, sw, mw), [d.grad for d in datas], [k.grad for k in kerns]


@pytest.mark.parametrize("bs,c,h,w,k,spp", [
    (1, 3, 16, 16, 3, 3),
    (2, 3, 19, 70, 5, 2),
    (1, 5, 9, 130, 7, 2),
    (1, 3, 40, 150, 21, 3),
    (1, 3, 5, 7, 21, 2),
    (1, 8, 12, 66, 3, 2),
])
def test_fused_splat_update_vs_oracle(oracle, bs, c, h, w, k, spp):
    """Fused fwd+bwd == the reference composition (oracle ops + torch autograd), with
    non-trivial upstream gradients on sum_r, sum_w AND max_w so that the arg-max
    routing of the running max is exercised."""
    from sbmc_amd import modules
    th.manual_seed(5)
    datas = [th.rand(bs, c, h, w) * 2 for _ in range(spp)]
    kerns = [th.randn(bs, k * k, h, w) * 2 for _ in range(spp)]
    grads = [th.randn(bs, c, h, w), th.randn(bs, 1, h, w), th.randn(bs, 1, h, w)]

    ref_out, ref_dd, ref_dk = _progressive(
        lambda d, kk, a, b, m: oracle.progressive_kernel_apply(d, kk, a, b, m, splat=True),
        datas, kerns, grads, "cpu")
    fused = modules.ProgressiveKernelApply(splat=True)
    out, dd, dk = _progressive(fused, datas, kerns, grads, "cuda")
    state_close(out, ref_out, truth=lambda: progressive_fp64(datas, kerns)[0])
    for s in range(spp):
        close(dd[s], ref_dd[s], what="d_data[%d]" % s)
        close(dk[s], ref_dk[s], what="d_kernels[%d]" % s)


@pytest.mark.parametrize("splat", [True, False])
def test_composed_path_on_gpu_vs_oracle(oracle, splat):
    """The non-fused composition (Scatter2Gather + KernelWeighting HIP ops + torch) on GPU."""
    from sbmc_amd import modules
    th.manual_seed(6)
    bs, c, h, w, k, spp = 1, 3, 21, 77, 5, 2
    datas = [th.rand(bs, c, h, w) for _ in range(spp)]
    kerns = [th.randn(bs, k * k, h, w) for _ in range(spp)]
    grads = [th.randn(bs, c, h, w), th.randn(bs, 1, h, w), th.randn(bs, 1, h, w)]
    ref_out, ref_dd, ref_dk = _progressive(
        lambda d, kk, a, b, m: oracle.progressive_kernel_apply(d, kk, a, b, m, splat=splat),
        datas, kerns, grads, "cpu")
    mod = modules.ProgressiveKernelApply(splat=splat, fused=False)
    out, dd, dk = _progressive(mod, datas, kerns, grads, "cuda")
    for a, b in zip(out, ref_out):
        close(a, b)
    for s in range(spp):
        close(dd[s], ref_dd[s])
        close(dk[s], ref_dk[s])


def test_fused_forces_running_max_branches(oracle):
    """Data-dependent branches need their own test: sample 2 raises the max everywhere
    (spike), sample 3 nowhere, and exact ties between kmax and the running max."""
    from sbmc_amd import modules
    th.manual_seed(7)
    bs, c, h, w, k = 1, 3, 12, 70, 5
    base = th.randn(bs, k * k, h, w)
    spike = base.clone()
    spike[:, 7] += 30.0
    low = base - 50.0
    datas = [th.rand(bs, c, h, w) for _ in range(4)]
    kerns = [base, spike, low, spike.clone()]  # last one ties exactly with the running max
    grads = [th.randn(bs, c, h, w), th.randn(bs, 1, h, w), th.randn(bs, 1, h, w)]
    ref_out, ref_dd, ref_dk = _progressive(
        lambda d, kk, a, b, m: oracle.progressive_kernel_apply(d, kk, a, b, m, splat=True),
        datas, kerns, grads, "cpu")
    out, dd, dk = _progressive(modules.ProgressiveKernelApply(splat=True), datas, kerns, grads, "cuda")
    for a, b in zip(out, ref_out):
        close(a, b)
    for s in range(4):
        close(dd[s], ref_dd[s], what="d_data[%d]" % s)
        close(dk[s], ref_dk[s], what="d_kernels[%d]" % s)


def test_cpu_tensors_are_refused():
    from sbmc_amd import functions as F
    with pytest.raises(RuntimeError):
        F.KernelWeighting.apply(th.zeros(1, 3, 8, 8), th.zeros(1, 3, 3, 8, 8))


@pytest.mark.parametrize("bs,c,h,w,spp", [(1, 3, 30, 150, 3), (2, 3, 9, 70, 2), (1, 4, 5, 7, 4)])
def test_splat_all_vs_oracle(oracle, bs, c, h, w, spp):
    """All samples in one launch (functions.SplatAll) == S chained progressive updates of the
    reference composition, forward and backward, with upstream gradients on all three outputs."""
    from sbmc_amd import functions as F
    th.manual_seed(8)
    k = 21
    data = th.rand(bs, spp, c, h, w) * 2
    kern = th.randn(bs, spp, k * k, h, w) * 2
    grads = [th.randn(bs, c, h, w), th.randn(bs, 1, h, w), th.randn(bs, 1, h, w)]
    ref_out, ref_dd, ref_dk = _progressive(
        lambda d, kk, a, b, m: oracle.progressive_kernel_apply(d, kk, a, b, m, splat=True),
        [data[:, s] for s in range(spp)], [kern[:, s] for s in range(spp)], grads, "cpu")
    dg = data.cuda().requires_grad_()
    kg = kern.cuda().requires_grad_()
    assert F.splat_all_supported(dg, kg)
    out = F.SplatAll.apply(dg, kg)
    th.autograd.backward(out, [g.cuda() for g in grads])
    state_close(out, ref_out, truth=lambda: progressive_fp64([data[:, s] for s in range(spp)], [kern[:, s] for s in range(spp)])[0])
    for s in range(spp):
        close(dg.grad[:, s], ref_dd[s], what="d_data[%d]" % s)
        close(kg.grad[:, s], ref_dk[s], what="d_kernels[%d]" % s)


def test_splat_all_running_max_branches(oracle):
    """spike / never-raises / exact tie between a sample's max and the running max"""
    from sbmc_amd import functions as F
    th.manual_seed(9)
    bs, c, h, w, k = 1, 3, 8, 70, 21
    base = th.randn(bs, k * k, h, w)
    spike = base.clone()
    spike[:, 200] += 30.0
    kern = th.stack([base, spike, base - 50.0, spike.clone()], 1)
    data = th.rand(bs, 4, c, h, w)
    grads = [th.randn(bs, c, h, w), th.randn(bs, 1, h, w), th.randn(bs, 1, h, w)]
    ref_out, ref_dd, ref_dk = _progressive(
        lambda d, kk, a, b, m: oracle.progressive_kernel_apply(d, kk, a, b, m, splat=True),
        [data[:, s] for s in range(4)], [kern[:, s] for s in range(4)], grads, "cpu")
    dg, kg = data.cuda().requires_grad_(), kern.cuda().requires_grad_()
    out = F.SplatAll.apply(dg, kg)
    th.autograd.backward(out, [g.cuda() for g in grads])
    for a, b in zip(out, ref_out):
        close(a, b)
    for s in range(4):
        close(dg.grad[:, s], ref_dd[s], what="d_data[%d]" % s)
        close(kg.grad[:, s], ref_dk[s], what="d_kernels[%d]" % s)


def test_multisteps_batched_samples_equals_sequential():
    """Multisteps(k=21): the all-samples path (one regressor pass + SplatAll) vs the per-sample
    loop of the reference, outputs and parameter gradients."""
    from sbmc_amd import Multisteps
    th.manual_seed(10)
    kw = dict(width=8, embedding_width=8, ksize=21, nsteps=1)
    from helpers import module_scales, multisteps_fp64
    a = Multisteps(6, 3, batch_samples=True, **kw).cuda()
    b = Multisteps(6, 3, batch_samples=False, **kw).cuda()
    b.load_state_dict(a.state_dict())
    g = th.Generator().manual_seed(11)
    batch = {"radiance": th.empty(1, 3, 3, 40, 72).exponential_(1.0, generator=g).cuda(),
             "features": th.rand(1, 3, 6, 40, 72, generator=g).cuda(),
             "global_features": th.rand(1, 3, 1, 1, generator=g).cuda()}
    oa = a(batch)["radiance"]
    ob = b(batch)["radiance"]
    close(oa, ob, what="output")
    go = th.randn(oa.shape, generator=g).cuda()
    oa.backward(go)
    ob.backward(go)
    # parameter gradients: both within 1e-5 of a float64 evaluation of the model or no further from it than
    # twice the other path (helpers.no_worse_than, scales per module)
    m64 = multisteps_fp64(a, (6, 3), kw)
    o64 = m64({k: v.cpu().double() for k, v in batch.items()})["radiance"]
    o64.backward(go.cpu().double())
    g64 = {k: q.grad for k, q in m64.named_parameters()}
    scales = module_scales(g64)
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        no_worse_than(pa.grad, pb.grad, g64[n], what=n + " (batched)", scale=scales[n])
        no_worse_than(pb.grad, pa.grad, g64[n], what=n + " (sequential)", scale=scales[n])


def test_32spp_fused_vs_scatter2gather_dual_path():
    """BASELINE configs[4] shape of the sample loop: 32 spp, 21x21 kernels, the fused path
    (per-sample updates and the all-samples launch) vs the Scatter2Gather + KernelWeighting dual
    path on the GPU."""
    from sbmc_amd import functions as F, modules
    th.manual_seed(12)
    bs, c, h, w, k, spp = 1, 3, 24, 130, 21, 32
    data = th.rand(bs, spp, c, h, w, device="cuda")
    kern = th.randn(bs, spp, k * k, h, w, device="cuda")
    dual = modules.ProgressiveKernelApply(splat=True, fused=False)
    fused = modules.ProgressiveKernelApply(splat=True, fused=True)
    sa = sb = (None, None, None)
    for s in range(spp):
        sa = dual(data[:, s], kern[:, s], *sa)
        sb = fused(data[:, s], kern[:, s], *sb)
    sc = F.SplatAll.apply(data, kern)
    for a, b, c_ in zip(sa, sb, sc):
        close(b, a)
        close(c_, a)
    out = sa[0] / (sa[1] + 1e-8)
    close(sc[0] / (sc[1] + 1e-8), out)


@pytest.mark.parametrize("bs,c,h,w,k", [(1, 3, 1, 1, 3), (1, 3, 2, 3, 1), (3, 2, 4, 64, 1), (1, 3, 1, 200, 21),
                                         (1, 3, 100, 1, 21), (1, 3, 7, 63, 5), (1, 3, 8, 65, 21)])
def test_degenerate_shapes(oracle, bs, c, h, w, k):
    """1x1 images, 1x1 kernels (the reference's scatter2gather profile script uses ksize=1),
    single rows / columns, widths around the 64-lane strip."""
    from sbmc_amd import functions as F, modules
    th.manual_seed(13)
    data = th.rand(bs, c, h, w)
    kern = th.randn(bs, k * k, h, w)
    x5 = kern.view(bs, k, k, h, w)
    assert th.equal(F.Scatter2Gather.apply(x5.cuda()).cpu(), oracle.Scatter2Gather.apply(x5))
    o_ref, s_ref = oracle.KernelWeighting.apply(data, x5)
    o, s = F.KernelWeighting.apply(data.cuda(), x5.cuda())
    close(o, o_ref); close(s, s_ref)
    grads = [th.randn(bs, c, h, w), th.randn(bs, 1, h, w), th.randn(bs, 1, h, w)]
    ref_out, ref_dd, ref_dk = _progressive(
        lambda d, kk, a, b, m: oracle.progressive_kernel_apply(d, kk, a, b, m, splat=True),
        [data, data * 0.5], [kern, kern.flip(1)], grads, "cpu")
    out, dd, dk = _progressive(modules.ProgressiveKernelApply(splat=True), [data, data * 0.5],
                               [kern, kern.flip(1)], grads, "cuda")
    for a, b in zip(out, ref_out):
        close(a, b)
    for i in range(2):
        close(dd[i], ref_dd[i]); close(dk[i], ref_dk[i])


def test_empty_batch_is_a_no_op():
    from sbmc_amd import functions as F, modules
    d = th.zeros(0, 3, 8, 8, device="cuda")
    k = th.zeros(0, 9, 8, 8, device="cuda")
    o, s = F.KernelWeighting.apply(d, k.view(0, 3, 3, 8, 8))
    assert o.shape == (0, 3, 8, 8) and s.shape == (0, 8, 8)
    assert F.Scatter2Gather.apply(k.view(0, 3, 3, 8, 8)).shape == (0, 3, 3, 8, 8)
    r = modules.ProgressiveKernelApply(splat=True)(d, k, None, None, None)
    assert r[0].shape == (0, 3, 8, 8) and r[1].shape == (0, 1, 8, 8)


def test_bad_arguments_raise():
    from sbmc_amd import functions as F, halide_ops
    d = th.zeros(1, 3, 8, 8, device="cuda")
    w = th.zeros(1, 3, 3, 8, 8, device="cuda")
    with pytest.raises(RuntimeError):   # shape mismatch between data and weights
        halide_ops.kernel_weighting_cuda_float32(d, th.zeros(1, 3, 3, 8, 9, device="cuda"), th.empty_like(d),
                                                 th.empty(1, 8, 8, device="cuda"))
    with pytest.raises(RuntimeError):   # non-contiguous
        halide_ops.scatter2gather_cuda_float32(w.transpose(3, 4), th.empty_like(w))
    with pytest.raises(RuntimeError):   # wrong dtype
        halide_ops.scatter2gather_cuda_float32(w.double(), th.empty_like(w).double())
    with pytest.raises(RuntimeError):   # output aliasing the input
        halide_ops.scatter2gather_cuda_float32(w, w)
    with pytest.raises(RuntimeError):   # partial running state
        F.SplatUpdate.apply(d, th.zeros(1, 9, 8, 8, device="cuda"), None, th.zeros(1, 1, 8, 8, device="cuda"), None)


@pytest.mark.parametrize("bs,c,h,w,spp", [(1, 3, 30, 150, 3), (1, 3, 6, 9, 2)])
def test_fp16_logits_vs_oracle(oracle, bs, c, h, w, spp):
    """fp16 logit storage (SURVEY row N4 / BASELINE configs[4]): the kernels read half logits and
    write half logit-gradients, all arithmetic in fp32.  The oracle gets the same (exactly
    representable) logits in fp32: forward within 1e-5; d_kernels within half rounding (2^-11)."""
    from sbmc_amd import functions as F, modules
    th.manual_seed(14)
    k = 21
    data = th.rand(bs, spp, c, h, w) * 2
    kern_h = (th.randn(bs, spp, k * k, h, w) * 2).half()
    grads = [th.randn(bs, c, h, w), th.randn(bs, 1, h, w), th.randn(bs, 1, h, w)]
    ref_out, ref_dd, ref_dk = _progressive(
        lambda d, kk, a, b, m: oracle.progressive_kernel_apply(d, kk, a, b, m, splat=True),
        [data[:, s] for s in range(spp)], [kern_h[:, s].float() for s in range(spp)], grads, "cpu")
    # all samples per launch
    dg = data.cuda().requires_grad_()
    kg = kern_h.cuda().requires_grad_()
    assert F.splat_all_supported(dg, kg)
    out = F.SplatAll.apply(dg, kg)
    th.autograd.backward(out, [g.cuda() for g in grads])
    assert kg.grad.dtype == th.float16
    state_close(out, ref_out, truth=lambda: progressive_fp64([data[:, s] for s in range(spp)],
                                                             [kern_h[:, s].float() for s in range(spp)])[0])
    for s in range(spp):
        close(dg.grad[:, s], ref_dd[s], what="d_data")
        close(kg.grad[:, s].float(), ref_dk[s], rtol=1e-3, what="d_kernels (half)")
    # per-sample module interface
    upd = modules.ProgressiveKernelApply(splat=True)
    out2, dd2, dk2 = _progressive(upd, [data[:, s] for s in range(spp)],
                                  [kern_h[:, s] for s in range(spp)], grads, "cuda")
    for a, b in zip(out2, ref_out):
        close(a, b)
    for s in range(spp):
        close(dd2[s], ref_dd[s])
        close(dk2[s].float(), ref_dk[s], rtol=1e-3)


def test_multisteps_under_fp16_autocast():
    """fp16 activations end to end: backbone under torch.autocast(float16), half logits into the
    fused splat, fp32 accumulation.  Sanity against the fp32 model (not a parity claim)."""
    from sbmc_amd import Multisteps
    th.manual_seed(15)
    model = Multisteps(6, 3, width=16, embedding_width=16, ksize=21, nsteps=1).cuda()
    g = th.Generator().manual_seed(16)
    batch = {"radiance": th.empty(1, 2, 3, 40, 72).exponential_(1.0, generator=g).cuda(),
             "features": th.rand(1, 2, 6, 40, 72, generator=g).cuda(),
             "global_features": th.rand(1, 3, 1, 1, generator=g).cuda()}
    ref = model(batch)["radiance"]
    with th.autocast("cuda", dtype=th.float16):
        out = model(batch)["radiance"]
    assert out.dtype == th.float32 and th.isfinite(out).all()
    assert (out - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
    out.sum().backward()
    assert all(th.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    # inference: the 1x1 chains keep the fused kernels with half storage (functions.pointwise_half)
    from sbmc_amd import functions as F
    launches = []
    F.enable_kernel_timing(launches)
    with th.no_grad(), th.autocast("cuda", dtype=th.float16):
        out_inf = model(batch)["radiance"]
    F.enable_kernel_timing(None)
    assert sum(n.startswith("pointwise_fwd_f16") for n, _, _ in launches) == 6     # 3 layers x 2 chains
    assert (out_inf - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()


@pytest.mark.parametrize("act,slope", [(0, 0.0), (1, 0.0), (2, 0.01), (2, 0.2)])
def test_bias_act_kernels(act, slope):
    """Fused bias + activation (csrc/bias_act.hip) vs torch, forward and backward."""
    from sbmc_amd import functions as F
    th.manual_seed(17)
    for shape in ((2, 5, 12, 20), (1, 128, 8, 64), (3, 1, 4, 4)):
        y0 = th.randn(*shape, device="cuda")
        bias = th.randn(shape[1], device="cuda", requires_grad=True)
        x = y0.clone().requires_grad_()
        pre = x + bias.view(1, -1, 1, 1)
        ref = pre if act == 0 else th.nn.functional.leaky_relu(pre, slope if act == 2 else 0.0)
        g = th.randn_like(ref)
        ref.backward(g)
        gx_ref, gb_ref = x.grad.clone(), bias.grad.clone()
        x2 = y0.clone().requires_grad_()
        b2 = bias.detach().clone().requires_grad_()
        out = F.BiasAct.apply(x2 * 1.0, b2, act, slope)   # *1.0: BiasAct works in place on its input
        assert F.BiasAct.supported(out)
        out.backward(g)
        assert th.equal(out.detach(), ref.detach())
        assert th.equal(x2.grad, gx_ref)
        close(b2.grad, gb_ref, rtol=1e-5)


@pytest.mark.parametrize("shape", [
    # B, S, cin, cout, hw, context (0 none, 1 per image, 2 per pixel), act
    (2, 1, 128, 128, 1024, 0, 1), (3, 1, 93, 128, 1000, 0, 2), (4, 2, 128, 128, 260, 2, 1),
    (4, 2, 93, 128, 516, 1, 1), (2, 1, 128, 441, 388, 0, 0), (2, 2, 128, 441, 132, 2, 2),
    (1, 1, 7, 5, 4, 0, 1), (2, 1, 33, 200, 36, 0, 0), (6, 3, 64, 25, 640, 2, 0), (16, 8, 128, 128, 128 * 40 + 4, 2, 1),
])
def test_pointwise_layer_kernel(shape):
    """A whole 1x1-convolution layer in one MFMA pass (csrc/pointwise.hip) vs an fp64 torch
    restatement of conv1x1 + context term + bias + activation, forward and backward."""
    from sbmc_amd import functions as F
    B, S, cin, cout, hw, tm, act = shape
    slope = 0.01 if act == 2 else 0.0
    th.manual_seed(B * 1000 + cin + cout + hw)
    x0 = th.randn(B, cin, hw, device="cuda")
    w0 = th.randn(cout, cin, device="cuda") / cin ** 0.5
    b0 = th.randn(cout, device="cuda")
    t0 = None if tm == 0 else (th.randn(B // S, cout, device="cuda") if tm == 1
                               else th.randn(B // S, cout, hw, device="cuda"))
    assert F.pointwise_supported(x0, cout)

    def leaves(dtype):
        return [None if v is None else v.to(dtype).requires_grad_() for v in (x0, w0, b0, t0)]
    x, w, b, t = leaves(th.float64)
    pre = th.matmul(w, x) + b.view(1, -1, 1)
    if tm == 1:
        pre = pre + t.repeat_interleave(S, 0).unsqueeze(-1)
    elif tm == 2:
        pre = pre + t.repeat_interleave(S, 0)
    ref = pre if act == 0 else th.nn.functional.leaky_relu(pre, slope)
    g = th.randn(B, cout, hw, device="cuda")
    # no gradient through pre-activations at the kink: fp32 and fp64 may disagree on their sign
    g = g * (pre.detach().abs() > 1e-4).float()
    pre.retain_grad()
    ref.backward(g.double())
    x2, w2, b2, t2 = leaves(th.float32)
    out = F.PointwiseLayer.apply(x2, w2, b2, t2, S, act, slope)
    out.backward(g)
    close(out, ref.float(), rtol=1e-5)
    close(x2.grad, x.grad.float(), rtol=1e-5)
    # the sums over every pixel: 1e-5, or 8 ulp of the sum of their terms' magnitudes (helpers.close_sum)
    gza = pre.grad.abs()
    close_sum(w2.grad, w.grad, th.einsum("bop,bcp->oc", gza, x.detach().abs()), what="gw")
    close_sum(b2.grad, b.grad, gza.sum((0, 2)), what="gbias")
    if tm:
        close(t2.grad, t.grad.float(), rtol=1e-5)


@pytest.mark.parametrize("shape", [
    # B, mean_s, cin, cout, hw, act, needs gx
    (8, 8, 128, 128, 1028, 1, True), (6, 3, 93, 128, 260, 0, False), (4, 2, 64, 32, 64, 2, True),
    (4, 4, 128, 200, 132, 0, True),          # cout > 128: composed backward, gradients summed first
])
def test_pointwise_layer_with_sample_mean(shape):
    """PointwiseLayerMean: (y, mean of y over groups of samples); one backward for both gradients
    (the fused kernel reads gy + gmean / S) vs an fp64 restatement."""
    from sbmc_amd import functions as F
    B, ms, cin, cout, hw, act, needx = shape
    slope = 0.01 if act == 2 else 0.0
    th.manual_seed(sum(shape[:5]))
    x0 = th.randn(B, cin, hw, device="cuda")
    w0 = th.randn(cout, cin, device="cuda") / cin ** 0.5
    b0 = th.randn(cout, device="cuda")
    x, w, b = x0.double().requires_grad_(needx), w0.double().requires_grad_(), b0.double().requires_grad_()
    pre = th.matmul(w, x) + b.view(1, -1, 1)
    ref = pre if act == 0 else th.nn.functional.leaky_relu(pre, slope)
    ref_mean = ref.view(B // ms, ms, cout, hw).mean(1)
    g = th.randn(B, cout, hw, device="cuda") * (pre.detach().abs() > 1e-4).float()
    gm = th.randn(B // ms, cout, hw, device="cuda")
    if act:   # keep the mean's gradient away from the kink as well
        gm = gm * (pre.detach().abs() > 1e-4).view(B // ms, ms, cout, hw).all(1).float()
    pre.retain_grad()
    th.autograd.backward([ref, ref_mean], [g.double(), gm.double()])
    x2, w2, b2 = x0.clone().requires_grad_(needx), w0.clone().requires_grad_(), b0.clone().requires_grad_()
    y, ym = F.PointwiseLayerMean.apply(x2, w2, b2, None, 1, act, slope, ms)
    th.autograd.backward([y, ym], [g, gm])
    close(y, ref.float(), rtol=1e-5)
    close(ym, ref_mean.float(), rtol=1e-5)
    if needx:
        close(x2.grad, x.grad.float(), rtol=1e-5)
    gza = pre.grad.abs()
    close_sum(w2.grad, w.grad, th.einsum("bop,bcp->oc", gza, x.detach().abs()), what="gw")
    close_sum(b2.grad, b.grad, gza.sum((0, 2)), what="gbias")


def test_pointwise_chain_as_gemm_matches_convolution():
    """ConvChain(ksize=1) through the batched-GEMM + fused bias/activation path == the nn.Conv2d path."""
    from sbmc_amd import modules
    th.manual_seed(18)
    for out_type, actv in (("linear", "relu"), ("leaky_relu", "leaky_relu")):
        mk = lambda: modules.ConvChain(12, 7, ksize=1, width=16, depth=3, pad=False, activation=actv, output_type=out_type)
        chain = mk().cuda()
        x = th.randn(3, 12, 10, 24, device="cuda")
        xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
        ya = chain(xa)
        chain.pointwise_as_gemm = True
        from sbmc_amd import functions as F
        launches = []
        F.enable_kernel_timing(launches)
        yb = chain(xb)
        F.enable_kernel_timing(None)
        assert sum(n.startswith("pointwise_fwd") for n, _, _ in launches) == 3   # the fused MFMA layers ran
        chain.pointwise_as_gemm = False
        close(yb, ya, rtol=1e-5)
        g = th.randn_like(ya)
        ga = th.autograd.grad(ya, [xa] + list(chain.parameters()), g)
        gb = th.autograd.grad(yb, [xb] + list(chain.parameters()), g)
        # either path within 1e-5 of a float64 evaluation of the chain, or no further from it than twice the other
        c64 = float64_twin(chain, mk)
        x64 = x.cpu().double().requires_grad_()
        g64 = th.autograd.grad(c64(x64), [x64] + list(c64.parameters()), g.cpu().double())
        for a, b, t64 in zip(ga, gb, g64):
            no_worse_than(b, a, t64, what="fused")
            no_worse_than(a, b, t64, what="convolution")


@pytest.mark.parametrize("per_pixel", [True, False])
@pytest.mark.parametrize("act,slope", [(0, 0.0), (1, 0.0), (2, 0.01)])
def test_ctx_act_kernels(per_pixel, act, slope):
    """y[b,s] = act(y[b,s] + t[b] + bias): fused context-term pass vs torch, forward and backward."""
    from sbmc_amd import functions as F
    th.manual_seed(19)
    b, s, c, h, w = 2, 3, 6, 5, 16
    y0 = th.randn(b * s, c, h, w, device="cuda")
    t0 = th.randn(b, c, h, w, device="cuda") if per_pixel else th.randn(b, c, 1, 1, device="cuda")
    bias0 = th.randn(c, device="cuda")
    x, t, bias = (v.clone().requires_grad_() for v in (y0, t0, bias0))
    pre = (x.view(b, s, c, h, w) + t.unsqueeze(1) + bias.view(1, 1, -1, 1, 1)).view(b * s, c, h, w)
    ref = pre if act == 0 else th.nn.functional.leaky_relu(pre, slope if act == 2 else 0.0)
    g = th.randn_like(ref)
    ref.backward(g)
    x2, t2, b2 = (v.clone().requires_grad_() for v in (y0, t0, bias0))
    inp = x2 * 1.0
    assert F.CtxAct.supported(inp, t2, s)
    out = F.CtxAct.apply(inp, t2, b2, s, act, slope)
    out.backward(g)
    close(out, ref, rtol=1e-6)
    close(x2.grad, x.grad, rtol=1e-6)
    close(t2.grad, t.grad, rtol=1e-5)
    close(b2.grad, bias.grad, rtol=1e-5)


def test_pointwise_chain_with_context_matches_concatenation():
    from sbmc_amd import modules
    th.manual_seed(20)
    for ctx_shape in ((2, 5, 12, 20), (2, 3, 1, 1)):
        cp = ctx_shape[1]
        chain = modules.ConvChain(7 + cp, 9, ksize=1, width=16, depth=3, pad=False,
                                  activation="leaky_relu").cuda()
        per_sample = th.randn(2, 4, 7, 12, 20, device="cuda")
        ctx = th.randn(*ctx_shape, device="cuda")
        pa, ca = per_sample.clone().requires_grad_(), ctx.clone().requires_grad_()
        pb, cb = per_sample.clone().requires_grad_(), ctx.clone().requires_grad_()
        cat = th.cat([pa, ca.expand(2, cp, 12, 20).unsqueeze(1).expand(2, 4, cp, 12, 20)], 2)
        ya = chain(cat.reshape(8, 7 + cp, 12, 20))
        chain.pointwise_as_gemm = True
        yb = modules.pointwise_chain_with_context(chain, pb, cb)
        assert yb is not None
        close(yb, ya, rtol=1e-5)
        g = th.randn_like(ya)
        ga = th.autograd.grad(ya, [pa, ca] + list(chain.parameters()), g)
        gb = th.autograd.grad(yb, [pb, cb] + list(chain.parameters()), g)
        for a, b_ in zip(ga, gb):
            close(b_, a, rtol=3e-5)


def test_half_logits_fall_back_to_fp32_operators_where_the_strip_kernels_do_not_apply():
    """k = 5 with half logits: no fp16 kernel exists, the module up-casts and composes the fp32 ops."""
    from sbmc_amd import modules
    th.manual_seed(21)
    d = th.rand(1, 3, 12, 70, device="cuda")
    kh = th.randn(1, 25, 12, 70, device="cuda").half().requires_grad_()
    kf = kh.detach().float().requires_grad_()
    a = modules.ProgressiveKernelApply(splat=True)(d, kh, None, None, None)
    b = modules.ProgressiveKernelApply(splat=True)(d, kf, None, None, None)
    for x, y in zip(a, b):
        close(x, y)
    a[0].sum().backward()
    b[0].sum().backward()
    assert kh.grad.dtype == th.float16
    close(kh.grad.float(), kf.grad, rtol=1e-3)
    o1, s1 = modules.KernelApply(softmax=True, splat=True)(d, kh.detach())
    o2, s2 = modules.KernelApply(softmax=True, splat=True)(d, kf.detach())
    close(o1, o2); close(s1, s2)


@pytest.mark.parametrize("bs,c,h,w,k,spp", [
    (1, 3, 16, 16, 3, 3), (2, 3, 19, 70, 5, 2), (1, 3, 40, 150, 21, 3), (1, 3, 5, 7, 21, 2),
    (1, 4, 9, 130, 21, 2), (1, 3, 33, 65, 9, 2)])
def test_fused_gather_update_vs_oracle(oracle, bs, c, h, w, k, spp):
    """ProgressiveKernelApply(splat=False) -- pixel-centred (gather) kernels, the reference's
    `--gather` ablation -- on the fused gather kernels vs the oracle composition, forward and
    backward with upstream gradients on all three outputs."""
    from sbmc_amd import functions as F, modules
    th.manual_seed(22)
    datas = [th.rand(bs, c, h, w) * 2 for _ in range(spp)]
    kerns = [th.randn(bs, k * k, h, w) * 2 for _ in range(spp)]
    grads = [th.randn(bs, c, h, w), th.randn(bs, 1, h, w), th.randn(bs, 1, h, w)]
    assert F.gather_update_supported(datas[0].cuda(), kerns[0].cuda())
    ref_out, ref_dd, ref_dk = _progressive(
        lambda d, kk, a, b, m: oracle.progressive_kernel_apply(d, kk, a, b, m, splat=False),
        datas, kerns, grads, "cpu")
    out, dd, dk = _progressive(modules.ProgressiveKernelApply(splat=False), datas, kerns, grads, "cuda")
    state_close(out, ref_out, truth=lambda: progressive_fp64(datas, kerns, splat=False)[0])
    # d_kernels: the element that receives the routed max-gradient is a difference of two k*k-term sums
    # in BOTH fp32 implementations; it is held to 1e-5 of the float64 restatement of the same graph, or to
    # twice the oracle's own fp32 error against it, whichever is larger (helpers.no_worse_than)
    _, _, dk64 = progressive_fp64(datas, kerns, grads, splat=False)
    for s in range(spp):
        close(dd[s], ref_dd[s], what="d_data[%d]" % s)
        no_worse_than(dk[s], ref_dk[s], dk64[s], what="d_kernels[%d]" % s)


def test_fused_gather_running_max_branches(oracle):
    from sbmc_amd import modules
    th.manual_seed(23)
    bs, c, h, w, k = 1, 3, 12, 70, 5
    base = th.randn(bs, k * k, h, w)
    spike = base.clone()
    spike[:, 7] += 30.0
    datas = [th.rand(bs, c, h, w) for _ in range(4)]
    kerns = [base, spike, base - 50.0, spike.clone()]
    grads = [th.randn(bs, c, h, w), th.randn(bs, 1, h, w), th.randn(bs, 1, h, w)]
    ref_out, ref_dd, ref_dk = _progressive(
        lambda d, kk, a, b, m: oracle.progressive_kernel_apply(d, kk, a, b, m, splat=False),
        datas, kerns, grads, "cpu")
    out, dd, dk = _progressive(modules.ProgressiveKernelApply(splat=False), datas, kerns, grads, "cuda")
    for a, b in zip(out, ref_out):
        close(a, b)
    _, _, dk64 = progressive_fp64(datas, kerns, grads, splat=False)
    for s in range(4):
        close(dd[s], ref_dd[s]); no_worse_than(dk[s], ref_dk[s], dk64[s], what="d_kernels[%d]" % s)


@pytest.mark.parametrize("shape", [(1, 3, 2, 1, 2), (2, 5, 3, 7, 10), (1, 16, 8, 45, 64), (3, 4, 0, 6, 4), (1, 2, 2, 1, 6)])
def test_upsample_cat_kernels(shape):
    """Bilinear x2 upsampling + concatenation in one pass (csrc/resample.hip) vs
    F.interpolate + th.cat, forward and backward (gather adjoint vs PyTorch's atomics)."""
    from sbmc_amd import functions as F
    import torch.nn.functional as nnf
    b, cu, cl, h, w = shape
    th.manual_seed(sum(shape))
    c0 = th.randn(b, cu, h, w, device="cuda")
    l0 = th.randn(b, cl, 2 * h, 2 * w, device="cuda")
    ca, la = c0.clone().requires_grad_(), l0.clone().requires_grad_()
    ref = th.cat([nnf.interpolate(ca, scale_factor=2, mode="bilinear", align_corners=False), la], 1)
    g = th.randn_like(ref)
    ref.backward(g)
    cb, lb = c0.clone().requires_grad_(), l0.clone().requires_grad_()
    assert F.upsample_cat_supported(cb, lb)
    out = F.UpsampleCat.apply(cb, lb)
    out.backward(g)
    close(out, ref, rtol=1e-6)
    close(cb.grad, ca.grad, rtol=1e-5)
    if cl:
        assert th.equal(lb.grad, la.grad)


def test_fused_layers_refuse_half_tensors():
    """The ctypes wrappers hand raw float pointers to the C ABI: a half tensor (autocast) must raise,
    not be read out of bounds; under autocast the chains take the library path instead."""
    from sbmc_amd import functions as F, modules
    x = th.randn(2, 16, 64, device="cuda")
    w = th.randn(8, 16, device="cuda")
    b = th.randn(8, device="cuda")
    with pytest.raises(TypeError):
        F.PointwiseLayer.apply(x, w, b, th.randn(2, 8, 64, device="cuda").half(), 1, 1, 0.0)
    with pytest.raises(TypeError):
        F.PointwiseLayer.apply(x.half(), w, b, None, 1, 1, 0.0)
    with pytest.raises(TypeError):
        F.BiasAct.apply(th.randn(2, 8, 64, device="cuda").half(), b, 1, 0.0)
    with th.autocast("cuda", dtype=th.float16):
        assert not F.pointwise_supported(x, 8)
        chain = modules.ConvChain(7 + 5, 9, ksize=1, width=16, depth=3, pad=False).cuda()
        chain.pointwise_as_gemm = True
        per_sample = th.randn(2, 4, 7, 12, 20, device="cuda")
        ctx = th.randn(2, 5, 12, 20, device="cuda")
        out = modules.pointwise_chain_with_context(chain, per_sample, ctx)
        ref_in = th.cat([per_sample, ctx.unsqueeze(1).expand(2, 4, 5, 12, 20)], 2).reshape(8, 12, 12, 20)
        chain.pointwise_as_gemm = False
        ref = chain(ref_in)
        if out is not None:
            assert (out.float() - ref.float()).abs().max().item() <= 3e-2 * ref.float().abs().max().item()


@pytest.mark.parametrize("x_half", [False, True])
@pytest.mark.parametrize("cfg", [(4, 2, 93, 128, 260, 2, 1), (2, 1, 128, 441, 132, 0, 0), (8, 8, 128, 128, 1028, 1, 2)])
def test_pointwise_half_storage(cfg, x_half):
    """Half-storage forward of the fused 1x1 layer (fp16 in HBM, fp32 arithmetic) vs the fp32 layer on
    the same (half-rounded) inputs: equal up to the rounding of the fp16 output."""
    from sbmc_amd import functions as F
    B, S, cin, cout, hw, tm, act = cfg
    slope = 0.01 if act == 2 else 0.0
    th.manual_seed(sum(cfg))
    x = th.randn(B, cin, hw, device="cuda")
    if x_half:
        x = x.half()
    w = th.randn(cout, cin, device="cuda") / cin ** 0.5
    b = th.randn(cout, device="cuda")
    t = None if tm == 0 else (th.randn(B // S, cout, device="cuda") if tm == 1 else th.randn(B // S, cout, hw, device="cuda"))
    # (half in / half out runs on the f16 matrix pipe with the weights rounded to half)
    ref = F.PointwiseLayer.apply(x.float(), w.half().float() if x_half else w, b, t, S, act, slope)
    with th.no_grad(), th.autocast("cuda", dtype=th.float16):
        assert F.pointwise_half_supported(x, cout)
        y = F.pointwise_half(x, w, b, t, S, act, slope)
    assert y.dtype == th.float16
    err = (y.float() - ref).abs().max().item()
    assert err <= 1e-3 * ref.abs().max().item() + 1e-3, err


@pytest.mark.parametrize("bs,c,h,w,k", SHAPES)
def test_float16_boundary_operators_vs_oracle(oracle, bs, c, h, w, k):
    """`*_cuda_float16`: the three boundary operators on torch.float16 tensors vs the
    oracle on the same half-rounded inputs.  Scatter2Gather is a permutation (bit exact); the sums are
    formed in fp32 and rounded once on the store, so they agree with the rounded fp32 result to one half ulp
    (2^-11 relative) plus the fp32 bound."""
    from sbmc_amd import functions as F, halide_ops
    th.manual_seed(40 + k)
    x = th.randn(bs, k, k, h, w).half()
    out = F.Scatter2Gather.apply(x.cuda())
    assert out.dtype == th.float16
    assert th.equal(out.cpu(), oracle.Scatter2Gather.apply(x.float()).half())

    data = (th.rand(bs, c, h, w) * 2).half()
    wts = th.randn(bs, k, k, h, w).half()
    g_out, g_sw = th.randn(bs, c, h, w).half(), th.randn(bs, h, w).half()
    dr, wr = data.float().requires_grad_(), wts.float().requires_grad_()
    ro, rs = oracle.KernelWeighting.apply(dr, wr)
    th.autograd.backward([ro, rs], [g_out.float(), g_sw.float()])
    dg, wg = data.cuda().requires_grad_(), wts.cuda().requires_grad_()
    o, s = F.KernelWeighting.apply(dg, wg)
    assert o.dtype == th.float16 and s.dtype == th.float16
    th.autograd.backward([o, s], [g_out.cuda(), g_sw.cuda()])

    def half_close(a, b, what, partial_roundings=0):
        a, b = a.detach().cpu().double(), b.detach().double()
        bound = 2.0 ** -10 * b.abs() + (1e-5 + partial_roundings * 2.0 ** -10) * b.abs().max().item() + 1e-7
        assert ((a - b).abs() <= bound).all(), "%s: max err %.3e" % (what, (a - b).abs().max().item())
    half_close(o, ro, "output"); half_close(s, rs, "sum_w")
    half_close(dg.grad, dr.grad, "d_data")
    # more than 8 channels go in groups of 8: d_weights is then accumulated THROUGH its half storage, one
    # more rounding per extra group -- of a partial sum, i.e. relative to the tensor's scale, not the element's
    half_close(wg.grad, wr.grad, "d_weights", partial_roundings=0 if c <= 8 else 2)
    # the C-ABI shim refuses mixed dtypes instead of reading out of bounds
    with pytest.raises(RuntimeError):
        halide_ops.kernel_weighting_cuda_float16(data.cuda(), wts.cuda().float(), o.detach(), s.detach())
    with pytest.raises(RuntimeError):
        halide_ops.scatter2gather_cuda_float32(x.cuda(), th.empty_like(x.cuda()))


@pytest.mark.parametrize("cfg", [
    # B, S, cin, cout, hw, t_mode, act, x_half, with_mean
    (4, 2, 93, 128, 260, 1, 1, False, False),     # a chain's first layer: fp32 features in, per-image context
    (8, 4, 128, 128, 1028, 2, 2, True, False),    # per-pixel context, leaky relu
    (6, 3, 128, 96, 516, 0, 1, True, True),       # last embedding layer: also returns the mean over samples
    (4, 2, 64, 128, 68, 1, 2, True, False),       # narrow input (one column block of gw), ragged last tile
    (3, 1, 96, 40, 132, 0, 1, True, False),       # ragged channel counts on both sides
    (2, 1, 128, 441, 132, 0, 0, True, False),     # the logits layer (wider than the fused backward: GEMM path)
])
def test_pointwise_half_training(cfg):
    """Half-storage forward AND backward of the fused 1x1 layer (training under torch.autocast(float16),
    SURVEY.md row N4) vs fp32 torch on exactly the values the kernel sees (half-rounded x / gy, the sign of
    the half y): y and gx to one half rounding, the fp32 outputs (gw, gbias, gt) to the fp32 bound."""
    from sbmc_amd import functions as F
    B, S, cin, cout, hw, tm, act, x_half, with_mean = cfg
    slope = 0.01 if act == 2 else 0.0
    th.manual_seed(sum(int(v) for v in cfg))
    x = th.randn(B, cin, hw, device="cuda")
    x = x.half() if x_half else x
    w = (th.randn(cout, cin, device="cuda") / cin ** 0.5).requires_grad_()
    b = th.randn(cout, device="cuda").requires_grad_()
    t = None if tm == 0 else (th.randn(B // S, cout, device="cuda") if tm == 1 else th.randn(B // S, cout, hw, device="cuda"))
    if t is not None:
        t.requires_grad_()
    xg = x.clone().requires_grad_()
    if with_mean:
        y, ym = F.PointwiseLayerMean.apply(xg, w, b, t, S, act, slope, S, True)
    else:
        y = F.PointwiseLayer.apply(xg, w, b, t, S, act, slope, True)
    assert y.dtype == th.float16
    gy = th.randn(B, cout, hw, device="cuda").half()
    gm = th.randn(B // S, cout, hw, device="cuda").half() if with_mean else None
    th.autograd.backward([y, ym] if with_mean else [y], [gy, gm] if with_mean else [gy])

    # fp32 reference on the same values
    xr = x.float()
    # half in / half out runs on the f16 matrix pipe: the weights are rounded to half there (forward only)
    wq = w.detach().half().float() if x_half else w.detach()
    pre = th.einsum("oc,bcp->bop", wq, xr) + b.detach().view(1, -1, 1)
    if tm == 1:
        pre = pre + t.detach().repeat_interleave(S, 0).unsqueeze(-1)
    elif tm == 2:
        pre = pre + t.detach().repeat_interleave(S, 0)
    yr = pre if act == 0 else th.where(pre > 0, pre, pre * slope)
    assert (y.float() - yr).abs().max().item() <= 2.0 ** -10 * yr.abs().max().item() + 1e-3
    g = gy.float()
    if with_mean:
        g = g + gm.float().repeat_interleave(S, 0) / S
    gz = g if act == 0 else th.where(y > 0, g, g * slope)
    # an all-half layer's backward runs on the f16 matrix pipe too: gz and the weights enter its products rounded
    # to half (exact products, fp32 sums); the side sums (gbias, gt) are taken before that rounding
    f16_pipe = x_half and cout <= 128
    gzq = gz.half().float() if f16_pipe else gz
    # (the sums against FLOAT64: torch's fp32 einsum over 1e4 terms is ten times further from it than the kernel's fp32
    # accumulators are -- tools/dev/half_gw_check.py, profiles/HISTORY.md)
    if cout <= 128:
        # (+ a few terms whose gz rounded to half the other way: gz is formed in fp32 from gy (+ gm / S) on either side)
        close_sum(w.grad, th.einsum("bop,bcp->oc", gzq.double(), xr.double()),
                  th.einsum("bop,bcp->oc", gzq.double().abs(), xr.double().abs()), what="gw",
                  extra=(8 * 2.0 ** -11 * gzq.abs().max() * xr.abs().max()).item() if f16_pipe else 0.0)
    else:
        close(w.grad, th.einsum("bop,bcp->oc", gzq.double(), xr.double()), rtol=2e-3, what="gw")
    close_sum(b.grad, gz.double().sum((0, 2)), gz.double().abs().sum((0, 2)), what="gbias")
    gxr = th.einsum("oc,bop->bcp", wq if f16_pipe else w.detach(), gzq)
    if cout <= 128:
        assert xg.grad.dtype == x.dtype
        tol = 2.0 ** -10 if x_half else 1e-5
        assert (xg.grad.float() - gxr).abs().max().item() <= tol * gxr.abs().max().item() + 1e-6
    else:                                            # half GEMM operands (w rounded to half)
        assert (xg.grad.float() - gxr).abs().max().item() <= 4e-3 * gxr.abs().max().item()
    if tm == 1:
        close_sum(t.grad, gz.double().view(B // S, S, cout, hw).sum((1, 3)), gz.double().abs().view(B // S, S, cout, hw).sum((1, 3)),
                  what="gt (per image)")
    elif tm == 2:
        close(t.grad, gz.view(B // S, S, cout, hw).sum(1), rtol=1e-5, what="gt (per pixel)")


def test_multisteps_trains_under_fp16_autocast_on_the_fused_kernels():
    """A training step under torch.autocast(float16): the per-sample chains stay on the fused MFMA kernels
    (half storage, forward and backward), the splat takes half logits; gradients agree with the fp32 step to
    fp16 accuracy."""
    from sbmc_amd import Multisteps, functions as F, losses
    from sbmc_amd.utils import crop_like
    th.manual_seed(33)
    model = Multisteps(12, 3, width=32, embedding_width=32, ksize=21, nsteps=2).cuda().train()
    g = th.Generator().manual_seed(34)
    batch = {"radiance": th.empty(1, 3, 3, 48, 72).exponential_(1.0, generator=g).cuda(),
             "features": th.rand(1, 3, 12, 48, 72, generator=g).cuda(),
             "global_features": th.rand(1, 3, 1, 1, generator=g).cuda()}
    tgt = th.empty(1, 3, 48, 72).exponential_(1.0, generator=g).cuda()
    loss_fn = losses.TonemappedRelativeMSE()
    out = model(batch)["radiance"]
    loss32 = loss_fn(out, crop_like(tgt, out))
    loss32.backward()
    ref = {k: q.grad.clone() for k, q in model.named_parameters()}
    model.zero_grad()
    store = []
    F.enable_kernel_timing(store)
    try:
        with th.autocast("cuda", dtype=th.float16):
            out = model(batch)["radiance"]
            loss16 = loss_fn(out.float(), crop_like(tgt, out))
        scale = 65536.0                        # static loss scaling, as torch.amp.GradScaler would apply:
        (loss16 * scale).backward()            # the raw gradients (1e-6 .. 1e-12 here) underflow in fp16
    finally:
        F.enable_kernel_timing(None)
    names = {n.split(" ")[0] for n, _, _ in store}
    assert "pointwise_fwd_f16" in names and "pointwise_bwd_f16" in names, names       # the fused half kernels ran
    assert "splat_update_fwd_all_f16" in names and "splat_update_bwd_all_f16" in names, names
    assert abs(loss16.item() - loss32.item()) <= 2e-2 * abs(loss32.item())
    got = {k: q.grad / scale for k, q in model.named_parameters()}
    for k, v in got.items():
        assert th.isfinite(v).all(), k
    flat_a = th.cat([got[k].reshape(-1) for k in ref]).double()
    flat_b = th.cat([ref[k].reshape(-1) for k in ref]).double()
    cos = (flat_a @ flat_b / (flat_a.norm() * flat_b.norm())).item()
    assert cos >= 0.995, cos                                   # the step direction is the fp32 one
    top = max(v.abs().max().item() for v in ref.values())
    for k in ref:                                              # every tensor that carries a significant gradient
        d = ref[k].abs().max().item()
        if d >= 1e-2 * top:
            assert (got[k] - ref[k]).abs().max().item() <= 0.1 * d, k
