"""Row-slab form of the splat (one frame sharded along H) on the GPU vs the CPU oracle.

A frame's samples are cut into row slabs; every slab is splatted on its own with
`functions.SplatAll(..., top, bot, zero_top, zero_bot)` (the `sbmc_splat_slab_*` entry points),
the overhang rows are handed to the neighbouring slab exactly as `dist.merge_overhang` does over
RCCL (here: in one process), and the merged state -- values and all gradients -- must equal the
oracle's whole-frame `progressive_kernel_apply` chain (the reference composition,
sbmc/modules.py:376-473) within the 1e-5 bound.
"""
import pytest
import torch as th

from helpers import no_worse_than, progressive_fp64, state_close

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-5, what=""):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    assert a.shape == b.shape, what
    scale = b.abs().max().item() if b.numel() else 0.0
    err = (a - b).abs()
    bad = err > rtol * scale + rtol * b.abs()
    assert not bad.any(), "%s: max err %.3e (scale %.3e), %d bad" % (what, err.max().item(), scale, int(bad.sum()))


def sharded_state(rad, kern, bounds, p):
    """rad [1,S,c,H,W], kern [1,S,k*k,H,W] on the GPU; bounds: [(y0, y1), ...] covering [0, H).
    Returns the merged (sum_r, sum_w, max_w) on the whole frame, built slab by slab."""
    from sbmc_amd import functions as F
    from sbmc_amd import dist as sdist
    c = rad.shape[2]
    n = len(bounds)
    ext = []
    for r, (y0, y1) in enumerate(bounds):
        up, down = r > 0, r < n - 1
        st = F.SplatAll.apply(rad[..., y0:y1, :].contiguous(), kern[..., y0:y1, :].contiguous(),
                              p if up else 0, p if down else 0, not up, not down)
        ext.append(th.cat(st, 1))
    rows = []
    for r, (y0, y1) in enumerate(bounds):
        up, down = r > 0, r < n - 1
        top, bot = (p if up else 0), (p if down else 0)
        own = ext[r][..., top:ext[r].shape[-2] - bot, :]
        if up:      # the neighbour above splatted into its bottom overhang = my first p rows
            own = sdist._merge_rows(own, ext[r - 1][..., ext[r - 1].shape[-2] - p:, :], 0, p, c)
        if down:
            own = sdist._merge_rows(own, ext[r + 1][..., :p, :], own.shape[-2] - p, own.shape[-2], c)
        rows.append(own)
    full = th.cat(rows, -2)
    return full[:, :c], full[:, c:c + 1], full[:, c + 1:]


CASES = [
    # k, H, W, S, bounds
    (5, 24, 70, 2, [(0, 12), (12, 24)]),
    (5, 30, 130, 3, [(0, 8), (8, 20), (20, 30)]),
    (21, 48, 150, 2, [(0, 24), (24, 48)]),
    (21, 64, 72, 2, [(0, 20), (20, 44), (44, 64)]),       # interior slab with two neighbours
    (21, 40, 200, 1, [(0, 10), (10, 20), (20, 30), (30, 40)]),   # slabs exactly as thin as the kernel radius
    (9, 33, 65, 2, [(0, 17), (17, 33)]),                  # ragged rows / columns
]


@pytest.mark.parametrize("k,H,W,S,bounds", CASES)
def test_slab_splat_merged_equals_oracle_whole_frame(oracle, k, H, W, S, bounds):
    p = (k - 1) // 2
    g = th.Generator().manual_seed(11 + k + H)
    rad = th.empty(1, S, 3, H, W).exponential_(1.0, generator=g)
    kern = th.randn(1, S, k * k, H, W, generator=g) * 2.0
    gr, gw, gm = (th.randn(1, 3, H, W, generator=g), th.randn(1, 1, H, W, generator=g),
                  th.randn(1, 1, H, W, generator=g) * 0.1)

    # oracle: the reference composition on the whole frame
    ro = rad.clone().requires_grad_()
    ko = kern.clone().requires_grad_()
    st = (None, None, None)
    for s in range(S):
        st = oracle.progressive_kernel_apply(ro[:, s], ko[:, s], *st, splat=True)
    th.autograd.backward(list(st), [gr, gw, gm])

    rg = rad.cuda().requires_grad_()
    kg = kern.cuda().requires_grad_()
    out = sharded_state(rg, kg, bounds, p)
    th.autograd.backward(list(out), [gr.cuda(), gw.cuda(), gm.cuda()])
    state_close(out, st, truth=lambda: progressive_fp64([rad[:, s] for s in range(S)], [kern[:, s] for s in range(S)])[0])
    close(rg.grad, ro.grad, what="d_radiance")
    # the routed arg-max element of d_kernels: 1e-5 of the float64 restatement, or the oracle's own fp32 error
    _, _, dk64 = progressive_fp64([rad[:, s] for s in range(S)], [kern[:, s] for s in range(S)], [gr, gw, gm])
    no_worse_than(kg.grad, ko.grad, th.stack(dk64, 1), what="d_kernels")


def test_slab_splat_normalised_output_equals_whole_frame_gpu():
    """sum_r / (sum_w + eps) of the sharded splat == the whole-frame SplatAll on the same GPU (values and
    gradients), at a frame wide enough for interior and border strips, k = 21."""
    from sbmc_amd import functions as F
    k, H, W, S = 21, 96, 300, 3
    p = 10
    g = th.Generator().manual_seed(5)
    rad = th.empty(1, S, 3, H, W).exponential_(1.0, generator=g).cuda()
    kern = (th.randn(1, S, k * k, H, W, generator=g) * 3).cuda()
    d_out = th.randn(1, 3, H, W, generator=g).cuda()

    r1, k1 = rad.clone().requires_grad_(), kern.clone().requires_grad_()
    sr, sw, _ = F.SplatAll.apply(r1, k1)
    o1 = sr / (sw + 1e-8)
    o1.backward(d_out)
    r2, k2 = rad.clone().requires_grad_(), kern.clone().requires_grad_()
    sr, sw, _ = sharded_state(r2, k2, [(0, 32), (32, 64), (64, 96)], p)
    o2 = sr / (sw + 1e-8)
    o2.backward(d_out)
    close(o2, o1, what="normalised output")
    close(r2.grad, r1.grad, what="d_radiance")
    close(k2.grad, k1.grad, what="d_kernels")


def test_slab_entry_points_validate():
    from sbmc_amd import _lib
    L = _lib.lib()
    assert L.sbmc_splat_slab_supported(3, 21, 90, 1280, 10, 10) == 1
    assert L.sbmc_splat_slab_supported(3, 21, 90, 1280, 11, 0) == 0       # overhang beyond the kernel radius
    assert L.sbmc_splat_slab_supported(5, 5, 90, 1280, 2, 2) == 0         # no strip kernel for c = 5 at k = 5
    assert L.sbmc_splat_slab_fwd_f32(*([None] * 6), 1, 3, 8, 8, 21, 11, 0, 0, 1, None) == -1
    assert L.sbmc_splat_slab_bwd_f32(*([None] * 13), 1, 1, 3, 8, 8, 21, 0, 12, None) == -1
    assert L.sbmc_splat_slab_fwd_f32(*([None] * 6), 0, 3, 8, 8, 21, 10, 10, 0, 0, None) == 0   # empty batch


def test_half_logit_slab(oracle):
    """fp16 logit storage through the slab entry points (k = 21): values vs the oracle on the same
    half-rounded logits."""
    k, H, W, S, p = 21, 40, 130, 2, 10
    g = th.Generator().manual_seed(8)
    rad = th.empty(1, S, 3, H, W).exponential_(1.0, generator=g)
    kern = th.randn(1, S, k * k, H, W, generator=g).half()
    st = (None, None, None)
    for s in range(S):
        st = oracle.progressive_kernel_apply(rad[:, s], kern[:, s].float(), *st, splat=True)
    out = sharded_state(rad.cuda(), kern.cuda(), [(0, 20), (20, 40)], p)
    state_close(out, st, truth=lambda: progressive_fp64([rad[:, s] for s in range(S)], [kern[:, s].float() for s in range(S)])[0])


@pytest.mark.parametrize("top,bot", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("shape", [(1, 5, 3, 6, 10), (2, 16, 8, 11, 64)])
def test_upsample_cat_row_slab(shape, top, bot):
    """UpsampleCat in its row-slab form (coarse map carrying the neighbours' edge rows) vs
    F.interpolate on the padded map + crop + cat -- what dist._level did before -- forward and backward."""
    import torch.nn.functional as nnf
    from sbmc_amd import functions as F
    b, cu, cl, h, w = shape
    hc = h + top + bot
    th.manual_seed(sum(shape) + top + 2 * bot)
    c0 = th.randn(b, cu, hc, w, device="cuda")
    l0 = th.randn(b, cl, 2 * h, 2 * w, device="cuda")
    ca, la = c0.clone().requires_grad_(), l0.clone().requires_grad_()
    up = nnf.interpolate(ca, scale_factor=2, mode="bilinear", align_corners=False)
    ref = th.cat([up[..., 2 * top:2 * hc - 2 * bot, :], la], 1)
    g = th.randn_like(ref)
    ref.backward(g)
    cb, lb = c0.clone().requires_grad_(), l0.clone().requires_grad_()
    assert F.upsample_cat_supported(cb, lb, top, bot)
    out = F.UpsampleCat.apply(cb, lb, top, bot)
    out.backward(g)
    close(out, ref, rtol=1e-6, what="output")
    close(cb.grad, ca.grad, rtol=1e-5, what="d_coarse")
    assert th.equal(lb.grad, la.grad)


@pytest.mark.parametrize("top,bot", [(1, 1), (1, 0), (0, 1)])
@pytest.mark.parametrize("shape", [(1, 8, 4, 6, 10), (1, 16, 8, 11, 64)])
def test_upsample_cat_row_slab_channels_last(shape, top, bot):
    """The channels-last form of the same (the U-nets run NHWC when that measures faster)."""
    import torch.nn.functional as nnf
    from sbmc_amd import functions as F
    b, cu, cl, h, w = shape
    hc = h + top + bot
    th.manual_seed(sum(shape) + top + 2 * bot)
    c0 = th.randn(b, cu, hc, w, device="cuda").contiguous(memory_format=th.channels_last)
    l0 = th.randn(b, cl, 2 * h, 2 * w, device="cuda").contiguous(memory_format=th.channels_last)
    ca, la = c0.clone().requires_grad_(), l0.clone().requires_grad_()
    up = nnf.interpolate(ca, scale_factor=2, mode="bilinear", align_corners=False)
    ref = th.cat([up[..., 2 * top:2 * hc - 2 * bot, :], la], 1)
    g = th.randn_like(ref)
    ref.backward(g)
    cb, lb = c0.clone().requires_grad_(), l0.clone().requires_grad_()
    assert F.upsample_cat_nhwc_supported(cb, lb, top, bot)
    out = F.UpsampleCatNHWC.apply(cb, lb, top, bot)
    out.backward(g)
    close(out, ref, rtol=1e-6, what="output")
    close(cb.grad, ca.grad, rtol=1e-5, what="d_coarse")
    close(lb.grad, la.grad, rtol=0, what="d_left")
