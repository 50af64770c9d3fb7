"""Per-launch times of the 1x1 layers at 1280x720 x 8 spp through the C ABI: the two-f16-plane form (magnitude words)
beside the three-bf16-plane form, forward and backward variants of the training step.
    python tools/bench_pw_scaled.py [--hw N]"""
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbmc_amd import _lib  # noqa: E402

dev = th.device("cuda")
L = _lib.lib()
HW = 1280 * 720
if "--hw" in sys.argv:
    HW = int(sys.argv[sys.argv.index("--hw") + 1])
B, S = 8, 8


def word(t):
    return t.abs().max().reshape(1).view(th.int32).clone()


def timeit(fn, n=8):
    for _ in range(3):
        fn()
    th.cuda.synchronize()
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    th.cuda.synchronize()
    return a.elapsed_time(b) / n


def fwd(cin, cout, t_mode, act, mean, scaled):
    x = th.randn(B, cin, HW, device=dev)
    w = th.randn(cout, cin, device=dev) / cin ** 0.5
    bias = th.randn(cout, device=dev)
    t = th.randn(1, cout, HW, device=dev) if t_mode == 2 else (th.randn(1, cout, device=dev) if t_mode == 1 else None)
    y = th.empty(B, cout, HW, device=dev)
    signs = th.empty(B, cout, (HW + 31) // 32, dtype=th.int32, device=dev) if act else None
    ym = th.empty(1, cout, HW, device=dev) if mean else None
    xm = word(x) if scaled else None
    am = th.zeros(1, dtype=th.int32, device=dev)

    def run():
        _lib.check(L.sbmc_pointwise_fwd_scaled_f32(
            _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(t) if t is not None else None, _lib.ptr(y),
            _lib.ptr(signs) if signs is not None else None, _lib.ptr(ym) if mean else None, S if mean else 1,
            _lib.ptr(xm) if scaled else None, _lib.ptr(am), B, S, cin, cout, HW, t_mode, act, 0.01,
            _lib.current_stream(dev)), "fwd")
    ms = timeit(run)
    by = 4.0 * B * HW * (cin + cout) + (4.0 * HW * cout if t_mode == 2 else 0) + (4.0 * HW * cout if mean else 0)
    return ms, by


def bwd(cin, cout, t_mode, act, gm, dx, scaled):
    x = th.randn(B, cin, HW, device=dev)
    w = th.randn(cout, cin, device=dev) / cin ** 0.5
    gy = th.randn(B, cout, HW, device=dev)
    signs = th.randint(-2 ** 31, 2 ** 31 - 1, (B, cout, (HW + 31) // 32), dtype=th.int32, device=dev)
    gmean = th.randn(1, cout, HW, device=dev) if gm else None
    groups = L.sbmc_pointwise_bwd_groups(B, S, 1 if (gm and t_mode == 0) else t_mode, HW)
    gx = th.empty(B, cin, HW, device=dev) if dx else None
    gwp = th.empty(groups, cout, cin, device=dev)
    gbp = th.empty(groups, 1, cout, device=dev)
    gt = th.empty(1, cout, HW, device=dev) if t_mode == 2 else None
    words = (word(gy), word(gmean) if gm else None, word(x)) if scaled else (None, None, None)
    gxm = th.zeros(1, dtype=th.int32, device=dev) if (dx and (scaled or not (gm or t_mode == 2))) else None

    def run():
        _lib.check(L.sbmc_pointwise_bwd_scaled_f32(
            _lib.ptr(gy), _lib.ptr(signs) if act else None, _lib.ptr(x), _lib.ptr(w), _lib.ptr(gx) if dx else None,
            _lib.ptr(gwp), _lib.ptr(gbp), _lib.ptr(gt) if gt is not None else None, _lib.ptr(gmean) if gm else None, S,
            _lib.ptr(words[0]) if scaled else None, _lib.ptr(words[1]) if words[1] is not None else None,
            _lib.ptr(words[2]) if scaled else None, _lib.ptr(gxm) if gxm is not None else None,
            B, S, cin, cout, HW, t_mode, act, 0.01, _lib.current_stream(dev)), "bwd")
    ms = timeit(run)
    by = 4.0 * B * HW * (cin + cout + (cin if dx else 0)) + (4.0 * HW * cout if (t_mode == 2 or gm) else 0)
    return ms, by


def wide_bwd():
    cin, cout = 128, 441
    gz = th.randn(B, cout, HW, device=dev) * 1e-3
    x = th.randn(B, cin, HW, device=dev)
    w = th.randn(cout, cin, device=dev) / cin ** 0.5
    groups = L.sbmc_pointwise_gw_wide_groups(B, HW)
    gwp = th.empty(groups, cout, cin, device=dev)
    gbp = th.empty(groups, cout, device=dev)
    gx = th.empty(B, cin, HW, device=dev)
    ws = th.empty(L.sbmc_pointwise_wide_bwd_ws_bytes(), dtype=th.uint8, device=dev)
    gm, xm, gxm = word(gz), word(x), th.zeros(1, dtype=th.int32, device=dev)

    def fused():
        _lib.check(L.sbmc_pointwise_wide_bwd_f32(_lib.ptr(gz), _lib.ptr(x), _lib.ptr(w), _lib.ptr(gx), _lib.ptr(gwp), _lib.ptr(gbp),
                                                 _lib.ptr(ws), _lib.ptr(gm), _lib.ptr(xm), _lib.ptr(gxm), B, cin, cout, HW,
                                                 _lib.current_stream(dev)), "wide")

    def two_pass():
        _lib.check(L.sbmc_pointwise_gw_wide_f32(_lib.ptr(gz), _lib.ptr(x), _lib.ptr(gwp), _lib.ptr(gbp), B, cin, cout, HW,
                                                _lib.current_stream(dev)), "gw_wide")
        return th.bmm(w.t().unsqueeze(0).expand(B, -1, -1), gz)
    by = 4.0 * B * HW * (cout + 2 * cin)
    a, b_ = timeit(fused), timeit(two_pass)
    print("%-40s one pass %.3f ms %.2f TB/s | wide gw kernel + library GEMM %.3f ms" % ("bwd 128->441 (gx, gw, gbias)", a, by / a / 1e9, b_), flush=True)


if __name__ == "__main__":
    if "--wide" in sys.argv:
        wide_bwd()
        sys.exit(0)
    rows = [("fwd 128->128 relu", lambda sc: fwd(128, 128, 0, 1, False, sc)),
            ("fwd 128->128 linear + mean", lambda sc: fwd(128, 128, 0, 0, True, sc)),
            ("fwd 128->128 per-pixel context", lambda sc: fwd(128, 128, 2, 1, False, sc)),
            ("fwd 96->128 per-image context", lambda sc: fwd(96, 128, 1, 1, False, sc)),
            ("fwd 128->441 linear", lambda sc: fwd(128, 441, 0, 0, False, sc)),
            ("bwd 128->128 relu", lambda sc: bwd(128, 128, 0, 1, False, True, sc)),
            ("bwd 128->128 linear + mean gradient", lambda sc: bwd(128, 128, 0, 0, True, True, sc)),
            ("bwd 128->128 per-pixel context", lambda sc: bwd(128, 128, 2, 1, False, True, sc)),
            ("bwd 96->128 no data gradient", lambda sc: bwd(96, 128, 1, 1, False, False, sc))]
    for name, fn in rows:
        out = []
        for sc in (True, False):
            ms, by = fn(sc)
            out.append("%s %.3f ms %.2f TB/s" % ("two f16 planes" if sc else "three bf16 planes", ms, by / ms / 1e9))
            th.cuda.empty_cache()
        print("%-40s %s | %s" % (name, out[0], out[1]), flush=True)
    wide_bwd()
