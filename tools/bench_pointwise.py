"""Dev tool: fused pointwise layer (csrc/pointwise.hip) vs bmm + BiasAct, correctness and time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from sbmc_amd import _lib, functions as funcs

dev = th.device("cuda")
L = _lib.lib()


def fused(x, w, bias, t, s, t_mode, act, slope):
    b, cin, hw = x.shape
    cout = w.shape[0]
    y = th.empty(b, cout, hw, device=dev)
    rc = L.sbmc_pointwise_fwd_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(t) if t is not None else None,
                                  _lib.ptr(y), b, s, cin, cout, hw, t_mode, act, slope, _lib.current_stream(dev))
    _lib.check(rc, "pointwise_fwd")
    return y


def ref(x, w, bias, t, s, t_mode, act, slope):
    y = th.matmul(w.double(), x.double()) + bias.double().view(1, -1, 1)
    if t_mode == 1:
        y = y + t.double().repeat_interleave(s, 0).unsqueeze(-1)
    elif t_mode == 2:
        y = y + t.double().repeat_interleave(s, 0)
    if act == 1:
        y = y.clamp(min=0)
    elif act == 2:
        y = th.where(y > 0, y, y * slope)
    return y


def check(b, s, cin, cout, hw, t_mode, act):
    th.manual_seed(b * 1000 + cin + cout + hw)
    x = th.randn(b, cin, hw, device=dev)
    w = th.randn(cout, cin, device=dev) / cin ** 0.5
    bias = th.randn(cout, device=dev)
    t = None
    if t_mode == 1:
        t = th.randn(b // s, cout, device=dev)
    elif t_mode == 2:
        t = th.randn(b // s, cout, hw, device=dev)
    y = fused(x, w, bias, t, s, t_mode, act, 0.01)
    r = ref(x, w, bias, t, s, t_mode, act, 0.01)
    err = (y.double() - r).abs().max().item()
    print("b%d s%d cin%d cout%d hw%d t%d act%d: max err %.2e" % (b, s, cin, cout, hw, t_mode, act, err), flush=True)
    assert err < 2e-5, err


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    th.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


if __name__ == "__main__":
    for args in [(2, 1, 128, 128, 1024, 0, 1), (3, 1, 93, 128, 1000, 0, 2), (4, 2, 128, 128, 260, 2, 1),
                 (4, 2, 93, 128, 516, 1, 1), (2, 1, 128, 441, 388, 0, 0), (2, 2, 128, 441, 132, 2, 2),
                 (1, 1, 7, 5, 4, 0, 1), (2, 1, 33, 200, 36, 0, 0)]:
        if "--notest" not in sys.argv:
            check(*args)
    if "--time" in sys.argv:
        hw = 1280 * 720
        for cin, cout, act in [(128, 128, 1), (93, 128, 1), (128, 441, 0)]:
            x = th.randn(8, cin, hw, device=dev)
            w = th.randn(cout, cin, device=dev) / cin ** 0.5
            bias = th.randn(cout, device=dev)
            tf = timeit(lambda: fused(x, w, bias, None, 1, 0, act, 0.0))

            def lib():
                y = th.bmm(w.unsqueeze(0).expand(8, -1, -1), x)
                return funcs.BiasAct.apply(y, bias, act, 0.0)
            tl = timeit(lib)
            fl = 2.0 * cin * cout * 8 * hw
            by = 4.0 * 8 * hw * (cin + cout)
            print("cin %d cout %d: fused %.3f ms (%.1f TFLOP/s, %.2f TB/s) | bmm+BiasAct %.3f ms" % (
                cin, cout, tf, fl / tf / 1e9, by / tf / 1e9, tl), flush=True)
            del x

if "--bwd" in sys.argv:
    hw = 1280 * 720
    for cin, cout, act, needx in [(128, 128, 1, True), (93, 128, 1, False)]:
        x = th.randn(8, cin, hw, device=dev, requires_grad=needx)
        w = (th.randn(cout, cin, device=dev) / cin ** 0.5).requires_grad_()
        bias = th.randn(cout, device=dev, requires_grad=True)
        y = funcs.PointwiseLayer.apply(x, w, bias, None, 1, act, 0.0)
        g = th.randn_like(y)
        tb = timeit(lambda: th.autograd.grad(y, [w, bias] + ([x] if needx else []), g, retain_graph=True))
        fl = 2.0 * cin * cout * 8 * hw * (2 if needx else 1)
        by = 4.0 * 8 * hw * (cin + 2 * cout + (cin if needx else 0))
        print("bwd cin %d cout %d dx=%s: %.3f ms (%.1f TFLOP/s, %.2f TB/s)" % (cin, cout, needx, tb, fl / tb / 1e9, by / tb / 1e9), flush=True)
        del x, y, g

    x = th.randn(8, 128, hw, device=dev, requires_grad=True)
    w = (th.randn(128, 128, device=dev) / 128 ** 0.5).requires_grad_()
    bias = th.randn(128, device=dev, requires_grad=True)
    t = th.randn(1, 128, hw, device=dev, requires_grad=True)
    y = funcs.PointwiseLayer.apply(x, w, bias, t, 8, 1, 0.0)
    g = th.randn_like(y)
    tb = timeit(lambda: th.autograd.grad(y, [w, bias, x, t], g, retain_graph=True))
    print("bwd 128x128 per-pixel context, 8 samples: %.3f ms" % tb, flush=True)
