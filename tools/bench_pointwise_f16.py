"""Dev tool: half-storage forward of the fused 1x1 layer, f16 MFMA (weights in half) vs fp32 MFMA, at 720p."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from sbmc_amd import functions as F

dev = th.device("cuda")
hw = 1280 * 720
for (B, S, cin, cout, tm) in ((8, 8, 128, 128, 0), (8, 8, 128, 128, 2), (8, 8, 128, 441, 0), (32, 32, 128, 128, 0)):
    th.manual_seed(0)
    x = th.randn(B, cin, hw, device=dev).half()
    w = th.randn(cout, cin, device=dev) / cin ** 0.5
    b = th.randn(cout, device=dev)
    t = th.randn(B // S, cout, hw, device=dev) if tm == 2 else None
    res = {}
    for knob in ("1", "0"):
        os.environ["SBMC_HIP_PW_F16MFMA"] = knob
        with th.no_grad(), th.autocast("cuda", dtype=th.float16):
            for _ in range(3):
                y = F.pointwise_half(x, w, b, t, S, 2, 0.01)
            th.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                y = F.pointwise_half(x, w, b, t, S, 2, 0.01)
            th.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        res[knob] = (ms, y.float())
        gb = (B * cin * hw * 2 + B * cout * hw * 2 + (t.numel() * 4 if t is not None else 0)) / 1e9
        print("B%d %d->%d t_mode %d  %s: %.3f ms  %.0f GB/s  %.1f TFLOP/s" % (
            B, cin, cout, tm, "f16 MFMA" if knob == "1" else "f32 MFMA", ms, gb / ms * 1e3,
            2.0 * cin * cout * B * hw / ms / 1e9), flush=True)
    d = (res["1"][1] - res["0"][1]).abs().max().item()
    print("   max |f16-mfma - f32-mfma| = %.3e (max |y| %.2f)" % (d, res["0"][1].abs().max().item()), flush=True)
    del x, y, res
    th.cuda.empty_cache()

# backward of the all-half layer: f16 MFMA (pw_bwd_h_kernel) vs fp32 MFMA (pw_bwd_kernel<.., half, half>)
for (B, S, cin, cout, tm, act) in ((8, 8, 128, 128, 0, 2), (8, 8, 128, 128, 2, 2), (8, 8, 64, 128, 0, 1)):
    th.manual_seed(0)
    x = th.randn(B, cin, hw, device=dev).half().requires_grad_()
    w = (th.randn(cout, cin, device=dev) / cin ** 0.5).requires_grad_()
    b = th.randn(cout, device=dev).requires_grad_()
    t = th.randn(B // S, cout, hw, device=dev).requires_grad_() if tm == 2 else None
    gy = th.randn(B, cout, hw, device=dev).half()
    y = F.PointwiseLayer.apply(x, w, b, t, S, act, 0.01, True)
    res = {}
    for knob in ("1", "0"):
        os.environ["SBMC_HIP_PW_F16MFMA"] = knob
        for _ in range(2):
            x.grad = w.grad = b.grad = None
            y.backward(gy, retain_graph=True)
        th.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            x.grad = w.grad = b.grad = None
            y.backward(gy, retain_graph=True)
        th.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        res[knob] = (ms, x.grad.float(), w.grad.clone())
        gb = (2 * B * cin * hw * 2 + 2 * B * cout * hw * 2) / 1e9
        print("bwd B%d %d->%d t_mode %d  %s: %.3f ms  %.0f GB/s  %.1f TFLOP/s" % (
            B, cin, cout, tm, "f16 MFMA" if knob == "1" else "f32 MFMA", ms, gb / ms * 1e3,
            4.0 * cin * cout * B * hw / ms / 1e9), flush=True)
    print("   max |dgx| = %.3e (max %.2f)   max |dgw| = %.3e (max %.2f)" % (
        (res["1"][1] - res["0"][1]).abs().max().item(), res["0"][1].abs().max().item(),
        (res["1"][2] - res["0"][2]).abs().max().item(), res["0"][2].abs().max().item()), flush=True)
    del x, y, res, gy
    th.cuda.empty_cache()
