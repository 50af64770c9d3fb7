"""Per-launch times of the fused 1x1 chains (csrc/pointwise_chain.hip) at 1280x720 x 8 spp beside the layer-by-layer
kernels they replace, forward only: the embeddings' three layers (training form: intermediates + sign words written;
inference form: nothing but the last layer) and the regressor's first two.
    python tools/bench_pw_chain.py [--hw N] [--spp S]"""
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbmc_amd import _lib, functions as funcs  # noqa: E402

dev = th.device("cuda")
HW = 1280 * 720
S = 8
if "--hw" in sys.argv:
    HW = int(sys.argv[sys.argv.index("--hw") + 1])
if "--spp" in sys.argv:
    S = int(sys.argv[sys.argv.index("--spp") + 1])
B = S


def timeit(fn, n=6):
    for _ in range(2):
        fn()
    th.cuda.synchronize()
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    th.cuda.synchronize()
    return a.elapsed_time(b) / n


def case(name, cin, couts, acts, t_mode, mean):
    th.manual_seed(0)
    x = th.randn(B, cin, HW, device=dev)
    layers, k = [], cin
    for c, a in zip(couts, acts):
        layers.append((th.randn(c, k, device=dev) / k ** 0.5, th.randn(c, device=dev) * 0.1, a, 0.01))
        k = c
    t = th.randn(1, couts[0], HW, device=dev) if t_mode == 2 else (th.randn(1, couts[0], device=dev) if t_mode == 1 else None)
    for train in (True, False):
        ms = timeit(lambda: funcs.pointwise_chain_forward(x, t, S, layers, store_mid=train, want_signs=train, mean=mean))
        nbytes = 4.0 * B * HW * (cin + (sum(couts) if train else couts[-1])) + (4.0 * HW * couts[0] if t_mode == 2 else 0)

        def separate():
            cur = x
            with th.set_grad_enabled(train):
                if train:
                    cur = cur.detach().requires_grad_(True)     # (the separate layers then write their sign words too)
                for l, (w, bias, act, slope) in enumerate(layers):
                    last = l + 1 == len(layers)
                    if last and mean:
                        cur, _ = funcs.PointwiseLayerMean.apply(cur, w, bias, None, 1, act, slope, S)
                    else:
                        cur = funcs.PointwiseLayer.apply(cur, w, bias, t if l == 0 else None, S if l == 0 else 1, act, slope)
            return cur
        funcs.ensure_amax(x)
        ms0 = timeit(separate)
        print("%-34s %-9s fused %6.3f ms (%5.2f TB/s of its bytes) | layer by layer %6.3f ms | x%.2f"
              % (name, "training" if train else "inference", ms, nbytes / ms / 1e9, ms0, ms0 / ms), flush=True)
    del x, layers, t
    th.cuda.empty_cache()


if __name__ == "__main__":
    _lib.lib()
    case("embedding 128+ctx->128->128->128", 128, (128, 128, 128), (1, 1, 0), 2, True)
    case("embedding_00 93+gf->128->128->128", 93, (128, 128, 128), (1, 1, 0), 1, True)
    case("regressor 128+ctx->128->128", 128, (128, 128), (2, 2), 2, False)
    if os.environ.get("SBMC_PC_TIMING"):
        # development: cycles per phase of wave 0 of workgroup 0 (the kernel leaves them in the output's first words)
        th.manual_seed(0)
        x = th.randn(B, 128, HW, device=dev)
        layers = [(th.randn(128, 128, device=dev) / 128 ** 0.5, th.randn(128, device=dev) * 0.1, a, 0.01) for a in (1, 1, 0)]
        t = th.randn(1, 128, HW, device=dev)
        for train in (True, False):
            ys = funcs.pointwise_chain_forward(x, t, S, layers, store_mid=train, want_signs=train, mean=True)[0]
            th.cuda.synchronize()
            v = ys[-1].view(-1)[:4].tolist()
            print("training" if train else "inference", "tiles %d; cycles per tile by phase (layer 0, 1, 2): %s; sum %.0f"
                  % (v[0], " ".join("%.0f" % (c / v[0]) for c in v[1:]), sum(v[1:]) / v[0]))
