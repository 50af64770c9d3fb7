"""Error growth through a chain of 3 x 3 convolutions, with and without a leaky ReLU between them: the split-precision
kernel vs the fp32 library vs float64 (the activation's sign flips dominate the gradients of either fp32 path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th, torch.nn.functional as F
from sbmc_amd import functions as funcs
dev = th.device("cuda"); th.manual_seed(0)
def cl(t): return t.contiguous(memory_format=th.channels_last)
def err(a, r): return (a.double() - r).abs().max().item() / r.abs().max().item()
L = 8
for (h, wd) in ((48, 80), (180, 320)):
    x = th.randn(1, 128, h, wd, device=dev)
    ws = [th.randn(128, 128, 3, 3, device=dev) * (1.0 / (128 * 9) ** 0.5) for _ in range(L)]
    gy = th.randn(1, 128, h, wd, device=dev)
    for act in (False, True):
        def run(kind):
            dt = th.float64 if kind == "ref" else th.float32
            xi = (cl(x) if kind == "ours" else x).to(dt).requires_grad_(True)
            wl = [w.to(dt).requires_grad_(True) for w in ws]
            t = xi
            for w in wl:
                t = funcs.Conv3x3NHWC.apply(t, w) if kind == "ours" else F.conv2d(t, w, padding=1)
                if act: t = F.leaky_relu(t, 0.01)
            gs = th.autograd.grad(t, [xi] + wl, (cl(gy) if kind == "ours" else gy).to(dt))
            return [t.detach()] + list(gs)
        ref, ours, lib = run("ref"), run("ours"), run("lib")
        print("%dx%d act=%d  y: ours %.2e lib %.2e | gx: %.2e %.2e" % (h, wd, act, err(ours[0], ref[0]), err(lib[0], ref[0]), err(ours[1], ref[1]), err(lib[1], ref[1])))
        print("    gw by layer ours: " + " ".join("%.1e" % err(ours[2 + i], ref[2 + i]) for i in range(L)))
        print("    gw by layer lib : " + " ".join("%.1e" % err(lib[2 + i], ref[2 + i]) for i in range(L)))
