"""fuzz_pointwise's failing shape (B2 S2 cin1 cout1 hw5472, half layer): the weight gradient's distance from FLOAT64 over 40 draws,
for whichever library SBMC_HIP_LIB names.  The fuzz holds it to 3e-5 |ref| + 1e-6 sqrt(N) of an fp32 einsum."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch as th
from sbmc_amd import functions as F
dev = "cuda"
worst = {}
for shape in ((2, 2, 1, 1, 5472), (2, 2, 3, 5, 5472), (4, 2, 128, 128, 4096), (2, 1, 7, 1, 20000)):
    B, S, cin, cout, hw = shape
    for x_half in (False, True):
        errs, refs = [], []
        for seed in range(20):
            g = th.Generator(device="cpu").manual_seed(seed)
            x0 = th.randn(B, cin, hw, generator=g).to(dev)
            w0 = th.randn(cout, cin, generator=g).to(dev)
            b0 = th.randn(cout, generator=g).to(dev)
            gy = th.randn(B, cout, hw, generator=g).to(dev).half()
            x = x0.half() if x_half else x0
            w, b = w0.clone().requires_grad_(), b0.clone().requires_grad_()
            xg = x.clone().requires_grad_()
            y = F.PointwiseLayer.apply(xg, w, b, None, S, 0, 1.0, True)
            y.backward(gy)
            gz = gy.float()
            ref = th.einsum("bop,bcp->oc", gz.double(), x.double())
            f32 = th.einsum("bop,bcp->oc", gz, x.float()).double()
            errs.append(((w.grad.double() - ref).abs().max().item(), (f32 - ref).abs().max().item()))
        n = (B * hw) ** 0.5
        print("B%d S%d %d->%d hw%d x_half=%d: ours vs float64 max %.3e (median %.3e), torch fp32 einsum vs float64 max %.3e; 1e-6 sqrt(N) = %.3e" % (
            B, S, cin, cout, hw, x_half, max(e[0] for e in errs), sorted(e[0] for e in errs)[10], max(e[1] for e in errs), 1e-6 * n), flush=True)
