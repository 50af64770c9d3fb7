"""Dev aid: where do the gradients of a training step under autocast(float16) differ from fp32?"""
import os, sys
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
from sbmc_amd import Multisteps, losses
from sbmc_amd.utils import crop_like

th.manual_seed(33)
model = Multisteps(12, 3, width=32, embedding_width=32, ksize=21, nsteps=2).cuda().train()
g = th.Generator().manual_seed(34)
batch = {"radiance": th.empty(1, 3, 3, 48, 72).exponential_(1.0, generator=g).cuda(),
         "features": th.rand(1, 3, 12, 48, 72, generator=g).cuda(),
         "global_features": th.rand(1, 3, 1, 1, generator=g).cuda()}
tgt = th.empty(1, 3, 48, 72).exponential_(1.0, generator=g).cuda()
loss_fn = losses.TonemappedRelativeMSE()


def grads(fp16, scale=1.0, gemm=True):
    model.zero_grad()
    for m in model.modules():
        if hasattr(m, "pointwise_as_gemm") and m.__class__.__name__ == "ConvChain":
            pass
    model.kernel_regressor.pointwise_as_gemm = gemm
    for s in range(model.nsteps):
        getattr(model, "embedding_%02d" % s).pointwise_as_gemm = gemm
    with th.autocast("cuda", dtype=th.float16, enabled=fp16):
        out = model(batch)["radiance"]
    loss = loss_fn(out.float(), crop_like(tgt, out))
    (loss * scale).backward()
    return loss.item(), {k: q.grad.clone() / scale for k, q in model.named_parameters()}


l32, ref = grads(False)
for name, kw in (("fp16 fused", dict(fp16=True)), ("fp16 fused, loss x 4096", dict(fp16=True, scale=4096.0)),
                 ("fp16 library path (MIOpen 1x1), loss x 4096", dict(fp16=True, scale=4096.0, gemm=False))):
    l, gr = grads(**kw)
    print("==", name, "loss", l, "vs", l32)
    rows = []
    for k in ref:
        d = ref[k].abs().max().item()
        e = (gr[k] - ref[k]).abs().max().item()
        rows.append((e / d if d > 0 else 0.0, k, d))
    rows.sort(reverse=True)
    for r in rows[:8]:
        print("   %.3e  %-55s max|ref| %.3e" % r)
