"""Backbone experiments (not part of the product): timing of the Multisteps training step under
different conv configurations.  usage: tools_model_exp.py [--cl] [--bench] [--gemm1x1] [--prof]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
import bench
from sbmc_amd import Multisteps, losses

cl = "--cl" in sys.argv
th.backends.cudnn.benchmark = "--bench" in sys.argv
dev = th.device("cuda")
th.manual_seed(0)
model = Multisteps(93, 3, ksize=21, pointwise_gemm="--nogemm1x1" not in sys.argv).to(dev)
model.train()
if "--nofuseconv" in sys.argv:
    from sbmc_amd import modules
    for m in model.modules():
        if isinstance(m, modules.ConvChain):
            m.fuse_bias_act = False
opt = th.optim.Adam(model.parameters(), lr=1e-4, fused="--fusedadam" in sys.argv)
loss_fn = losses.TonemappedRelativeMSE()
batch = bench.make_model_inputs(720, 1280, 8, dev, seed=1234)
def step():
    bench.train_step(model, opt, loss_fn, batch)
for i in range(2):
    t0 = time.time(); step(); th.cuda.synchronize(); print("warm", i, time.time() - t0, flush=True)
import gc
if "--gcfreeze" in sys.argv:
    gc.collect(); gc.freeze()
if "--gcoff" in sys.argv:
    gc.collect(); gc.disable()
_g0 = [g["collections"] for g in gc.get_stats()]
t0 = time.time()
for i in range(3):
    step()
th.cuda.synchronize()
print("ms/step", (time.time() - t0) / 3 * 1e3, "cfg", sys.argv[1:], "gc", [b["collections"] - a for a, b in zip(_g0, gc.get_stats())], flush=True)
if "--graph" in sys.argv:
    from sbmc_amd.utils import crop_like
    opt = th.optim.Adam(model.parameters(), lr=1e-4, capturable=True)
    def gstep():
        opt.zero_grad(set_to_none=False)
        out = model(batch)["radiance"]
        loss = loss_fn(out, crop_like(batch["target_image"], out))
        loss.backward()
        th.nn.utils.clip_grad_norm_(model.parameters(), 1000)
        opt.step()
        return loss
    s_ = th.cuda.Stream()
    s_.wait_stream(th.cuda.current_stream())
    with th.cuda.stream(s_):
        for i in range(2):
            gstep()
    th.cuda.current_stream().wait_stream(s_)
    th.cuda.synchronize(); th.cuda.empty_cache()
    g = th.cuda.CUDAGraph()
    with th.cuda.graph(g):
        static_loss = gstep()
    th.cuda.synchronize()
    for i in range(2):
        g.replay()
    th.cuda.synchronize()
    t0 = time.time()
    for i in range(3):
        g.replay()
    th.cuda.synchronize()
    print("graph ms/step", (time.time() - t0) / 3 * 1e3, "loss", static_loss.item(), "mem GB", th.cuda.max_memory_allocated() / 1e9, flush=True)
if "--prof" in sys.argv:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step(); th.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60))
