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
opt = th.optim.Adam(model.parameters(), lr=1e-4)
loss_fn = losses.TonemappedRelativeMSE()
batch = bench.make_model_inputs(720, 1280, 8, dev, seed=1234)
def step():
    bench.train_step(model, opt, loss_fn, batch)
for i in range(2):
    t0 = time.time(); step(); th.cuda.synchronize(); print("warm", i, time.time() - t0, flush=True)
t0 = time.time()
for i in range(3):
    step()
th.cuda.synchronize()
print("ms/step", (time.time() - t0) / 3 * 1e3, "cfg", sys.argv[1:], "mem GB", th.cuda.max_memory_allocated() / 1e9, flush=True)
if "--prof" in sys.argv:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step(); th.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60))
