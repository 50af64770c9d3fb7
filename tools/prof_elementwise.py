import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch as th
import bench
from sbmc_amd import Multisteps, losses
dev = th.device("cuda")
th.manual_seed(0)
model = Multisteps(93, 3, ksize=21).to(dev).train()
opt = th.optim.Adam(model.parameters(), lr=1e-4, fused=True)
loss_fn = losses.TonemappedRelativeMSE()
batch = bench.make_model_inputs(720, 1280, 8, dev, seed=1234)
for i in range(2):
    bench.train_step(model, opt, loss_fn, batch)
th.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    bench.train_step(model, opt, loss_fn, batch); th.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::copy_", "aten::cat", "aten::add_", "aten::add", "aten::sum", "aten::mean", "aten::mul", "aten::threshold_backward", "aten::clamp_min", "aten::relu", "aten::contiguous", "aten::clone", "aten::expand", "aten::div"):
        rows.append((e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total, e.key, e.count, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
for r in rows[:28]:
    print("%9.2f ms  %-26s n=%-3d %s" % (r[0] / 1e3, r[1], r[2], r[3]))
