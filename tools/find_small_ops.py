"""Who launches the step's small torch kernels?  One training step of an interior rank of 8 (tools/rank_cost.py
--ipc-self 8) -- or the single-GPU step with `1` -- under torch.profiler with Python stacks; prints, for the aten
operators given (default: the fills, copies and reductions), how often each Python call site issues them.
    python tools/find_small_ops.py [1|8] [aten::fill_ aten::zero_ ...]"""
import collections, os, sys
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch as th
from torch.profiler import profile, ProfilerActivity

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ops = sys.argv[2:] or ["aten::fill_", "aten::zero_", "aten::copy_", "aten::sum", "aten::add_", "aten::add", "aten::mul",
                       "aten::clone", "aten::contiguous", "aten::maximum", "aten::cat", "aten::constant_pad_nd"]
import bench
from sbmc_amd import Multisteps, losses, dist as sdist
dev = th.device("cuda")
H, W, S, K = 720, 1280, 8, 21
full = bench.make_model_inputs(H, W, S, dev, seed=1234)
th.manual_seed(0)
model = Multisteps(93, 3, ksize=K).to(dev).train()
opt = th.optim.Adam(model.parameters(), lr=1e-4, fused=True)
loss_fn = losses.TonemappedRelativeMSE()
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29543")
    th.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=th.device("cuda", 0))
    sdist.SlabPartition.peer = lambda self, delta: 0
    part = sdist.SlabPartition(H, world, 1)
    batch = {k: (v if k == "global_features" else v[..., part.y0:part.y1, :].contiguous()) for k, v in full.items()}
    runner = sdist.ShardedDenoiser(model, part)
    runner._channel_tried = True
    from sbmc_amd.halo import HaloChannel
    part.channel = HaloChannel(dev, 64 << 20, 4).loopback()
    step = lambda: runner.train_step(opt, loss_fn, batch)
else:
    step = lambda: bench.train_step(model, opt, loss_fn, full)
for _ in range(2):
    step()
th.cuda.synchronize()
if "SBMC_SMALL_KERNELS" in os.environ:
    # which operator (and Python frame) launches the step's fill / memset / copy KERNELS -- what the dispatch-mode
    # listing below cannot see: composite C++ operators (constant_pad_nd = fill + copy), memsets inside libraries
    pat = tuple(os.environ["SBMC_SMALL_KERNELS"].split(",")) if os.environ["SBMC_SMALL_KERNELS"] else (
        "fill", "Fill", "copyBuffer", "copy_kernel", "Memset", "Memcpy")
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        th.cuda.synchronize()
    by = collections.Counter()
    dur = collections.Counter()
    for ev in prof.events():
        for k in getattr(ev, "kernels", []) or []:
            if any(q in k.name for q in pat):
                frames = [f for f in (ev.stack or []) if "sbmc_amd/" in f or "bench.py" in f or "clip_grad" in f or "/optim/" in f]
                where = frames[0].split("/")[-1] if frames else "(no python frame)"
                key = (k.name[:48], ev.name[:40], where[:70])
                by[key] += 1
                dur[key] += k.duration
    print("world %d: small kernels of one step by launching operator" % world)
    for key, n in sorted(by.items(), key=lambda kv: -kv[1])[:80]:
        print("%5d %8.1f us  %-48s %-40s %s" % (n, dur[key], key[0], key[1], key[2]))
    sys.exit(0)
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
sites = collections.Counter()
WATCH = ("zero_", "fill_", "zeros", "sum", "copy_", "constant_pad_nd", "add_", "add", "mul", "clone", "maximum", "cat",
         "zeros_like", "new_zeros", "full", "_to_copy", "div", "sub", "neg", "where", "clamp", "index_put_", "sqrt", "reciprocal")


class Watch(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WATCH:
            dev = None
            for a in list(args) + list((kwargs or {}).values()):
                if isinstance(a, th.Tensor):
                    dev = a.device.type
                    break
                if isinstance(a, (list, tuple)) and a and isinstance(a[0], th.Tensor):
                    dev = a[0].device.type
                    break
            if dev is None:
                dev = str((kwargs or {}).get("device", "?"))
            if "cuda" in str(dev):
                frames = [f for f in traceback.extract_stack() if "/sbmc_amd/" in f.filename or f.filename.endswith("bench.py")
                          or "clip_grad" in f.filename or "/optim/" in f.filename]
                where = "%s:%d %s" % (os.path.relpath(frames[-1].filename, ROOT), frames[-1].lineno, frames[-1].name) if frames else "(C++)"
                node = th._C._current_autograd_node()
                sites[(name, where, type(node).__name__ if node is not None else "-")] += 1
        return func(*args, **(kwargs or {}))


with Watch():
    step()
    th.cuda.synchronize()
print("world %d: small aten operators on GPU tensors by call site and autograd node (one step)" % world)
for (name, where, node), n in sorted(sites.items(), key=lambda kv: -kv[1])[:120]:
    print("%5d  %-16s %-60s %s" % (n, name, where, node))
