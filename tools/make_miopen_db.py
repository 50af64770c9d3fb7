"""Produces the MIOpen find records shipped in sbmc_amd/miopen_db/ (run on an MI355X through gpurun):

    MIOPEN_FIND_MODE=1 MIOPEN_USER_DB_PATH=<dir> python tools/make_miopen_db.py [--layout nhwc|nchw] [--ranks N [--rank R]] [--4k] [--fp16]

One training step (forward + backward: all three convolution directions) of Multisteps(93,3) at 1280x720 with
MIOpen's full find, so that MIOpen writes what it measured for every convolution configuration of the U-nets
into <dir>/<gpu>.<miopen build>.ufdb.txt.  --ranks N: the slab of one interior rank of N (the shapes the
sharded path sees).  Copy the resulting file into sbmc_amd/miopen_db/ (records of several runs accumulate).
"""
import os, sys, time
assert os.environ.get("MIOPEN_FIND_MODE") == "1" and os.environ.get("MIOPEN_USER_DB_PATH"), __doc__
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
layout = sys.argv[sys.argv.index("--layout") + 1] if "--layout" in sys.argv else "nhwc"
os.environ["SBMC_UNET_LAYOUT"] = layout
import torch as th
import bench
from sbmc_amd import Multisteps, losses
from sbmc_amd import dist as sdist

ranks = int(sys.argv[sys.argv.index("--ranks") + 1]) if "--ranks" in sys.argv else 1
rank = int(sys.argv[sys.argv.index("--rank") + 1]) if "--rank" in sys.argv else ranks // 2
H, W = (2160, 3840) if "--4k" in sys.argv else (720, 1280)
dev = th.device("cuda")
th.manual_seed(0)
model = Multisteps(93, 3, ksize=21).to(dev).train()
opt = th.optim.Adam(model.parameters(), lr=1e-4, fused=True)
loss_fn = losses.TonemappedRelativeMSE()
t0 = time.time()
fp16 = "--fp16" in sys.argv          # records for the half convolutions of torch.autocast(float16)
if ranks == 1:
    batch = bench.make_model_inputs(H, W, 8, dev, seed=1)
    bench.train_step(model, opt, loss_fn, batch, fp16=fp16)
else:
    sdist._exchange = lambda part, a, b: (th.zeros_like(a) if part.has_up else None, th.zeros_like(b) if part.has_down else None)

    def _into(part, to_up, to_down, into_up, into_down, between=None):
        if part.has_up:
            into_up.zero_()
        if part.has_down:
            into_down.zero_()
        if between is not None:
            between()
    sdist._exchange_into = _into
    sdist._all_reduce_sum = lambda t, part: t.cuda() if not t.is_cuda else t
    sdist._all_reduce_sum_start = lambda t, part: None
    sdist._all_reduce_min = lambda t, part: t
    part = sdist.SlabPartition(H, ranks, rank)
    batch = bench.make_model_inputs(H, W, 8, dev, seed=1, rows=(part.y0, part.y1))
    sdist.ShardedDenoiser(model, part).train_step(opt, loss_fn, batch)
th.cuda.synchronize()
print("find + one step (%s, rank %d of %d, %dx%d): %.0f s" % (layout, rank, ranks, W, H, time.time() - t0), flush=True)
for f in os.listdir(os.environ["MIOPEN_USER_DB_PATH"]):
    p = os.path.join(os.environ["MIOPEN_USER_DB_PATH"], f)
    print(f, os.path.getsize(p), "bytes", sum(1 for _ in open(p)) if f.endswith(".txt") else "")
