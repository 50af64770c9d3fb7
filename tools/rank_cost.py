"""Per-rank cost of the sharded training step on ONE GPU (no communication: neighbour halos are
replaced by zeros): an upper bound on the strong-scaling speed-up, T(1 GPU) / T(one rank of N)."""
import os, sys, time
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
import bench
from sbmc_amd import Multisteps, losses
from sbmc_amd import dist as sdist

def fake_exchange(part, to_up, to_down):
    return (th.zeros_like(to_up) if part.has_up else None, th.zeros_like(to_down) if part.has_down else None)

IPC_SELF = "--ipc-self" in sys.argv
if IPC_SELF:
    # neighbour rows through the IPC mailboxes with this very rank as both neighbours (halo.HaloChannel in
    # loop-back: every put / get / merge kernel, flag and slot of the multi-GPU step runs, only the xGMI
    # transfer itself is local); the gradient all-reduce over a single-rank RCCL group as with --rccl-self
    sys.argv.remove("--ipc-self")
    sys.argv.append("--rccl-self")
RCCL_SELF = "--rccl-self" in sys.argv
if RCCL_SELF:
    # the exchanges for real, over RCCL, with this very rank as both neighbours (a single-rank `nccl` group:
    # send/recv to self, one-rank all-reduce): every launch, copy and stream dependency of the multi-GPU step is
    # there, only the xGMI transfer itself (<= 6 rows per message, 139 MB once) is not
    sys.argv.remove("--rccl-self")
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    th.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=th.device("cuda", 0))
    sdist.SlabPartition.peer = lambda self, delta: 0
else:
    def fake_exchange_into(part, to_up, to_down, into_up, into_down, between=None):
        if part.has_up:
            into_up.zero_()
        if part.has_down:
            into_down.zero_()
        if between is not None:
            between()
    sdist._exchange = fake_exchange
    sdist._exchange_into = fake_exchange_into
    sdist._all_reduce_sum = lambda t, part: t.cuda() if not t.is_cuda else t
    sdist._all_reduce_sum_start = lambda t, part: None
    sdist._all_reduce_min = lambda t, part: t

dev = th.device("cuda")
H, W, S, K = 720, 1280, 8, 21
if "--4k" in sys.argv:                      # BASELINE configs[3]: 3840x2160 (one rank of 8 fits one GPU)
    sys.argv.remove("--4k")
    H, W = 2160, 3840
if "--quarter" in sys.argv:                 # 640x360: the GPU's share shrinks 4x, the host's does not
    sys.argv.remove("--quarter")
    H, W = 368, 640
HOST_TIME = "--host-time" in sys.argv
if HOST_TIME:
    sys.argv.remove("--host-time")
full = bench.make_model_inputs(H, W, S, dev, seed=1234)
for world in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    th.manual_seed(0)
    model = Multisteps(93, 3, ksize=K).to(dev).train()
    opt = th.optim.Adam(model.parameters(), lr=1e-4, fused=True)   # as bench.py
    loss_fn = losses.TonemappedRelativeMSE()
    # the rank that bounds the step: most rows, and among those an interior one (halos on both sides)
    parts = [sdist.SlabPartition(H, world, r) for r in range(world)]
    rank = max(range(world), key=lambda r: (parts[r].rows, parts[r].has_up and parts[r].has_down))
    if "SBMC_RANK" in os.environ:
        rank = int(os.environ["SBMC_RANK"])
    part = parts[rank]
    batch = {k: (v if k == "global_features" else v[..., part.y0:part.y1, :].contiguous()) for k, v in full.items()}
    runner = sdist.ShardedDenoiser(model, part)
    runner._channel_tried = True                 # (no other ranks to connect to: the transport is chosen here)
    if IPC_SELF and world > 1:
        if not (part.has_up and part.has_down):
            raise SystemExit("--ipc-self pairs what goes up with what comes from below: interior ranks only")
        from sbmc_amd.halo import HaloChannel
        part.channel = HaloChannel(dev, 64 << 20, 4).loopback()
    if world == 1:
        step = lambda: bench.train_step(model, opt, loss_fn, batch)
    else:
        step = lambda: runner.train_step(opt, loss_fn, batch)
    for _ in range(2):
        step()
    # (--host-time: how long the HOST takes to enqueue a step, up to its one synchronisation -- `th.stack` of the status
    # and the loss in ShardedDenoiser.train_step: a step cannot be shorter than that)
    host = []
    if HOST_TIME and world > 1:
        orig_stack = th.stack
        begun = [0.0]

        def stack_mark(*a, **k):
            host.append(time.time() - begun[0])
            return orig_stack(*a, **k)
        th.stack = stack_mark
        inner = step
        step = lambda: (begun.__setitem__(0, time.time()), inner())[1]
    if "SBMC_RANK_COST_CPROFILE" in os.environ:       # where the host's time goes: cProfile of 3 steps
        import cProfile, pstats
        th.cuda.synchronize()
        prof = cProfile.Profile()
        prof.enable()
        for _ in range(3):
            step()
        th.cuda.synchronize()
        prof.disable()
        st = pstats.Stats(prof, stream=open(os.environ["SBMC_RANK_COST_CPROFILE"], "w"))
        st.sort_stats("tottime").print_stats(70)
        st.sort_stats("cumulative").print_stats(90)
    th.cuda.synchronize(); t0 = time.time()
    for _ in range(3):
        step()
    th.cuda.synchronize()
    ms = (time.time() - t0) / 3 * 1e3
    if host:
        th.stack = orig_stack
        # (two calls per step: the step's own, then one inside clip_grad_norm_ -- after the synchronisation)
        print("host enqueue up to the step's synchronisation: %s ms" % ", ".join("%.1f" % (1e3 * h) for h in host[-6::2]), flush=True)
    print("world %d rank %d rows %d%s: %.1f ms/step" % (world, rank, part.rows,
                                                         " (exchanges through IPC mailboxes to self)" if IPC_SELF else " (exchanges over RCCL to self)" if RCCL_SELF else "", ms), flush=True)
    del model, opt, runner, batch
    th.cuda.empty_cache()
