#!/usr/bin/env python
"""Benchmark of the SBMC hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload model|splat]

Prints ONE JSON line (rank 0).  Metric (BASELINE.json): Msamples/s = S*H*W / time for the
denoiser's forward+backward at 1280x720, 8 spp, 21x21 kernels, fp32, inputs resident in HBM.

Workloads
  model  (default; BASELINE.json configs[2], the configuration the metric is quoted on)
         one step = the reference training step (sbmc/interfaces.py:78-105): Multisteps
         forward, TonemappedRelativeMSE, backward, non-finite guard, grad-norm clip 1000,
         Adam(1e-4).  The conv backbone rides MIOpen; the splat path is the hand-written
         HIP kernels.
  infer  (BASELINE.json configs[1] with --spp 4) one step = Multisteps forward only, eval mode,
         no_grad -- what scripts/denoise.py runs per frame.  Not the headline metric.
  splat  one step = S progressive splat updates (ProgressiveKernelApply(splat=True), one per
         sample) + normalisation sum_r/(sum_w+eps) + backward to logits and radiance
         (SURVEY.md section 8d metric (i)): the part of the step the HIP kernels replace.
         With the default workload this is also measured, after the timed region, and
         reported under "stages" together with the roofline figure of its dominant kernel.

Multi-GPU (--gpus N): ONE frame, split along H into N slabs (strong scaling), one process per
GPU over RCCL.  Under torch.distributed.run (WORLD_SIZE set) this process is one rank; a plain
`python bench.py --gpus N` starts the N ranks itself (re-executes through torch.distributed.run on
127.0.0.1) and rank 0 prints the one JSON line.  model: U-net halo exchange + cross-rank merge of
the splat's running state (sbmc_amd/dist.py) + gradient all-reduce; splat: every rank splats its
own samples into its slab extended by the kernel radius and merges the overhang with its neighbours.
"""
import argparse
import gc
import json
import os
import sys
import time

# MIOpen's default exhaustive "find" costs ~5 minutes on the first step of this model; the
# FAST mode picks solvers from its heuristics in seconds (and was not slower here).
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

import torch as th  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
FP32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32, dense (MI355X_MICROARCH.md)
F16_MFMA_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_f16 / bf16, dense, at 2.4 GHz (same guide; ~2.0-2.1 GHz sustained)


def fwd_bytes_per_pixel(k, c=3):
    # read logits 4k^2 + radiance 4c + read+write running state (c + 2 floats each way)
    return 4 * k * k + 4 * c + 2 * 4 * (c + 2)


def bwd_bytes_per_pixel(k, c=3):
    # read logits + write d_logits 8k^2, radiance r/w 8c, state + upstream grads ~ 40
    return 8 * k * k + 8 * c + 40


def splat_source_hash():
    """sha256 of what the splat kernels are built from (tools/pmc_summary.py records the same with every PMC pass)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("sbmc_amd/csrc/splat_fused.hip", "sbmc_amd/csrc/common.hpp"):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()


def measured_traffic(kernel_substr):
    """HBM bytes per launch from the newest committed PMC summary (profiles/r*_pmc.json, produced by tools/prof.sh:
    separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for gfx950) -- a figure of ANOTHER run (PMC counters cannot be read from inside a run), quoted only while the kernel
    sources are the ones that were profiled.  -> (bytes or None, source, note)"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    if not files:
        return None, None, "no committed PMC summary"
    try:
        d = json.load(open(files[-1]))
    except Exception:
        return None, None, "unreadable PMC summary"
    meta = d.get("_meta") or {}
    src = os.path.basename(files[-1])
    if meta.get("commit") or meta.get("date"):
        src += " (tree %s, %s)" % ((meta.get("commit") or "?")[:10], meta.get("date") or "?")
    if meta.get("kernel_source_sha256") != splat_source_hash():
        return None, src, ("profiles/%s was taken on other kernel sources than this tree's (splat_fused.hip / common.hpp "
                           "differ): not quoted; re-run tools/prof.sh" % os.path.basename(files[-1]))
    tot = 0
    for k, v in d.items():
        if k != "_meta" and any(sub in k for sub in kernel_substr):
            tot += int(v["hbm_bytes_per_launch"])
    return (tot or None), src, None


def model_flops(model, h, w, spp, train):
    """Algorithmic flop of one Multisteps pass as the reference computes it: every per-sample
    1x1 convolution at full resolution over its whole (per-sample + context) input, every U-net convolution at
    its level's resolution; a training step = forward + data gradient + weight gradient of every convolution
    (no data gradient for the network's input)."""
    import torch.nn as nn
    fwd = first = 0.0
    for name, m in model.named_modules():
        if not isinstance(m, nn.Conv2d):
            continue
        lvl = name.count("next_level")
        per = 2.0 * m.in_channels * m.out_channels * m.kernel_size[0] * m.kernel_size[1]
        px = (h >> lvl) * (w >> lvl) * (1 if name.startswith("propagation") else spp)
        fwd += per * px
        if name == "embedding_00.layer_0.layer.0":
            first = per * px
    return 3.0 * fwd - first if train else fwd


def unet_layout():
    """What modules.unet_channels_last decided for the U-nets of this run."""
    from sbmc_amd import modules
    mode = os.environ.get("SBMC_UNET_LAYOUT", "auto").lower()
    picks = set(modules._LAYOUT_DECISIONS.values())
    if mode in ("nchw", "nhwc"):
        return {"nchw": "planar (forced)", "nhwc": "channels_last (forced)"}[mode]
    if not picks:
        return None
    if any(len(k) == 2 and k[1] == "own 3x3 kernel" for k in modules._LAYOUT_DECISIONS):
        return "channels_last (own 3x3 convolution kernel)" if picks == {True} else "mixed"
    return "channels_last (measured)" if picks == {True} else "planar (measured)" if picks == {False} else "mixed"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["model", "splat", "infer"], default="model")
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--ksize", type=int, default=21)
    ap.add_argument("--fp16-activations", action="store_true",
                    help="informational (BASELINE configs[4]): run the network under torch.autocast(float16) "
                         "(infer, or the single-GPU training step); splat arithmetic stays fp32. Never the "
                         "fp32 metric.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stages", action="store_true")
    return ap.parse_args()


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks through torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1, a free port) with the same arguments; rank 0's
    JSON line goes to this process's stdout.  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def make_splat_inputs(h, w, spp, k, device, seed):
    g = th.Generator(device="cpu").manual_seed(seed)
    rad = [th.empty(1, 3, h, w).exponential_(1.0, generator=g).to(device).requires_grad_()
           for _ in range(spp)]
    logits = []
    for _ in range(spp):
        t = th.empty(1, k * k, h, w, device=device)
        t.normal_(0, 1)
        logits.append(t.requires_grad_())
    d_out = th.randn(1, 3, h, w, generator=g).to(device)
    return rad, logits, d_out


def splat_step(update, rad, logits, d_out, eps=1e-8):
    for t in logits:
        t.grad = None
    for t in rad:
        t.grad = None
    sum_r = sum_w = max_w = None
    for r, kk in zip(rad, logits):
        sum_r, sum_w, max_w = update(r, kk, sum_r, sum_w, max_w)
    out = sum_r / (sum_w + eps)
    out.backward(d_out)
    return out


def make_model_inputs(h, w, spp, device, seed, rows=None):
    """Synthetic batch.  rows=(r0, r1): only those rows are generated -- the frame is drawn in
    blocks of ROW_BLOCK rows, each from its own seeded generator, so that a rank of a sharded run draws exactly
    its slab of the SAME frame the single-GPU run sees without ever holding the whole frame (2.7 GB of features
    at 720p x 8 spp, 25 GB at 4K -- per rank)."""
    r0, r1 = (0, h) if rows is None else rows
    g = th.Generator(device="cpu").manual_seed(seed)
    gf = th.rand(1, 3, 1, 1, generator=g)
    rad, feat, tgt = [], [], []
    for blk in range(r0 // ROW_BLOCK, (r1 + ROW_BLOCK - 1) // ROW_BLOCK):
        gb = th.Generator(device="cpu").manual_seed(seed * 100003 + blk + 1)
        nb = min(ROW_BLOCK, h - blk * ROW_BLOCK)
        r = th.empty(1, spp, 3, nb, w).exponential_(1.0, generator=gb)
        f = th.rand(1, spp, 93, nb, w, generator=gb)
        lr = th.log1p(r) / 10.0  # radiance channels of the feature vector (datasets.py:760-768)
        f[:, :, 5:8] = lr
        f[:, :, 8:11] = lr
        t = th.empty(1, 3, nb, w).exponential_(1.0, generator=gb)
        lo, hi = max(r0 - blk * ROW_BLOCK, 0), min(r1 - blk * ROW_BLOCK, nb)
        rad.append(r[..., lo:hi, :])
        feat.append(f[..., lo:hi, :])
        tgt.append(t[..., lo:hi, :])
    rad, feat, tgt = th.cat(rad, -2), th.cat(feat, -2), th.cat(tgt, -2)
    return {"radiance": rad.contiguous().to(device), "features": feat.contiguous().to(device),
            "global_features": gf.to(device), "target_image": tgt.contiguous().to(device)}


ROW_BLOCK = 8


def train_step(model, opt, loss_fn, batch, fp16=False):
    """The reference training step, sbmc/interfaces.py:78-105.  fp16 (informational, never the fp32
    metric): the network runs under torch.autocast(float16) -- half activations end to end, fp32 splat
    arithmetic, fp32 loss and optimizer."""
    from sbmc_amd.utils import crop_like
    opt.zero_grad()
    with th.autocast("cuda", dtype=th.float16, enabled=fp16):
        out = model(batch)["radiance"]
    out = out.float()
    tgt = crop_like(batch["target_image"], out)
    loss = loss_fn(out, tgt)
    loss.backward()
    if not th.isfinite(loss).item():
        raise RuntimeError("non-finite loss")
    th.nn.utils.clip_grad_norm_(model.parameters(), 1000)
    opt.step()
    return loss


def validate_sharded(model, runner, loss_fn, batch, part, H, W, S, device):
    """Before anything is timed on N > 1 GPUs: ONE sharded training step (SGD with lr 0: nothing moves) whose loss
    must equal the loss of rank 0's own single-GPU forward of the whole frame -- the first thing a real multi-GPU
    node runs is a correctness check, not a timing.  If a rank's halo mailbox reports a time-out (a platform on
    which the IPC path misbehaves between devices), ALL ranks fall back to torch.distributed P2P together
    (`ShardedDenoiser.settle_transport`) and the step is repeated: a slower line instead of a hang.  Collective."""
    from sbmc_amd import dist as sdist
    from sbmc_amd.utils import crop_like
    sgd = th.optim.SGD(model.parameters(), lr=0.0)
    loss = None
    try:
        loss = float(runner.train_step(sgd, loss_fn, batch))
    except sdist.HaloTimeout as e:              # (every rank gets it together and has settled the transport already)
        runner.transport_note = (runner.transport_note or "") + " [first sharded step: %s]" % e
        loss = float(runner.train_step(sgd, loss_fn, batch))
    out = {"sharded_loss": loss, "single_gpu_loss": None, "rel_diff": None, "bound": 1e-5}
    if S * H * W > 30e6:
        out["note"] = "whole-frame reference skipped: the frame does not fit rank 0's GPU next to its slab"
    elif part.rank == 0:
        full = make_model_inputs(H, W, S, device, seed=1234)
        with th.no_grad():
            o = model(full)["radiance"]
            ref = float(loss_fn(o, crop_like(full["target_image"], o)))
        del full, o
        out["single_gpu_loss"] = ref
        out["rel_diff"] = float("%.3g" % (abs(loss - ref) / max(abs(ref), 1e-30)))
    for q in model.parameters():
        q.grad = None
    th.cuda.empty_cache()
    return out


def scale_ok(pm, H, W, S):
    """Was a committed PMC pass taken at this run's frame size?"""
    return pm.get("height") == H and pm.get("width") == W and pm.get("spp") == S


def cpu_baseline(args, device=None):
    """Times the CPU port on a bounded sample of the same workload (rank 0, N=1 only).

    model workload: the same Python model code with torch-CPU convolutions and the oracle's splat
    operators behind the `*_cpu_float32` names, on a full-width, QUARTER-height crop of the real frame
    (all samples): one timed step, behind a warm-up step and a timed step on a sixteenth of the height (Adam
    state, thread pools, allocator -- at full crop size a warmed step measured 3 % faster than the first one, 71.2 s vs
    73.5 s at 1280x180, which does not justify doubling the CPU time).  The op-level figure
    (SURVEY.md 8d metric (i): the oracle's splat forward + backward alone, same crop) rides in the
    same object.  The same seeded model and inputs are also run on the GPU, which gives the parity
    figures of BASELINE.json's metric ("PSNR vs ref", SURVEY.md section 8d: 10*log10(1/MSE) after
    the display curve x/(1+x), and the max relative error)."""
    from oracle import sbmc_oracle as orc
    from sbmc_amd import halide_ops
    orc.lib()
    k, spp = args.ksize, args.spp
    threads = th.get_num_threads()
    th.manual_seed(0)
    parity = None
    # a full-width crop of a sixteenth of the height (44 rows at 720p: the kernel's support + a band of output rows):
    # one warm-up step and THREE timed steps on the same crop fit the ~1-2 minutes the default run may spend here
    h, w = max(2 * k + 2, (args.height // 16) // 4 * 4), args.width

    # op level: S progressive updates + normalise + backward on the oracle (C operators + torch-CPU glue)
    def splat_cpu_once(hh=None):
        hh = hh or h
        rad = [th.empty(1, 3, hh, w).exponential_(1.0).requires_grad_() for _ in range(spp)]
        logits = [th.randn(1, k * k, hh, w).requires_grad_() for _ in range(spp)]
        d_out = th.randn(1, 3, hh, w)
        t0 = time.perf_counter()
        splat_step(lambda d, kk, a, b, m: orc.progressive_kernel_apply(d, kk, a, b, m, splat=True),
                   rad, logits, d_out)
        return time.perf_counter() - t0
    splat_cpu_once()                                    # warm-up (thread pool, page faults)
    sdts = sorted(splat_cpu_once() for _ in range(3))
    splat_dt = sdts[1]
    op = {"value": round(spp * h * w / splat_dt / 1e6, 4), "unit": "Msamples/s",
          "sample": "oracle splat fwd+bwd (%d x progressive_kernel_apply + normalise + backward) on %dx%d, "
                    "%d spp, k=%d: median of 3 after 1 warm-up = %.2f s" % (spp, w, h, spp, k, splat_dt)}
    # the same op on a QUARTER of the height (SURVEY.md 8d's crop: the sixteenth above has a larger share of border rows
    # and sits better in the host's caches) -- one run behind a warm-up, bounded to what fits ~48 GB of host memory
    hq = (args.height // 4) // 4 * 4
    if hq > h and 4.0 * spp * k * k * hq * w * 6 < 48e9:
        try:
            splat_cpu_once(hq)
            qdt = splat_cpu_once(hq)
            op["quarter_height"] = {"value": round(spp * hq * w / qdt / 1e6, 4), "unit": "Msamples/s",
                                    "sample": "the same on %dx%d (a quarter of the height): 1 run after 1 warm-up = %.2f s" % (w, hq, qdt)}
        except Exception as e:          # (informational: a host short of memory must not sink the line)
            op["quarter_height"] = {"error": repr(e)}
    if args.workload != "model":
        base = dict(op)
        base.update({"cores": threads, "kind": "port"})
        base["sample"] += " on %d threads (host has %d logical cpus)" % (threads, os.cpu_count())
        return base, None

    from sbmc_amd import Multisteps, losses
    halide_ops.register_cpu_ops_for_testing(orc)
    try:
        model = Multisteps(93, 3, ksize=k)
        model.train()
        state = {n: v.clone() for n, v in model.state_dict().items()}
        opt = th.optim.Adam(model.parameters(), lr=1e-4)
        batch = make_model_inputs(h, w, spp, "cpu", seed=1)
        with th.no_grad():
            ref_out = model(batch)["radiance"]
        loss_fn = losses.TonemappedRelativeMSE()
        t0 = time.perf_counter()
        train_step(model, opt, loss_fn, batch)           # warm-up: Adam state, thread pools, allocator
        warm = time.perf_counter() - t0
        t0 = time.perf_counter()
        train_step(model, opt, loss_fn, batch)
        dt16 = time.perf_counter() - t0
        # THE figure: one timed step on a QUARTER of the height (SURVEY.md 8d's crop; VERDICT r5: the sixteenth flatters
        # neither side equally -- fewer border rows per output row on the larger crop), behind the two steps above
        hq = max(h, (args.height // 4) // 4 * 4)
        qbatch = make_model_inputs(hq, w, spp, "cpu", seed=1) if hq > h else batch
        t0 = time.perf_counter()
        train_step(model, opt, loss_fn, qbatch)
        dt = time.perf_counter() - t0
        del qbatch
        model.load_state_dict(state)                     # (the parity run below compares the seeded weights)
    finally:
        halide_ops.register_cpu_ops_for_testing(None)
    if device is not None:
        gmodel = Multisteps(93, 3, ksize=k)
        gmodel.load_state_dict(state)
        gmodel.to(device).train()
        with th.no_grad():
            out = gmodel({n: v.to(device) for n, v in batch.items()})["radiance"].cpu()
        tm = lambda x: x.clamp(min=0) / (1 + x.clamp(min=0))  # noqa: E731
        mse = ((tm(out) - tm(ref_out)) ** 2).mean().item()
        parity = {
            "psnr_db_vs_cpu_oracle": round(10 * __import__("math").log10(1.0 / max(mse, 1e-30)), 1),
            "max_rel_err": float("%.3g" % ((out - ref_out).abs() / (ref_out.abs() + 1e-6)).max().item()),
            "max_abs_err_over_max": float("%.3g" % ((out - ref_out).abs().max() / ref_out.abs().max()).item()),
            "sample": "Multisteps forward, %dx%d, %d spp, k=%d: this build on the GPU (HIP splat, own split-precision "
                      "3x3 and 1x1 convolution kernels) vs the same model on the CPU (oracle splat operators + torch-CPU "
                      "fp32 convolutions)" % (w, h, spp, k),
            # what the GRADIENTS are held to (tests/) -- the forward figures above say nothing
            # about them
            "grad_bound": "operator level (every splat / 1x1 / 3x3 kernel vs the oracle or float64): 1e-5 of the "
                          "tensor's scale; whole-model parameter gradients at test sizes: 1e-5 of a float64 evaluation "
                          "or no further from it than 2-3x the reference-order fp32 gradient is (fp32 sums of 1e3-1e4 "
                          "mixed-sign terms sit at 0.3-1.2e-5 themselves); at 1280x720x8spp (sums of 7.4 M terms): "
                          "within 2x the MEASURED fp32 noise floor between two single-GPU evaluations, up to 4e-4 of a "
                          "gradient's scale -- not 1e-5"}
    base = {
        "value": round(spp * hq * w / dt / 1e6, 4), "unit": "Msamples/s",
        "cores": threads, "kind": "port",
        "sample": "Multisteps training step (torch-CPU convs + oracle splat ops) on a %dx%d crop (full width, a QUARTER of "
                  "the height: SURVEY.md 8d) of the frame, %d spp, k=%d: ONE timed step = %.1f s, behind a warm-up step (%.1f s) "
                  "and a timed step (%.1f s) on a %dx%d crop (a sixteenth of the height), on %d threads (host has %d "
                  "logical cpus)" % (w, hq, spp, k, dt, warm, dt16, w, h, threads, os.cpu_count()),
        "sixteenth_height": {"value": round(spp * h * w / dt16 / 1e6, 4), "unit": "Msamples/s",
                             "sample": "the same step on %dx%d: 1 timed step after 1 warm-up = %.1f s" % (w, h, dt16)},
        "splat_op": op,
    }
    return base, parity


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not th.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (MI355X)")
    # test hooks (single-GPU dry run of the multi-rank code path): all ranks on device 0 over gloo
    backend = os.environ.get("SBMC_BENCH_BACKEND", "nccl")
    if os.environ.get("SBMC_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    th.cuda.set_device(local_rank)
    device = th.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from sbmc_amd import _lib, functions, modules
    from sbmc_amd import dist as sdist
    _lib.lib()  # fail loudly if the HIP extension is missing

    H, W, S, K = args.height, args.width, args.spp, args.ksize
    pad = (K - 1) // 2
    is_model = args.workload in ("model", "infer")
    infer = args.workload == "infer"
    # SURVEY.md 8d protocol: >= 5 warm-up and >= 20 timed iterations (whole run: a few minutes)
    steps = args.steps if args.steps is not None else 20
    warmup = args.warmup if args.warmup is not None else 5
    if is_model:
        # the first TWO steps carry one-time work (MIOpen solver selection forward and backward,
        # Adam state allocation, caching-allocator growth): never time them
        warmup = max(warmup, 2)

    def sync():
        th.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            th.cuda.synchronize(device)

    def timed(step_fn, nwarm, nsteps, store=None, events_inside=True, inside=None):
        """Wall time of `nsteps` steps.  `store` collects (name, start, end) HIP-event triples of the
        fused operators: inside the timed steps when `events_inside`, else in two extra, untimed
        steps: with ~50 event pairs per step switched on at the start of the 3 timed model steps they
        read 579 instead of 554 ms per step (not so when the events are already on during the warm-up,
        nor in the splat workload -- a start-up effect, not isolated further), which must not leak
        into the model workload's `value`.  `inside` = (list, name prefixes): with `events_inside` off, the operators of
        these names (the splat kernels: two event pairs per step) are timed INSIDE the timed steps all the same -- into
        that list, warm-up steps included so that nothing switches on at the start of the timed region."""
        if os.environ.get("SBMC_BENCH_EVENTS_OUTSIDE"):   # debugging aid: never any event in a timed step
            events_inside = False
        if inside is not None and not events_inside:
            functions.enable_kernel_timing(inside[0], inside[1])
        for _ in range(nwarm):
            step_fn()
        sync()
        if inside is not None:
            del inside[0][:]                               # (the warm-up steps' launches)
        # as interfaces.train does after its first step: what is alive now stays alive, keep the
        # cyclic collector from walking it (a full collection costs ~50 ms of host time here and
        # lands right after a step's loss.item() sync, where the GPU waits for the host)
        gc.collect()
        gc.freeze()
        if events_inside or inside is None:
            functions.enable_kernel_timing(store if events_inside else None)
        marks = [th.cuda.Event(enable_timing=True) for _ in range(nsteps + 1)]   # one per step boundary
        stream = th.cuda.current_stream(device)
        t0 = time.perf_counter()
        for i in range(nsteps):
            marks[i].record(stream)
            step_fn()
            if os.environ.get("SBMC_BENCH_VERBOSE"):   # debugging aid: adds a sync per step
                th.cuda.synchronize(device)
                print("step %d: %.1f ms since start" % (i, (time.perf_counter() - t0) * 1e3),
                      file=sys.stderr, flush=True)
        marks[nsteps].record(stream)
        sync()
        dt = time.perf_counter() - t0
        functions.enable_kernel_timing(None)
        # per-step device time between the boundary events (no extra synchronisation): the median
        # SURVEY.md 8d asks for; `value` stays the whole-region figure the driver's contract defines
        per_step = sorted(marks[i].elapsed_time(marks[i + 1]) * 1e-3 for i in range(nsteps))
        med = per_step[nsteps // 2] if nsteps % 2 else 0.5 * (per_step[nsteps // 2 - 1] + per_step[nsteps // 2])
        if store is not None and not events_inside:
            functions.enable_kernel_timing(store)
            for _ in range(2):
                step_fn()
            sync()
            functions.enable_kernel_timing(None)
        if world > 1:
            t = th.tensor([dt, med], dtype=th.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, med = t[0].item(), t[1].item()
        timed.median = med
        timed.spread = (per_step[0], per_step[-1])
        return dt

    part = sdist.SlabPartition(H, world, rank)
    timings = []
    model_timings = []
    in_step = []          # the splat operators' launches INSIDE the training step's timed region (N = 1)
    validation = None

    # ---------------------------------------------------------------- main timed region
    if is_model:
        from sbmc_amd import Multisteps, losses
        th.manual_seed(0)
        model = Multisteps(93, 3, ksize=K).to(device)
        model.train(not infer)
        opt = th.optim.Adam(model.parameters(), lr=1e-4, fused=True)   # same update, one kernel
        loss_fn = losses.TonemappedRelativeMSE()
        if infer:
            if world > 1:
                batch = make_model_inputs(H, W, S, device, seed=1234, rows=(part.y0, part.y1))
                runner = sdist.ShardedDenoiser(model, part)
            else:
                batch = make_model_inputs(H, W, S, device, seed=1234)
                runner = model

            def step():
                with th.no_grad(), th.autocast("cuda", dtype=th.float16, enabled=args.fp16_activations):
                    runner(batch)
        elif world == 1:
            batch = make_model_inputs(H, W, S, device, seed=1234)

            def step():
                train_step(model, opt, loss_fn, batch, fp16=args.fp16_activations)
        else:
            batch = make_model_inputs(H, W, S, device, seed=1234, rows=(part.y0, part.y1))
            runner = sdist.ShardedDenoiser(model, part)
            sync()      # (the ranks finish building their slabs seconds apart: the first exchange must not wait for that)
            validation = validate_sharded(model, runner, loss_fn, batch, part, H, W, S, device)
            sync()      # (rank 0 has just drawn and denoised the whole frame alone: nobody's exchange waits through that)
            if rank == 0 and validation["rel_diff"] is not None and validation["rel_diff"] > 1e-3:
                # (1e-5 is the bar and is reported; beyond 1e-3 the sharded step is WRONG: no number is printed)
                print(json.dumps({"error": "sharded step disagrees with the single-GPU step", "validation": validation}))
                raise SystemExit(3)

            def step():
                try:
                    runner.train_step(opt, loss_fn, batch)
                except sdist.HaloTimeout:
                    # (raised on every rank together; the ranks have left the IPC transport: the step is repeated
                    # over torch.distributed P2P and the line says so in `transport` / `transport_note`)
                    runner.train_step(opt, loss_fn, batch)
        dt = timed(step, warmup, steps, timings if world == 1 else None, events_inside=False,
                   inside=(in_step, ("splat_update_",)) if (world == 1 and not infer) else None)
        med_s, spread = timed.median, timed.spread
        model_timings = timings
        if world > 1:
            runner.check()
        # source pixels whose logits this rank's splat kernels stream (its own rows: the overhang of the
        # running state is exchanged, nothing is recomputed)
        local_px = part.rows * W
    elif world == 1:
        update = modules.ProgressiveKernelApply(splat=True)
        rad, logits, d_out = make_splat_inputs(H, W, S, K, device, seed=1234)
        dt = timed(lambda: splat_step(update, rad, logits, d_out), warmup, steps, timings)
        med_s, spread = timed.median, timed.spread
        local_px = H * W
    else:
        # every rank splats the samples of its own rows into its slab extended by the kernel radius,
        # exchanges the overhang rows of the running state with its neighbours and merges them
        # (sbmc_amd/dist.py), normalises and runs the backward -- no other communication
        rad, logits, d_out = make_splat_inputs(part.rows, W, S, K, device, seed=1234 + rank)
        all_rad = th.stack([r.detach() for r in rad], 1).requires_grad_()
        all_log = th.stack([t.detach() for t in logits], 1).requires_grad_()
        del rad, logits
        slab = (pad if part.has_up else 0, pad if part.has_down else 0, not part.has_up, not part.has_down)
        if part.min_rows < pad or not functions.splat_slab_supported(all_rad, all_log, slab[0], slab[1]):
            raise SystemExit("bench.py --workload splat --gpus %d: slabs of %d rows are not supported" % (
                world, part.rows))

        def slab_step():
            all_rad.grad = None
            all_log.grad = None
            st = functions.SplatAll.apply(all_rad, all_log, *slab)
            sr, sw, _ = sdist.merge_overhang(st[0], st[1], st[2], pad, part)
            (sr / (sw + 1e-8)).backward(d_out)
        dt = timed(slab_step, warmup, steps, timings)
        med_s, spread = timed.median, timed.spread
        local_px = part.rows * W

    # ---------------------------------------------------------------- per-rank split of the sharded step
    per_rank = None
    if is_model and world > 1:
        # two more, untimed, steps with events around every neighbour exchange (launches on this rank's stream:
        # they include the wait for the neighbour's rows) and around the wait for the gradient all-reduces the
        # backward did not hide; "compute" is the rest of the step
        store = []
        marks = [th.cuda.Event(enable_timing=True) for _ in range(3)]
        functions.enable_kernel_timing(store)
        for i in range(2):
            marks[i].record()
            step()
        marks[2].record()
        sync()
        functions.enable_kernel_timing(None)
        tot = {}
        for name, a, b in store:
            tot[name] = tot.get(name, 0.0) + a.elapsed_time(b) / 2
        step_ms = marks[0].elapsed_time(marks[2]) / 2
        mine = {"rank": rank, "rows": part.rows, "step_ms": round(step_ms, 3),
                "exchange_ms": round(tot.get("halo_exchange", 0.0), 3),
                "exchanges": sum(1 for n, _, _ in store if n == "halo_exchange") // 2,
                "all_reduce_exposed_ms": round(tot.get("grad_all_reduce_exposed", 0.0), 3)}
        mine["compute_ms"] = round(step_ms - mine["exchange_ms"] - mine["all_reduce_exposed_ms"], 3)
        mine["transport"] = runner.transport
        mine["handshake_ms"] = None if part.channel is None or part.channel.handshake_ms is None else round(part.channel.handshake_ms, 3)
        mine["device"] = th.cuda.get_device_name(device) + " #%d" % device.index
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        model_timings = [t for t in store if t[0].startswith("pointwise")]

    # ---------------------------------------------------------------- the same step on the fp32 matrix pipe only
    fp32_pipe = None
    if is_model and not infer and world == 1 and not args.no_stages and not args.fp16_activations:
        # `value` is measured with the convolutions at fp32 ACCURACY on the f16 / bf16 matrix pipes (split precision,
        # DESIGN.md 4.6 / 4.10).  The same step with every product on the fp32 pipe (MIOpen's fp32 3x3 solvers, the
        # fp32-MFMA 1x1 kernels), same run, same box: 2 warm-up + 5 timed steps
        keep = {k_: os.environ.get(k_) for k_ in ("SBMC_CONV3X3", "SBMC_HIP_PW_SPLIT", "SBMC_HIP_PW_GWS")}
        os.environ.update({"SBMC_CONV3X3": "0", "SBMC_HIP_PW_SPLIT": "0", "SBMC_HIP_PW_GWS": "0"})
        try:
            fdt = timed(step, 2, 5, None)
            fp32_pipe = {"ms_per_step": round(fdt / 5 * 1e3, 3), "value": round(S * H * W / (fdt / 5) / 1e6, 2),
                         "steps": 5, "warmup": 2,
                         "how": "SBMC_CONV3X3=0 SBMC_HIP_PW_SPLIT=0 SBMC_HIP_PW_GWS=0: 3x3 convolutions on MIOpen's fp32 "
                                "NHWC solvers, 1x1 layers on v_mfma_f32_32x32x2_f32"}
        except Exception as e:      # informational: must never sink the line
            fp32_pipe = {"error": repr(e)}
        finally:
            for k_, v_ in keep.items():
                if v_ is None:
                    os.environ.pop(k_, None)
                else:
                    os.environ[k_] = v_

    # ---------------------------------------------------------------- north_star's matrix: forward only
    infer_stages = None
    if is_model and not infer and not args.no_stages and not args.fp16_activations:
        # 1280x720 (this run's frame) at 4 / 8 / 32 spp, eval mode, no_grad, 5 warm-up + 10 timed frames each
        # (the timing protocol of the reference's scripts/denoise.py:152-165), on this run's ranks
        infer_stages = {}
        del batch
        opt.zero_grad(set_to_none=True)
        th.cuda.empty_cache()
        model.train(False)
        for spp_i in (4, 8, 32):
            if world > 1:
                b_i = make_model_inputs(H, W, spp_i, device, seed=1234, rows=(part.y0, part.y1))
                fwd = runner
            else:
                b_i = make_model_inputs(H, W, spp_i, device, seed=1234)
                fwd = model
            b_i.pop("target_image")

            def frame():
                with th.no_grad():
                    fwd(b_i)
            idt = timed(frame, 5, 10, None)
            infer_stages["infer_%dspp" % spp_i] = {
                "workload": "Multisteps(93,3,ksize=%d) forward only (eval, no_grad), %dx%d, %d spp" % (K, W, H, spp_i),
                "value": round(spp_i * H * W / (idt / 10) / 1e6, 2), "unit": "Msamples/s",
                "ms_per_frame": round(idt / 10 * 1e3, 3), "ms_per_frame_median": round(timed.median * 1e3, 3),
                "TFLOPs": round(model_flops(model, H, W, spp_i, False) / (idt / 10) / 1e12, 1),
                "steps": 10, "warmup": 5}
            del b_i
            th.cuda.empty_cache()
        if world > 1:
            runner.check()
        model.train(True)

    # ---------------------------------------------------------------- splat stages (N=1, model)
    stage = stage_all = stage_f16 = None
    step_flops = model_flops(model, H, W, S, not infer) if is_model else None
    layout = unet_layout() if is_model else None
    if is_model and not infer and world == 1 and not args.no_stages:
        del model, opt
        th.cuda.empty_cache()
        if os.environ.get("SBMC_BENCH_SPLAT_COOLDOWN"):      # measurement aid: seconds of idle device before the splat stages
            sync()
            time.sleep(float(os.environ["SBMC_BENCH_SPLAT_COOLDOWN"]))
        timings = []
        update = modules.ProgressiveKernelApply(splat=True)
        rad, logits, d_out = make_splat_inputs(H, W, S, K, device, seed=1234)
        sdt = timed(lambda: splat_step(update, rad, logits, d_out), 5, 20, timings)
        local_px = H * W
        stage = {"workload": "splat fwd+bwd only, reference module API: %d x ProgressiveKernelApply("
                             "splat=True) + normalise + backward" % S,
                 "value": round(S * H * W / (sdt / 20) / 1e6, 2), "unit": "Msamples/s",
                 "ms_per_step": round(sdt / 20 * 1e3, 3), "ms_per_step_median": round(timed.median * 1e3, 3),
                 "steps": 20, "warmup": 5}
        # the same work the way Multisteps issues it: all samples in one launch per kernel
        all_rad = th.stack([r.detach() for r in rad], 1).requires_grad_()
        all_log = th.stack([t.detach() for t in logits], 1).requires_grad_()
        del rad, logits
        th.cuda.empty_cache()
        if functions.splat_all_supported(all_rad, all_log):
            def all_step():
                all_rad.grad = None
                all_log.grad = None
                sr, sw, _ = functions.SplatAll.apply(all_rad, all_log)
                (sr / (sw + 1e-8)).backward(d_out)
            adt = timed(all_step, 5, 20, timings)
            stage_all = {"workload": "splat fwd+bwd only, as Multisteps issues it: functions.SplatAll "
                                     "(all %d samples per launch) + normalise + backward" % S,
                         "value": round(S * H * W / (adt / 20) / 1e6, 2), "unit": "Msamples/s",
                         "ms_per_step": round(adt / 20 * 1e3, 3), "ms_per_step_median": round(timed.median * 1e3, 3),
                 "steps": 20, "warmup": 5}
            # informational: the same with fp16 logit storage (BASELINE configs[4] "fp16 activations";
            # NOT the fp32 metric -- reported separately, never mixed into `value`)
            all_log16 = all_log.detach().half().requires_grad_()
            del all_log
            th.cuda.empty_cache()

            def all_step16():
                all_rad.grad = None
                all_log16.grad = None
                sr, sw, _ = functions.SplatAll.apply(all_rad, all_log16)
                (sr / (sw + 1e-8)).backward(d_out)
            if functions.splat_all_supported(all_rad, all_log16):   # (half logits: the k = 21 strip kernels only)
                hdt = timed(all_step16, 5, 20, timings)
                stage_f16 = {"workload": "as splat_all_samples but with fp16 logit / logit-gradient storage "
                                         "(fp32 arithmetic)", "dtype": "f16 storage, f32 math",
                             "value": round(S * H * W / (hdt / 20) / 1e6, 2), "unit": "Msamples/s",
                             "ms_per_step": round(hdt / 20 * 1e3, 3), "ms_per_step_median": round(timed.median * 1e3, 3),
                             "steps": 20, "warmup": 5}

    # per-call device time of the fused operators (events on the launch stream)
    per = {}
    for name, a, b in timings:
        per.setdefault(name, []).append(a.elapsed_time(b))  # ms
    kern = {}
    for _once in (0,):
        for name, bpp, nsamp in (("splat_update_fwd", fwd_bytes_per_pixel(K), 1),
                                 ("splat_update_bwd", bwd_bytes_per_pixel(K), 1),
                                 ("splat_update_fwd_all", fwd_bytes_per_pixel(K), S),
                                 ("splat_update_bwd_all", bwd_bytes_per_pixel(K), S)):
            if name in per:
                if os.environ.get("SBMC_BENCH_DUMP_KERNELS"):          # measurement aid: every launch's time
                    print(name, " ".join("%.3f" % t for t in per[name]), file=sys.stderr, flush=True)
                avg_ms = sum(per[name]) / len(per[name])
                kern[name] = {"calls": len(per[name]), "samples_per_launch": nsamp,
                              "avg_ms": round(avg_ms, 4), "alg_bytes": local_px * bpp * nsamp,
                              "GBps": round(local_px * bpp * nsamp / (avg_ms * 1e-3) / 1e9, 1)}

    # the same two operators INSIDE the training step's timed region (every timed step; the tensors are the step's own:
    # allocated once at the start of the process.  The isolated stages above run on 13 GB tensors allocated after the
    # model's memory went back to the driver, and their 8-sample launches read 4.4 or 5.0-5.4 ms from run to run on
    # the same box -- every launch of a run alike: where the pages land, not the kernel, profiles/HISTORY.md)
    kern_step = {}
    per_s = {}
    for name, a, b in in_step:
        per_s.setdefault(name, []).append(a.elapsed_time(b))
    for name, bpp in (("splat_update_fwd_all", fwd_bytes_per_pixel(K)), ("splat_update_bwd_all", bwd_bytes_per_pixel(K))):
        if name in per_s:
            if os.environ.get("SBMC_BENCH_DUMP_KERNELS"):
                print("in step:", name, " ".join("%.3f" % t for t in per_s[name]), file=sys.stderr, flush=True)
            avg_ms = sum(per_s[name]) / len(per_s[name])
            kern_step[name] = {"calls": len(per_s[name]), "samples_per_launch": S, "avg_ms": round(avg_ms, 4),
                               "alg_bytes": local_px * bpp * S,
                               "GBps": round(local_px * bpp * S / (avg_ms * 1e-3) / 1e9, 1),
                               "where": "inside the timed steps of the training step (HIP events on the launch stream "
                                        "around the operator call, every timed step)"}

    # the per-sample 1x1 layers (csrc/pointwise.hip, csrc/pointwise_chain.hip), from two instrumented steps after the timed
    # ones.  They are HBM-bound: each is priced by its ALGORITHMIC bytes (every activation it must read or write once, 4 bytes
    # per value) against the HBM peak -- not by its matrix work (a fraction of the fp32 matrix peak read > 1 for
    # kernels that run on the f16 pipe).
    layers = {}
    for name, a, b in model_timings:
        if name.startswith("pointwise"):
            layers.setdefault(name, []).append(a.elapsed_time(b))
    npx = S * (local_px if world > 1 else H * W)
    for name in list(layers):
        v = layers[name]
        kind, dims = name.split(" ")[0], name.split(" ")[1]
        if "<-" in dims:                                   # a fused chain: C0xC1(xC2)<-K
            couts = [int(t) for t in dims.split("<-")[0].split("x")]
            cin = int(dims.split("<-")[1])
            chans = cin + (couts[-1] if "inference" in name else sum(couts))
            flop = 2.0 * npx * sum(k * c for k, c in zip([cin] + couts[:-1], couts))
        else:
            cout, cin = (int(t) for t in dims.split("x"))
            nprod = 1 if kind.endswith("fwd") or "no gx" in name else 2
            flop = 2.0 * cin * cout * npx * nprod
            # forward: x in, y out; backward: gy and x in (+ gx out)
            chans = cin + cout + (cin if (not kind.endswith("fwd") and "no gx" not in name) else 0)
        avg = sum(v) / len(v)
        gbps = 4.0 * chans * npx / (avg * 1e-3) / 1e9
        layers[name] = {"calls_per_step": len(v) // 2, "avg_ms": round(avg, 3), "alg_bytes": int(4 * chans * npx),
                        "GBps": round(gbps, 1), "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBPS, 3),
                        "TFLOPs_fp32_equivalent": round(flop / (avg * 1e-3) / 1e12, 1)}

    # the U-nets' 3 x 3 convolutions (csrc/conv3x3.hip: fp32 values from three f16 matrix products per term), as
    # the model issues them -- scale lookup and weight preparation included -- from the same two instrumented steps
    convs = {}
    for name, a, b in model_timings:
        if name.startswith("conv3x3"):
            kind, dims = name.split(" ")
            cout, cin = (int(t) for t in dims.split("@")[0].split("x"))
            bb, hh, ww = (int(t) for t in dims.split("@")[1].split("x"))
            c = convs.setdefault(kind, {"calls": 0, "ms": 0.0, "flop": 0.0})
            c["calls"] += 1
            c["ms"] += a.elapsed_time(b)
            c["flop"] += 2.0 * 9 * cin * cout * bb * hh * ww
    for kind in list(convs):
        c = convs[kind]
        tf = c["flop"] / (c["ms"] * 1e-3) / 1e12
        convs[kind] = {"calls_per_step": c["calls"] // 2, "ms_per_step": round(c["ms"] / 2, 2),
                       "TFLOPs_fp32_equivalent": round(tf, 1), "TFLOPs_f16_issued": round(3 * tf, 1),
                       "frac_of_f16_mfma_peak": round(3 * tf / F16_MFMA_PEAK_TFLOPS, 3)}

    if rank == 0:
        ms = dt / steps * 1e3
        value = S * H * W / (dt / steps) / 1e6
        res = {
            "metric": "Msamples/s (SxHxW) denoise %s, %dx%d %dspp %dx%d kernel" % (
                "fwd only" if infer else "fwd+bwd", W, H, S, K, K),
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 3),
            "ms_per_step_median": round(med_s * 1e3, 3),
            "ms_per_step_min_max": [round(spread[0] * 1e3, 3), round(spread[1] * 1e3, 3)],
            "value_at_median": round(S * H * W / med_s / 1e6, 2),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f16 activations, f32 splat math" if (is_model and args.fp16_activations) else "f32",
            # what "f32" means here: every tensor is fp32 in HBM and every sum is fp32; the convolutions'
            # products are formed from exact low-precision pieces of the fp32 operands
            "arith": None if (not is_model or args.fp16_activations) else
                     "fp32 storage and accumulation; 3x3 convolutions: operands as 2 f16 planes, 3 of 4 partial products "
                     "(error <= 2^-22 per term); 1x1 layers: the same two-plane form wherever the producer left the tensor's "
                     "largest magnitude in a device word (all but each network input's first layer, which runs 3 bf16 planes, "
                     "6 of 9 partial products, <= 2^-23 per term); "
                     "splat, losses, optimizer: plain fp32",
            "data": "synthetic",
            "world_size": world if world == 1 else dist.get_world_size(),
            "backend": None if world == 1 else dist.get_backend(),
            "config": {
                "workload": "Multisteps(93,3,ksize=%d) forward only (eval, no_grad)" % K if infer else
                            "Multisteps(93,3,ksize=%d) training step: fwd + TonemappedRelativeMSE + bwd "
                            "+ clip + Adam (BASELINE.json configs[2])" % K if is_model else
                            "splat fwd+bwd only: %d x ProgressiveKernelApply(splat=True) + normalise + "
                            "backward" % S,
                "height": H, "width": W, "spp": S, "ksize": K, "batch": 1,
                "parallelism": "single GPU" if world == 1 else
                               ("H-slabs x%d: U-net halo exchange, cross-rank merge of the splat state (%d "
                                "overhang rows), one flat gradient all-reduce" % (world, pad) if is_model else
                                "H-slabs x%d: own samples splatted into slab + %d overhang rows/side, running "
                                "state exchanged and merged" % (world, pad)),
            },
        }
        if step_flops is not None:
            res["whole_step_tflops"] = round(step_flops / (dt / steps) / 1e12, 1)
            # (above 1 since round 3: the 3 x 3 convolutions and the fp32 1 x 1 layers run on the f16 / bf16 matrix pipe
            # at fp32 accuracy -- three resp. six low-precision products per fp32 multiply-add)
            # (a multiple, not a fraction: most of the step's multiply-adds run on the f16 pipe)
            res["multiples_of_fp32_mfma_peak"] = round(step_flops / (dt / steps) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 3)
            res["unet_layout"] = layout
        if fp32_pipe is not None:
            res["ms_per_step_fp32_pipe"] = fp32_pipe.get("ms_per_step")
            res["fp32_pipe"] = fp32_pipe
        if world > 1 and is_model and not infer:
            res["transport"] = runner.transport
            res["transport_note"] = runner.transport_note
            res["rccl_ranks"] = dist.get_world_size()
            try:
                res["rccl_version"] = ".".join(str(v) for v in th.cuda.nccl.version())
            except Exception:
                res["rccl_version"] = None
            res["validation"] = validation
        if per_rank is not None:
            res["per_rank"] = per_rank
        if stage is not None or infer_stages:
            res["stages"] = {}
            if infer_stages:
                res["stages"].update(infer_stages)
            if stage is not None:
                res["stages"]["splat"] = stage
            if stage_all is not None:
                res["stages"]["splat_all_samples"] = stage_all
            if stage_f16 is not None:
                res["stages"]["splat_all_samples_fp16_storage"] = stage_f16
        if kern:
            res["kernels"] = kern
        if kern_step:
            res["kernels_in_step"] = kern_step
        if layers:
            res["pointwise_layers"] = layers
            # the 1x1 kernel furthest below its (HBM) roof; PMC traffic where a committed pass of THIS tree's kernels holds it
            worst = min(layers, key=lambda n: layers[n]["frac_of_hbm_peak"])
            traffic, tsrc = None, None
            try:
                pj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r06_pointwise_pmc.json")
                with open(pj) as f:
                    pm = json.load(f)
                ent = pm.get("kernels", {}).get(worst)
                if ent and scale_ok(pm, H, W, S):
                    traffic, tsrc = ent.get("hbm_bytes_per_launch"), "profiles/r06_pointwise_pmc.json (another run of this command)"
            except (OSError, ValueError):
                pass
            res["roofline_pointwise"] = {
                "kernel": worst, "bound": "hbm", "achieved": layers[worst]["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": layers[worst]["frac_of_hbm_peak"], "traffic": traffic, "traffic_source": tsrc,
                "alg_bytes": layers[worst]["alg_bytes"], "avg_ms": layers[worst]["avg_ms"],
                "note": "the per-sample 1x1 layer furthest below the HBM roof (two instrumented steps after the timed ones)"}
        if convs:
            res["conv3x3"] = convs
            tot_flop = sum(v["TFLOPs_fp32_equivalent"] * v["ms_per_step"] for v in convs.values())
            tot_ms = sum(v["ms_per_step"] for v in convs.values())
            res["roofline_conv3x3"] = {
                "kernel": "sbmc::conv3_kernel / conv3_wgrad_kernel (all U-net convolutions of a step, forward + both "
                          "gradients; the timed calls also hold the weight preparation)",
                "bound": "mfma", "achieved": round(3 * tot_flop / tot_ms, 1), "peak": F16_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(3 * tot_flop / tot_ms / F16_MFMA_PEAK_TFLOPS, 3), "traffic": None,
                "note": "f16 matrix operations issued = 3 per fp32 multiply-add (split precision); the same work on "
                        "the fp32 matrix pipe is bounded by %d TFLOP/s" % int(FP32_MFMA_PEAK_TFLOPS),
                "ms_per_step": round(tot_ms, 2)}
        rk = "splat_update_bwd_all" if "splat_update_bwd_all" in kern else "splat_update_bwd"
        if "splat_update_bwd_all" in kern_step:       # the training step's own launches, inside its timed region
            rk = "splat_update_bwd_all"
        if rk in kern or rk in kern_step:
            kb = kern_step[rk] if rk in kern_step else kern[rk]
            traffic, src, tnote = (None, None, "PMC passes exist for 1280x720, k = 21 on one GPU only")
            if (H, W, K) == (720, 1280, 21) and world == 1:   # the PMC summary was taken at this size
                traffic, src, tnote = measured_traffic(("splat_bwd_strip_kernel",))
                if traffic is not None:
                    traffic *= kb["samples_per_launch"]        # PMC figure is per 1-sample launch
            res["roofline"] = {
                "kernel": "sbmc::splat_bwd_strip_kernel<21,3> (%d sample(s) per launch; the timed call "
                          "also holds its per-pixel state pre-pass, ~1-3%%)" % kb["samples_per_launch"],
                "bound": "hbm", "achieved": kb["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(kb["GBps"] / HBM_PEAK_GBPS, 4),
                # HBM bytes per launch from the counters.  PMC counters cannot be read from inside a run: the figure
                # comes from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate passes, gfx950
                # correction) of this kernel at this size, named in `traffic_source` with the tree they were taken on
                "traffic": traffic,
                "traffic_measured_in_this_run": False,
                "traffic_note": tnote,
                "traffic_source": None if traffic is None else "profiles/" + src + ": per 1-sample launch x samples per "
                                  "launch; tools/prof.sh",
                "traffic_over_algorithmic": None if traffic is None else round(traffic / kb["alg_bytes"], 3),
                "alg_bytes_per_launch": kb["alg_bytes"],
                "avg_launch_ms": kb["avg_ms"],
                "launches": kb["calls"],
                # (identical kernels read ~4.4 or ~5.35 ms per 8-sample launch depending on where the allocator put the two
                # 13 GB tensors -- per placement, stable in time: tools/placement_experiment*.py, profiles/HISTORY.md)
                "placement_note": "bimodal in where the logit / gradient tensors live: 0.74-0.75 or 0.62 on identical kernels",
                "timed": kb.get("where", "the splat-only stage of this run (HIP events on the launch stream around the "
                                         "operator call, every timed step of that stage)"),
            }
            if rk in kern_step and rk in kern:
                # the same launch in the splat-only stage of this run (its tensors: allocated after the model's memory was freed)
                res["roofline"]["isolated_stage"] = {"avg_launch_ms": kern[rk]["avg_ms"], "achieved": kern[rk]["GBps"],
                                                     "frac": round(kern[rk]["GBps"] / HBM_PEAK_GBPS, 4),
                                                     "launches": kern[rk]["calls"]}
            if "splat_update_bwd" in kern and rk != "splat_update_bwd":
                # the same kernel launched per sample (the reference's module API, `stages.splat`): shorter launches
                # reach a lower rate -- ramp-up and tail of a 0.65 ms launch against a 4.5 ms one
                k1 = kern["splat_update_bwd"]
                res["roofline"]["one_sample_launches"] = {"avg_launch_ms": k1["avg_ms"], "achieved": k1["GBps"],
                                                          "frac": round(k1["GBps"] / HBM_PEAK_GBPS, 4)}
        fk = "splat_update_fwd_all" if "splat_update_fwd_all" in kern else "splat_update_fwd"
        if "splat_update_fwd_all" in kern_step:
            fk = "splat_update_fwd_all"
        if (fk in kern or fk in kern_step) and "roofline" in res:
            kf = kern_step[fk] if fk in kern_step else kern[fk]
            ftraffic, fsrc, fnote = (None, None, "PMC passes exist for 1280x720, k = 21 on one GPU only")
            if (H, W, K) == (720, 1280, 21) and world == 1:
                ftraffic, fsrc, fnote = measured_traffic(("splat_fwd_strip_kernel",))
                if ftraffic is not None:
                    ftraffic *= kf["samples_per_launch"]
            res["roofline_fwd"] = {
                "kernel": "sbmc::splat_fwd_strip_kernel<21,3> (%d sample(s) per launch; the timed call also holds "
                          "the per-pixel fold of the samples' partial states)" % kf["samples_per_launch"],
                "bound": "hbm", "achieved": kf["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(kf["GBps"] / HBM_PEAK_GBPS, 4), "traffic": ftraffic,
                "traffic_measured_in_this_run": False, "traffic_note": fnote,
                "traffic_source": None if ftraffic is None else "profiles/" + fsrc + ": per 1-sample launch x samples per "
                                  "launch; tools/prof.sh",
                "traffic_over_algorithmic": None if ftraffic is None else round(ftraffic / kf["alg_bytes"], 3),
                "alg_bytes_per_launch": kf["alg_bytes"], "avg_launch_ms": kf["avg_ms"], "launches": kf["calls"],
                "timed": kf.get("where", "the splat-only stage of this run"),
            }
            if fk in kern_step and fk in kern:
                res["roofline_fwd"]["isolated_stage"] = {"avg_launch_ms": kern[fk]["avg_ms"], "achieved": kern[fk]["GBps"],
                                                         "frac": round(kern[fk]["GBps"] / HBM_PEAK_GBPS, 4),
                                                         "launches": kern[fk]["calls"]}
            if "splat_update_fwd" in kern and fk != "splat_update_fwd":
                k1 = kern["splat_update_fwd"]
                res["roofline_fwd"]["one_sample_launches"] = {"avg_launch_ms": k1["avg_ms"], "achieved": k1["GBps"],
                                                              "frac": round(k1["GBps"] / HBM_PEAK_GBPS, 4)}
        if world == 1 and not args.no_cpu_baseline and not infer:
            try:
                res["cpu_baseline"], parity = cpu_baseline(args, device)
                if parity is not None:
                    res["parity"] = parity
            except Exception as e:  # the baseline must never sink the GPU measurement
                res["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
