#!/usr/bin/env python
"""Benchmark of the SBMC splat hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload splat|model]

Prints ONE JSON line (rank 0).  Metric (BASELINE.json): Msamples/s = S*H*W / time,
1280x720, 8 spp, 21x21 kernels, forward + backward, fp32, inputs resident in HBM.

Workloads
  splat  one step = S progressive splat updates (ProgressiveKernelApply(splat=True),
         one per sample) + normalisation sum_r/(sum_w+eps) + backward to the logits
         and the radiance -- SURVEY.md section 8d metric (i).  This is the part of the
         reference step that the hand-written kernels replace.
  model  one step = the reference training step (sbmc/interfaces.py:78-105): Multisteps
         forward, TonemappedRelativeMSE, backward, grad-norm clip 1000, Adam(1e-4) --
         SURVEY.md section 8d metric (ii).  The conv backbone rides MIOpen.

Multi-GPU (--gpus N under torch.distributed.run): the frame is split along H into N
slabs; every rank splats the samples of its slab extended by the kernel radius
(strong scaling of one frame, see DESIGN.md section "Multi-GPU").
"""
import argparse
import json
import os
import sys
import time

import torch as th

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def fwd_bytes_per_pixel(k, c=3):
    # read logits 4k^2 + radiance 4c + read+write running state (c + 2 floats each way)
    return 4 * k * k + 4 * c + 2 * 4 * (c + 2)


def bwd_bytes_per_pixel(k, c=3):
    # read logits + write d_logits 8k^2, radiance r/w 8c, state + upstream grads ~ 40
    return 8 * k * k + 8 * c + 40


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["splat", "model"], default="splat")
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--ksize", type=int, default=21)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=180,
                    help="rows of the frame used for the bounded CPU-oracle sample")
    ap.add_argument("--cpu-spp", type=int, default=2)
    return ap.parse_args()


def slab_rows(h, world, rank, pad):
    """Rows [y0, y1) owned by `rank` and the haloed source range it must hold."""
    base, rem = divmod(h, world)
    y0 = rank * base + min(rank, rem)
    y1 = y0 + base + (1 if rank < rem else 0)
    return y0, y1, max(0, y0 - pad), min(h, y1 + pad)


def make_splat_inputs(h, w, spp, k, device, seed):
    g = th.Generator(device="cpu").manual_seed(seed)
    rad = [th.empty(1, 3, h, w).exponential_(1.0, generator=g).to(device) for _ in range(spp)]
    logits = []
    for _ in range(spp):
        t = th.empty(1, k * k, h, w, device=device)
        t.normal_(0, 1)
        logits.append(t.requires_grad_())
    for r in rad:
        r.requires_grad_()
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


def make_model_inputs(h, w, spp, device, seed):
    g = th.Generator(device="cpu").manual_seed(seed)
    rad = th.empty(1, spp, 3, h, w).exponential_(1.0, generator=g)
    feat = th.rand(1, spp, 93, h, w, generator=g)
    lr = th.log1p(rad) / 10.0  # radiance channels of the feature vector (datasets.py:760-768)
    feat[:, :, 5:8] = lr
    feat[:, :, 8:11] = lr
    gf = th.rand(1, 3, 1, 1, generator=g)
    tgt = th.empty(1, 3, h, w).exponential_(1.0, generator=g)
    return {"radiance": rad.to(device), "features": feat.to(device),
            "global_features": gf.to(device), "target_image": tgt.to(device)}


def cpu_baseline(args):
    """Times the CPU oracle ("port") on a bounded sample of the same workload."""
    from oracle import sbmc_oracle as orc
    orc.lib()
    h, w, k, spp = min(args.cpu_rows, args.height), args.width, args.ksize, args.cpu_spp
    threads = th.get_num_threads()
    th.manual_seed(0)
    rad = [th.empty(1, 3, h, w).exponential_(1.0).requires_grad_() for _ in range(spp)]
    logits = [th.randn(1, k * k, h, w).requires_grad_() for _ in range(spp)]
    d_out = th.randn(1, 3, h, w)

    def update(d, kk, a, b, m):
        return orc.progressive_kernel_apply(d, kk, a, b, m, splat=True)
    t0 = time.time()
    splat_step(update, rad, logits, d_out)
    dt = time.time() - t0
    return {
        "value": round(spp * h * w / dt / 1e6, 4), "unit": "Msamples/s",
        "cores": threads, "kind": "port",
        "sample": "oracle (C ops + torch-CPU composition) splat fwd+bwd on %dx%d, %d spp, k=%d, "
                  "%.1f s, host has %d logical cpus" % (w, h, spp, k, dt, os.cpu_count()),
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not th.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (MI355X)")
    th.cuda.set_device(local_rank)
    device = th.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    from sbmc_amd import _lib, functions, modules
    _lib.lib()  # fail loudly if the HIP extension is missing

    H, W, S, K = args.height, args.width, args.spp, args.ksize
    pad = (K - 1) // 2
    steps = args.steps if args.steps is not None else (10 if args.workload == "splat" else 3)
    warmup = args.warmup if args.warmup is not None else (3 if args.workload == "splat" else 1)

    y0, y1, s0, s1 = slab_rows(H, world, rank, pad)
    local_h = s1 - s0
    timings = []

    if args.workload == "splat":
        update = modules.ProgressiveKernelApply(splat=True)
        rad, logits, d_out = make_splat_inputs(local_h, W, S, K, device, seed=1234 + rank)

        def step():
            splat_step(update, rad, logits, d_out)
    else:
        if world > 1:
            raise SystemExit("model workload: multi-GPU H-slab path not wired into bench yet")
        from sbmc_amd import Multisteps, losses
        from sbmc_amd.utils import crop_like
        th.manual_seed(0)
        model = Multisteps(93, 3, ksize=K).to(device)
        model.train()
        opt = th.optim.Adam(model.parameters(), lr=1e-4)
        loss_fn = losses.TonemappedRelativeMSE()
        batch = make_model_inputs(H, W, S, device, seed=1234)

        def step():
            opt.zero_grad()
            out = model(batch)["radiance"]
            tgt = crop_like(batch["target_image"], out)
            loss = loss_fn(out, tgt)
            loss.backward()
            if not th.isfinite(loss).item():
                raise RuntimeError("non-finite loss")
            th.nn.utils.clip_grad_norm_(model.parameters(), 1000)
            opt.step()

    def sync():
        th.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            th.cuda.synchronize(device)

    for _ in range(warmup):
        step()
    sync()
    functions.enable_kernel_timing(timings)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    functions.enable_kernel_timing(None)
    if world > 1:
        t = th.tensor([dt], device=device, dtype=th.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    # per-call device time of the fused operators inside the timed region
    per = {}
    for name, a, b in timings:
        per.setdefault(name, []).append(a.elapsed_time(b))  # ms
    kern = {}
    px = local_h * W
    for name, bpp in (("splat_update_fwd", fwd_bytes_per_pixel(K)),
                      ("splat_update_bwd", bwd_bytes_per_pixel(K))):
        if name in per:
            avg_ms = sum(per[name]) / len(per[name])
            kern[name] = {"calls": len(per[name]), "avg_ms": round(avg_ms, 4),
                          "alg_bytes": px * bpp,
                          "GBps": round(px * bpp / (avg_ms * 1e-3) / 1e9, 1)}

    if rank == 0:
        ms = dt / steps * 1e3
        value = S * H * W / (dt / steps) / 1e6
        res = {
            "metric": "Msamples/s (SxHxW) denoise fwd+bwd, %dx%d %dspp %dx%d kernel" % (W, H, S, K, K),
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": ("splat fwd+bwd: %d x ProgressiveKernelApply(splat=True) + normalise + "
                             "backward" % S) if args.workload == "splat" else
                            "Multisteps(93,3) training step: fwd + TonemappedRelativeMSE + bwd + clip + Adam",
                "height": H, "width": W, "spp": S, "ksize": K, "batch": 1,
                "parallelism": "single GPU" if world == 1 else "H-slabs x%d (+%d halo rows/side)" % (world, pad),
            },
            "kernels": kern,
        }
        if "splat_update_bwd" in kern:
            kb = kern["splat_update_bwd"]
            res["roofline"] = {
                "kernel": "splat_update_bwd (state + main + route launches of one call)",
                "bound": "hbm", "achieved": kb["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(kb["GBps"] / HBM_PEAK_GBPS, 4), "traffic": None,
            }
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(args)
            except Exception as e:  # the baseline must never sink the GPU measurement
                res["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
