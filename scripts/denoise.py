#!/usr/bin/env python
"""Denoise one scene folder of `.bin` sample tiles with a trained checkpoint (counterpart of the
reference's scripts/denoise.py:96-194; same flags, int-typed tile arguments).

    python scripts/denoise.py --input <scene folder> --checkpoint <file.pth | folder> --output out.exr \
        [--spp N] [--tile_size 1024] [--tile_pad 256]

Like the reference, everything about the model comes from the checkpoint's `meta`
(scripts/denoise.py:107-123): `kpcn_mode` selects [Bako2017]'s KPCN over the sample-based
`Multisteps`, `model_params` (ksize, gather, pixel; written by scripts/train.py) configure it and
`data_params["spp"]` is the default sample count.  `--ksize/--gather/--pixel/--kpcn_mode` override the
meta (needed for bare state-dict checkpoints).  Output: `<output>.exr` (float32 OpenEXR) + a clipped 8-bit
`.png` next to it, as the reference (:166-173); an `--output` ending in `.npy` saves the array instead.

Several GPUs: start it through `python -m torch.distributed.run --nproc-per-node N scripts/denoise.py ...`;
the frame is then cut into N row slabs, one per rank (tiles -> ranks: halo exchange instead of the
overlapped tiles' recomputation, sbmc_amd/dist.py), and rank 0 writes the result.
"""
import argparse
import logging
import os
import sys
import time

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbmc_amd import KPCN, Multisteps, binio, denoise, imageio  # noqa: E402

LOG = logging.getLogger("denoise")


def data_flags(meta, kpcn_mode):
    """The feature groups the checkpoint was trained on: `data_params` of its meta (reference
    scripts/denoise.py:109-112 passes them to the dataset, train.py:139-148 sets them)."""
    dp = meta.get("data_params") or {}
    return binio.feature_flags("kpcn" if kpcn_mode else "sbmc", **{k: dp[k] for k in binio.FEATURE_FLAGS if k in dp})


def build_model(meta, args):
    """The network the checkpoint was trained as (reference scripts/denoise.py:116-123 + train.py:56-70)."""
    params = dict(meta.get("model_params") or {})
    kpcn_mode = bool(meta.get("kpcn_mode", False)) if args.kpcn_mode is None else args.kpcn_mode
    ksize = args.ksize if args.ksize is not None else int(params.get("ksize", 21))
    if kpcn_mode:
        LOG.info("Using [Bako2017] denoiser.")
        return KPCN(27, ksize=ksize), True
    gather = args.gather if args.gather is not None else bool(params.get("gather", False))
    pixel = args.pixel if args.pixel is not None else bool(params.get("pixel", False))
    width = args.width if args.width is not None else int(params.get("width", 128))
    for k in ("gather", "pixel"):
        if getattr(args, k) is not None and k in params and bool(params[k]) != getattr(args, k):
            LOG.warning("--%s overrides the checkpoint's model_params[%r] = %r", k, k, params[k])
    nf = len(binio.feature_labels(**data_flags(meta, False)))      # data.num_features (datasets.py:413-419)
    return Multisteps(nf, len(binio.GLOBAL_LABELS), ksize=ksize, splat=not gather,
                      pixel=pixel, width=width, embedding_width=width), False


def save_output(path, img):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    if path.endswith(".npy"):
        np.save(path, img)
        return
    imageio.write_exr(path, img)
    png = os.path.splitext(path)[0] + ".png"
    imageio.write_png(png, (np.clip(img, 0, 1) * 255).astype(np.uint8))


def main(args):
    start = time.time()
    if not os.path.isdir(args.input):
        raise ValueError("input {} does not exist".format(args.input))
    if not th.cuda.is_available():
        raise SystemExit("sbmc_amd runs its operators on MI355X only; no GPU is visible")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(1, th.cuda.device_count())
    device = th.device("cuda", local_rank)
    th.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SBMC_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))

    meta = denoise.load_meta(args.checkpoint) if args.checkpoint else {}
    spp = args.spp if args.spp else (meta.get("data_params") or {}).get("spp")
    model, kpcn_mode = build_model(meta, args)
    LOG.info("Loading model %s", args.checkpoint)
    if args.checkpoint:
        denoise.load_checkpoint(args.checkpoint, model)
    else:
        LOG.warning("no checkpoint given: running with the seeded random initialisation")
    model.train(False)
    model.to(device)

    frame = binio.read_scene(args.input, spp=spp, mode="kpcn" if kpcn_mode else "sbmc", **data_flags(meta, kpcn_mode))
    hdr = frame["header"]
    LOG.info("Denoising input %dx%d with %s spp", hdr["image_width"], hdr["image_height"], spp or hdr["sample_count"])
    keys = denoise.TILED_KEYS + denoise.UNCHANGED_KEYS + ("low_spp",)
    sharded = world > 1 and not kpcn_mode
    if sharded:
        # only this rank's rows go to its GPU (the whole frame is 2.7 GB of features at 720p x 8 spp)
        from sbmc_amd import dist as sdist
        part = sdist.SlabPartition(hdr["image_height"], world, rank)
        batch = {k: th.from_numpy(np.ascontiguousarray(
            frame[k] if k in denoise.UNCHANGED_KEYS else frame[k][..., part.y0:part.y1, :])).unsqueeze(0).to(device)
            for k in keys if k in frame}
    else:
        batch = {k: th.from_numpy(np.ascontiguousarray(frame[k])).unsqueeze(0).to(device) for k in keys if k in frame}
    LOG.info("setup time %.1f ms", (time.time() - start) * 1000)
    th.cuda.synchronize()
    start = time.time()
    if sharded:
        out = denoise.denoise_frame_sharded(model, batch, part, slab_only=True, height=hdr["image_height"])
    else:
        out = denoise.denoise_frame(model, batch, args.tile_size, args.tile_pad, kpcn_mode=kpcn_mode)
    th.cuda.synchronize()
    LOG.info("    denoising time %.1f ms", (time.time() - start) * 1000)
    if rank == 0:
        save_output(args.output, out[0].cpu().numpy().transpose(1, 2, 0))
    if world > 1:
        dist.destroy_process_group()
    return out


def parser():
    p = argparse.ArgumentParser()
    p.add_argument("--input", required=True, help="scene folder containing the sample .bin tiles")
    p.add_argument("--checkpoint", default=None, help="checkpoint file (.pth) or folder holding it")
    p.add_argument("--output", required=True, help="output .exr (a .png preview is written next to it) or .npy")
    p.add_argument("--spp", type=int, default=None, help="number of samples to use as input")
    p.add_argument("--tile_size", type=int, default=1024)
    p.add_argument("--tile_pad", type=int, default=256)
    # overrides of the checkpoint's meta (None = take the meta / the reference's defaults)
    p.add_argument("--ksize", type=int, default=None)
    p.add_argument("--width", type=int, default=None)
    p.add_argument("--gather", action=argparse.BooleanOptionalAction, default=None)       # --gather / --no-gather
    p.add_argument("--pixel", action=argparse.BooleanOptionalAction, default=None)
    p.add_argument("--kpcn_mode", action=argparse.BooleanOptionalAction, default=None)
    return p


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO)
    main(parser().parse_args())
