#!/usr/bin/env python
"""Denoise one scene folder of `.bin` sample tiles with a Multisteps checkpoint (counterpart of
the reference's scripts/denoise.py:96-194 for the SBMC model; same flags, int-typed tile args).

    python scripts/denoise.py --input <scene folder> --checkpoint <file.pth> --output out.npy \
        [--spp N] [--tile_size 1024] [--tile_pad 256] [--ksize 21] [--width 128]

Writes the denoised radiance as .npy ([H, W, 3] float32) and, when Pillow is present, a
clipped 8-bit .png next to it (the reference writes .exr + .png through pyexr / skimage,
neither of which exists here).
"""
import argparse
import logging
import os
import sys
import time

import numpy as np
import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbmc_amd import Multisteps, binio, denoise  # noqa: E402

LOG = logging.getLogger("denoise")


def main(args):
    start = time.time()
    if not os.path.isdir(args.input):
        raise ValueError("input {} does not exist".format(args.input))
    frame = binio.read_scene(args.input, spp=args.spp)
    LOG.info("frame %dx%d, %d spp", frame["header"]["image_width"], frame["header"]["image_height"],
             frame["features"].shape[0])
    model = Multisteps(binio.NUM_FEATURES, len(binio.GLOBAL_LABELS), ksize=args.ksize,
                       width=args.width, embedding_width=args.width)
    if args.checkpoint:
        denoise.load_checkpoint(args.checkpoint, model)
    else:
        LOG.warning("no checkpoint given: running with the seeded random initialisation")
    model.train(False)
    if not th.cuda.is_available():
        raise SystemExit("sbmc_amd runs its operators on MI355X only; no GPU is visible")
    device = th.device("cuda")
    model.to(device)
    batch = {k: th.from_numpy(np.ascontiguousarray(frame[k])).unsqueeze(0).to(device)
             for k in ("radiance", "features", "global_features", "low_spp")}
    LOG.info("setup time %.1f ms", (time.time() - start) * 1000)
    th.cuda.synchronize()
    start = time.time()
    out = denoise.denoise_frame(model, batch, args.tile_size, args.tile_pad)
    th.cuda.synchronize()
    LOG.info("denoising time %.1f ms", (time.time() - start) * 1000)
    img = out[0].cpu().numpy().transpose(1, 2, 0)
    os.makedirs(os.path.dirname(os.path.abspath(args.output)), exist_ok=True)
    np.save(args.output, img)
    try:
        from PIL import Image
        Image.fromarray((np.clip(img, 0, 1) * 255).astype(np.uint8)).save(
            os.path.splitext(args.output)[0] + ".png")
    except ImportError:
        pass


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--input", required=True, help="scene folder containing the sample .bin tiles")
    p.add_argument("--checkpoint", default=None, help="state-dict checkpoint (.pth)")
    p.add_argument("--output", required=True, help="output .npy")
    p.add_argument("--spp", type=int, default=None, help="number of samples to use as input")
    p.add_argument("--tile_size", type=int, default=1024)
    p.add_argument("--tile_pad", type=int, default=256)
    p.add_argument("--ksize", type=int, default=21)
    p.add_argument("--width", type=int, default=128)
    logging.basicConfig(level=logging.INFO)
    main(p.parse_args())
