#!/usr/bin/env python
"""Train the SBMC denoiser on a folder of `.bin` tiles (counterpart of the reference's
scripts/train.py:33-152 for the sample-based model; visdom / progress-bar callbacks dropped).

    python scripts/train.py --data <root> --checkpoint_dir <dir> [--val_data <root>] [--spp 8]
        [--ksize 21] [--gather] [--pixel] [--kpcn_mode] [--lr 1e-4] [--bs 1] [--num_epochs 1]
"""
import argparse
import logging
import os
import sys

import numpy as np
import torch as th
from torch.utils.data import DataLoader

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbmc_amd import KPCN, Multisteps, interfaces  # noqa: E402


def main(args):
    """-> the run's record: {"history": per-step {"loss", "rmse"}, "validation": per-epoch running means,
    "start_epoch": the epoch a found checkpoint resumed from}."""
    if not th.cuda.is_available():
        raise SystemExit("sbmc_amd runs its operators on MI355X only; no GPU is visible")
    return run(args, cuda=True)


def run(args, cuda=True):
    """The body of main().  cuda=False exists for the host tests only: the `*_cpu_float32` operators raise unless a
    test has installed an implementation behind them (sbmc_amd/halide_ops.py)."""
    np.random.seed(0)
    th.manual_seed(0)
    mode = "kpcn" if args.kpcn_mode else "sbmc"          # reference scripts/train.py:39-43
    data_args = dict(spp=args.spp, mode=mode, load_coords=args.load_coords, load_gbuffer=args.load_gbuffer,
                     load_p=args.load_p, load_ld=args.load_ld, load_bt=args.load_bt)
    if args.randomize_spp:
        if args.bs != 1:
            raise RuntimeError("Training with randomized spp is only valid for batch_size=1")
        data = interfaces.MultiSampleCountDataset(args.data, **data_args)
    else:
        data = interfaces.TilesDataset(args.data, **data_args)
    if args.kpcn_mode:                                    # reference scripts/train.py:58-60
        model = KPCN(data.num_features, ksize=args.ksize)
    else:
        model = Multisteps(data.num_features, data.num_global_features, ksize=args.ksize,
                           splat=not args.gather, pixel=args.pixel)
    loader = DataLoader(data, batch_size=args.bs, num_workers=args.num_worker_threads, shuffle=True)
    val_loader = None
    if args.val_data:
        val_loader = DataLoader(interfaces.TilesDataset(args.val_data, **data_args),
                                batch_size=args.bs, num_workers=1, shuffle=False)
    meta = dict(model_params=dict(ksize=args.ksize, gather=args.gather, pixel=args.pixel),
                kpcn_mode=args.kpcn_mode, data_params=data_args)   # (the reference stores them all: train.py:85-87)
    interface = interfaces.SampleBasedDenoiserInterface(model, lr=args.lr, cuda=cuda)
    ckpt = interfaces.Checkpointer(args.checkpoint_dir, model, interface.optimizer, meta=meta)
    extras, _ = ckpt.load_latest()
    start = extras["epoch"] if extras else 0
    validation = []
    history = interfaces.train(interface, loader, num_epochs=args.num_epochs, val_dataloader=val_loader,
                               checkpointer=ckpt, start_epoch=start, validation_log=validation)
    return {"history": history, "validation": validation, "start_epoch": start}


def parser():
    p = argparse.ArgumentParser()
    p.add_argument("--data", required=True)
    p.add_argument("--val_data", default=None)
    p.add_argument("--checkpoint_dir", required=True)
    p.add_argument("--spp", type=int, default=8)
    p.add_argument("--ksize", type=int, default=21)
    p.add_argument("--gather", action="store_true")
    p.add_argument("--pixel", action="store_true")
    p.add_argument("--kpcn_mode", action="store_true", help="train [Bako2017]'s KPCN baseline instead")
    p.add_argument("--constant_spp", dest="randomize_spp", action="store_false", default=True)
    # feature groups (reference scripts/train.py:139-148)
    p.add_argument("--dont_use_coords", dest="load_coords", action="store_false", default=True)
    p.add_argument("--dont_use_gbuffer", dest="load_gbuffer", action="store_false", default=True)
    p.add_argument("--dont_use_p", dest="load_p", action="store_false", default=True)
    p.add_argument("--dont_use_ld", dest="load_ld", action="store_false", default=True)
    p.add_argument("--dont_use_bt", dest="load_bt", action="store_false", default=True)
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--bs", type=int, default=1)
    p.add_argument("--num_epochs", type=int, default=1)
    p.add_argument("--num_worker_threads", type=int, default=0)
    return p


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO)
    main(parser().parse_args())
