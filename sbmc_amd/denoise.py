"""Frame-level inference harness: the tiling / padding / pasting logic of the reference's
`scripts/denoise.py` (:42-93, :142-165), with its two defects fixed: the
reference appends each tile once per tensor key and never forwards `global_features` to the
tiles, so any frame with a side > tile_size fails there; the intended behaviour is
implemented here and checked by tests/test_denoise_bin.py (tiled == untiled in the valid
region).  With 288 GB of HBM a 1280x720 frame needs no tiling at all (tile_size >= 1280).
"""
import torch as th

TILED_KEYS = ("radiance", "features", "kpcn_diffuse_in", "kpcn_specular_in",
              "kpcn_diffuse_buffer", "kpcn_specular_buffer", "kpcn_albedo")
UNCHANGED_KEYS = ("global_features",)


def split_tiles(batch, max_sz=1024, pad=256):
    """Cuts a frame into overlapping tiles of at most max_sz x max_sz.

    Returns a list of (tile_batch, start_y, end_y, start_x, end_x, (pad_top, pad_bottom,
    pad_left, pad_right)): the tile's valid region [start_y, end_y) x [start_x, end_x) in
    frame coordinates and the number of overlap rows / columns to strip from its output.
    """
    ref = batch["low_spp"] if "low_spp" in batch else batch["radiance"]
    h, w = ref.shape[-2:]
    if h <= max_sz and w <= max_sz:
        return [(batch, 0, h, 0, w, (0, 0, 0, 0))]
    step = max_sz - 2 * pad
    if step <= 0:
        raise ValueError("tile_size must exceed 2 * tile_pad")

    def spans(n):
        out = []
        for start in range(0, n, step):
            p0 = 0 if start == 0 else pad
            end, p1 = start + max_sz, pad
            if end >= n:
                end, p1 = n, 0
            if start + p0 < end - p1:
                out.append((start, end, p0, p1))
            if end == n:
                break
        return out

    tiles = []
    for sy, ey, py0, py1 in spans(h):
        for sx, ex, px0, px1 in spans(w):
            part = {k: batch[k] for k in UNCHANGED_KEYS if k in batch}
            for k in TILED_KEYS:
                if k in batch:
                    part[k] = batch[k][..., sy:ey, sx:ex]
            tiles.append((part, sy + py0, ey - py1, sx + px0, ex - px1, (py0, py1, px0, px1)))
    return tiles


def pad_output(part, out, kpcn_mode=False):
    """Zero-pads the model output back to the tile's size (the model crops (ksize-1)/2)."""
    src = part["kpcn_diffuse_in"] if kpcn_mode else part["features"]
    pad_h = (src.shape[-2] - out.shape[-2]) // 2
    pad_w = (src.shape[-1] - out.shape[-1]) // 2
    return th.nn.functional.pad(out, (pad_w, pad_w, pad_h, pad_h))


def denoise_frame(model, batch, tile_size=1024, tile_pad=256, kpcn_mode=False):
    """Runs `model` over a frame (tiled if needed) under no_grad; returns [bs, 3, H, W]."""
    ref = batch["low_spp"] if "low_spp" in batch else batch["radiance"].mean(1)
    out_radiance = th.zeros_like(ref)
    for part, y0, y1, x0, x1, (pt, pb, pl, pr) in split_tiles(batch, int(tile_size), int(tile_pad)):
        with th.no_grad():
            out = pad_output(part, model(part)["radiance"], kpcn_mode)
        out = out[..., pt:out.shape[-2] - pb, pl:out.shape[-1] - pr]
        out_radiance[..., y0:y1, x0:x1] = out
    return out_radiance


def denoise_frame_sharded(model, batch, part, gather=True, slab_only=False, height=None):
    """One frame on several GPUs: this rank denoises the rows [part.y0, part.y1) of the frame with
    `dist.ShardedDenoiser` (U-net halo exchange + cross-rank merge of the splat state) and -- with
    gather=True -- every rank returns the whole [bs, 3, H, W] frame (zero border of (ksize-1)/2 pixels,
    as `denoise_frame`).  `batch` holds the WHOLE frame (only the slab is moved through the network), or --
    slab_only=True, `height` = rows of the frame -- just this rank's rows.  The multi-GPU counterpart of the
    tile loop of scripts/denoise.py:142-165 (tiles -> ranks, nothing recomputed in the overlap)."""
    import torch.distributed as dist
    from . import dist as sdist
    p = (model.ksize - 1) // 2
    if slab_only:
        slab = {k: v for k, v in batch.items() if k in ("radiance", "features") + UNCHANGED_KEYS}
        h, w = int(height), batch["radiance"].shape[-1]
        if slab["radiance"].shape[-2] != part.rows:
            raise ValueError("slab_only: the batch must hold exactly this rank's %d rows" % part.rows)
    else:
        slab = {k: (v if k in UNCHANGED_KEYS else v[..., part.y0:part.y1, :].contiguous())
                for k, v in batch.items() if k in ("radiance", "features") + UNCHANGED_KEYS}
        h, w = batch["radiance"].shape[-2:]
    runner = sdist.ShardedDenoiser(model, part)
    with th.no_grad():
        out = runner(slab)["radiance"]
    runner.check()
    lo, hi = max(part.y0, p), min(part.y1, h - p)          # frame rows this rank's output covers
    assert out.shape[-2] == hi - lo
    if not gather:
        return out, lo, hi
    rows_max = -(-h // part.world) + 4
    mine = out.new_zeros(out.shape[:-2] + (rows_max, out.shape[-1]))
    mine[..., :hi - lo, :] = out
    staged = mine.is_cuda and dist.get_backend(part.group) != "nccl"
    send = mine.cpu() if staged else mine
    parts = [th.empty_like(send) for _ in range(part.world)]
    dist.all_gather(parts, send, group=part.group)
    frame = out.new_zeros(out.shape[:-2] + (h, w))
    for r in range(part.world):
        pr = sdist.SlabPartition(h, part.world, r, group=part.group)
        a, b = max(pr.y0, p), min(pr.y1, h - p)
        frame[..., a:b, p:w - p] = parts[r][..., :b - a, :].to(frame.device)
    return frame


def find_checkpoint(path):
    """`path` is a checkpoint file, or -- like the reference's --checkpoint -- a folder whose most
    recent *.pth is taken (ttools.Checkpointer.load_latest, scripts/denoise.py:107,133-134)."""
    import glob
    import os
    if os.path.isdir(path):
        files = sorted(glob.glob(os.path.join(path, "*.pth")), key=os.path.getmtime)
        if not files:
            raise RuntimeError("no checkpoint (*.pth) in %s" % path)
        return files[-1]
    return path


def load_meta(path):
    """The `meta` dict stored next to the weights ({"model_params", "data_params", "kpcn_mode"} as
    scripts/train.py writes it; reference scripts/train.py:84-86, scripts/denoise.py:107-123), {} for
    a bare state dict."""
    obj = th.load(find_checkpoint(path), map_location="cpu")
    if isinstance(obj, dict) and "model" in obj and isinstance(obj["model"], dict):
        return obj.get("meta", {}) or {}
    return {}


def load_checkpoint(path, model):
    """Loads a state-dict checkpoint: either a bare state dict or {"model": sd, "meta": {...}}.
    (The reference goes through ttools.Checkpointer, whose on-disk format is not in its tree.)"""
    obj = th.load(find_checkpoint(path), map_location="cpu")
    meta = {}
    if isinstance(obj, dict) and "model" in obj and isinstance(obj["model"], dict):
        meta = obj.get("meta", {}) or {}
        obj = obj["model"]
    model.load_state_dict(obj)
    return meta
