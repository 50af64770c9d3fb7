"""`.bin` sample format (row N1) and the frame-level denoise harness (row a9)."""
import os
import struct
import sys

import numpy as np
import pytest
import torch as th

from helpers import GOLDEN, close, golden

sys.path.insert(0, os.path.join(GOLDEN))


def test_reader_matches_what_the_reference_reads():
    """tests/golden/bin_scene/ was written by sbmc_amd.binio; bin_scene_expected.npz is what the
    REFERENCE's sbmc/datasets.py (FullImagesDataset, 'sbmc' mode) read from it."""
    from sbmc_amd import binio
    g = golden("bin_scene_expected.npz")
    for spp in (3, 2):
        frame = binio.read_scene(os.path.join(GOLDEN, "bin_scene", "scene0"), spp=spp)
        assert frame["features"].shape == (spp, binio.NUM_FEATURES, 32, 32)
        for k in ("features", "radiance", "low_spp", "target_image", "global_features",
                  "image_data", "image_data_var"):
            np.testing.assert_array_equal(frame[k], g["spp%d.%s" % (spp, k)], err_msg=k)


@pytest.mark.parametrize("tag,flags", [("no_p_bt", dict(load_p=False, load_bt=False)),
                                       ("no_coords_gbuffer_ld", dict(load_coords=False, load_gbuffer=False, load_ld=False))])
def test_feature_groups_match_what_the_reference_reads(tag, flags):
    """Feature-group selection (reference sbmc/datasets.py:194-215, :309-354, :706-717; set by
    scripts/train.py:139-148 and handed to denoise.py through the checkpoint's data_params):
    bin_scene_groups_expected.npz is what the REFERENCE's FullImagesDataset read with these groups off."""
    from sbmc_amd import binio
    g = golden("bin_scene_groups_expected.npz")
    frame = binio.read_scene(os.path.join(GOLDEN, "bin_scene", "scene0"), spp=3, **flags)
    assert frame["features"].shape[1] == int(g[tag + ".num_features"])
    assert frame["labels"] == [str(x) for x in g[tag + ".labels"]]
    assert len(binio.feature_labels(**binio.feature_flags("sbmc", **flags))) == int(g[tag + ".num_features"])
    for k in ("features", "radiance", "low_spp", "target_image"):
        np.testing.assert_array_equal(frame[k], g["%s.%s" % (tag, k)], err_msg=k)
    with pytest.raises(TypeError):
        binio.read_tile(os.path.join(GOLDEN, "bin_scene", "scene0", "tile_000.bin"), load_everything=True)


def test_roundtrip_and_errors(tmp_path):
    from sbmc_amd import binio
    from make_golden import synthetic_scene
    scene = str(tmp_path / "scene")
    synthetic_scene(scene, 24, 16, 8, 2, seed=5)
    frame = binio.read_scene(scene)
    assert frame["features"].shape == (2, 93, 16, 24)
    raw = binio.read_tile(os.path.join(scene, "tile_000.bin"), preprocess=False)
    # bounce-type planes are 0/1, radiance = diffuse + specular of the raw samples
    assert set(np.unique(raw["features"][:, 63:])) <= {0.0, 1.0}
    np.testing.assert_allclose(raw["radiance"], raw["features"][:, 5:8] + raw["features"][:, 8:11])
    with pytest.raises(RuntimeError):
        binio.read_scene(scene, spp=5)                        # more samples than stored
    bad = tmp_path / "bad.bin"
    data = bytearray(open(os.path.join(scene, "tile_000.bin"), "rb").read())
    data[:4] = struct.pack("i", 123)
    bad.write_bytes(bytes(data))
    with pytest.raises(ValueError):
        binio.read_tile(str(bad))                             # unsupported version
    bad.write_bytes(open(os.path.join(scene, "tile_000.bin"), "rb").read()[:200])
    with pytest.raises(RuntimeError):
        binio.read_tile(str(bad))                             # truncated
    empty = tmp_path / "empty"
    empty.mkdir()
    with pytest.raises(RuntimeError):
        binio.read_scene(str(empty))


def _frame_batch(h, w, spp, nf, seed, device="cpu"):
    g = th.Generator().manual_seed(seed)
    rad = th.empty(1, spp, 3, h, w).exponential_(1.0, generator=g)
    return {"radiance": rad.to(device), "features": th.rand(1, spp, nf, h, w, generator=g).to(device),
            "global_features": th.rand(1, 3, 1, 1, generator=g).to(device),
            "low_spp": rad.mean(1).to(device)}


def test_split_tiles_covers_the_frame_once():
    from sbmc_amd import denoise
    batch = _frame_batch(70, 100, 1, 2, 0)
    tiles = denoise.split_tiles(batch, max_sz=40, pad=8)
    cover = th.zeros(70, 100)
    for part, y0, y1, x0, x1, pads in tiles:
        assert "global_features" in part                      # the reference drops it
        assert part["features"].shape[-2] <= 40 and part["features"].shape[-1] <= 40
        cover[y0:y1, x0:x1] += 1
    assert (cover == 1).all()
    assert len(denoise.split_tiles(batch, max_sz=128, pad=8)) == 1
    with pytest.raises(ValueError):
        denoise.split_tiles(batch, max_sz=16, pad=8)


def _tiled_equals_untiled(device):
    from sbmc_amd import Multisteps, denoise
    th.manual_seed(1)
    # small U-net receptive field: 1 step, so that a 28-px tile overlap covers it
    model = Multisteps(6, 3, width=8, embedding_width=8, ksize=5, nsteps=1).to(device).train(False)
    batch = _frame_batch(96, 120, 2, 6, 2, device)
    whole = denoise.denoise_frame(model, batch, tile_size=256, tile_pad=0)
    tiled = denoise.denoise_frame(model, batch, tile_size=96, tile_pad=40)
    p = 2 + 40  # outside the kernel crop and the U-net's border influence the two must agree
    # the two agree to 1e-5 of a float64 evaluation of the whole frame, or the tiled one is no further from it than twice
    # the whole one is (the convolution library picks its summation order by the tile's shape)
    from helpers import multisteps_fp64, no_worse_than
    m64 = multisteps_fp64(model, (6, 3), dict(width=8, embedding_width=8, ksize=5, nsteps=1)).train(False)
    w64 = denoise.denoise_frame(m64, {k: v.cpu().double() for k, v in batch.items()}, tile_size=256, tile_pad=0)
    no_worse_than(tiled[..., p:-p, p:-p], whole[..., p:-p, p:-p], w64[..., p:-p, p:-p], what="tiled")
    # border: the reference zero-pads the (ksize-1)/2 crop back
    assert whole[..., :2, :].abs().max().item() == 0 and whole[..., :, -2:].abs().max().item() == 0


def test_tiled_equals_untiled_cpu(cpu_ops):
    _tiled_equals_untiled("cpu")


@pytest.mark.gpu
def test_tiled_equals_untiled_gpu():
    _tiled_equals_untiled("cuda")


def _config0_batch(tmp_path, device):
    """BASELINE.json configs[0]: 64x64, 4 spp synthetic .bin, 5x5 kernel."""
    from sbmc_amd import binio
    from make_golden import synthetic_scene
    scene = str(tmp_path / "scene0")
    synthetic_scene(scene, 64, 64, 32, 4, seed=7)
    frame = binio.read_scene(scene)
    return {k: th.from_numpy(np.ascontiguousarray(frame[k])).unsqueeze(0).to(device)
            for k in ("radiance", "features", "global_features", "low_spp")}


def _config0_model(device):
    from sbmc_amd import Multisteps, binio
    th.manual_seed(8)
    return Multisteps(binio.NUM_FEATURES, 3, ksize=5, width=16, embedding_width=16).to(device).train(False)


def test_config0_bin_to_denoised_frame_cpu(cpu_ops, tmp_path):
    """configs[0] plumbing, no GPU: .bin -> reader -> Multisteps(k=5) -> tiling harness, with the
    oracle behind the operator names."""
    from sbmc_amd import denoise
    out = denoise.denoise_frame(_config0_model("cpu"), _config0_batch(tmp_path, "cpu"))
    assert out.shape == (1, 3, 64, 64) and th.isfinite(out).all()
    assert out[..., 2:-2, 2:-2].abs().sum() > 0


@pytest.mark.gpu
def test_config0_gpu_matches_cpu_oracle(cpu_ops, tmp_path):
    from sbmc_amd import denoise
    """configs[0] on the GPU against the same frame through the CPU oracle: within 1e-5 of a float64 evaluation of the model,
    or no further from it than twice the CPU-oracle run is."""
    from sbmc_amd import binio
    from helpers import multisteps_fp64, no_worse_than
    cpu_model, batch = _config0_model("cpu"), _config0_batch(tmp_path, "cpu")
    ref = denoise.denoise_frame(cpu_model, batch)
    out = denoise.denoise_frame(_config0_model("cuda"), _config0_batch(tmp_path, "cuda"))
    m64 = multisteps_fp64(cpu_model, (binio.NUM_FEATURES, 3), dict(ksize=5, width=16, embedding_width=16)).train(False)
    truth = denoise.denoise_frame(m64, {k: v.double() for k, v in batch.items()})
    no_worse_than(out, ref, truth, what="configs[0]")
