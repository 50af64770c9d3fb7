"""Reader / writer of the SBMC `.bin` sample-tile format.

The format is produced by the reference's patched PBRT (pbrt_patches/sbmc_pbrt.diff:6232-6316)
and parsed by `sbmc/datasets.py` (`_read_globals_and_meta` :501-550, `_read_compressed`
:570-579, `_read_data` :581-739).  One file = one `tile_size` x `tile_size` tile of one scene:

    int32 x9   version (20181212 | 20190401), tile_size, image_width, image_height,
               sample_count, gt_sample_count, sample_features (27), pixel_features (30),
               path_depth (6)
    float32 x4 focus_distance, aperture_radius, fov, scene_radius
    int32 x2   block_x, block_y                       (pixel offset of the tile in the frame)
    block      pixel data:  [pixel_features, ts, ts] float32   (15 means then 15 variances)
    block x sample_count, each:
               [27, ts, ts] float32  base sample features (5 coords, 6 radiance, 16 g-buffer)
               [24, ts, ts] float32  sampling probabilities (4 per path vertex)
               [12, ts, ts] float32  light directions (2 per path vertex)
               [ 6, ts, ts] int16    bounce-type bit flags per path vertex
                                     (bit0 reflection, 1 transmission, 2 diffuse, 3 glossy, 4 specular)
    where block = int32 nbytes + an LZ4 *frame* of that many bytes.

`read_tile` returns what the reference's `TilesDataset.__getitem__` returns (93 features in "sbmc"
mode with every feature group enabled, datasets.py:309-354 and `_preprocess_standard` :741-778; fewer
with the reference's `load_coords / load_gbuffer / load_p / load_ld / load_bt` flags off, :194-215,
:706-717); `read_scene` assembles a frame like `FullImagesDataset.__getitem__` (:930-964).
LZ4 framing goes through the system `liblz4.so.1` via ctypes (no Python lz4 in this image).
"""
import ctypes
import os
import struct

import numpy as np

VERSIONS = (20181212, 20190401)
PATH_DEPTH = 6
SAMPLE_FEATURES = 27
PIXEL_FEATURES = 30
N_BT = 5
NUM_FEATURES = SAMPLE_FEATURES + 4 * PATH_DEPTH + 2 * PATH_DEPTH + N_BT * PATH_DEPTH  # 93
GLOBAL_LABELS = ("aperture_radius", "focus_distance", "fov")   # datasets.py:312
I_DIFFUSE, I_SPECULAR = 5, 8                                   # datasets.py:319-323
I_NORMAL, I_DEPTH, I_ALBEDO = 14, 18, 24                       # g-buffer labels, datasets.py:326-335
C_DIFFUSE, C_SPECULAR, C_ALBEDO = 0, 3, 6                      # pixel-data channels, datasets.py:300-306
MODES = ("sbmc", "kpcn", "raw")
FEATURE_FLAGS = ("load_coords", "load_gbuffer", "load_p", "load_ld", "load_bt")


def feature_flags(mode="sbmc", load_coords=True, load_gbuffer=True, load_p=True, load_ld=True, load_bt=True):
    """The five feature-group switches as the reference resolves them (datasets.py:194-215): the "raw" and
    "kpcn" modes always read radiance + g-buffer only."""
    if mode != "sbmc":
        return dict(load_coords=False, load_gbuffer=True, load_p=False, load_ld=False, load_bt=False)
    return dict(load_coords=bool(load_coords), load_gbuffer=bool(load_gbuffer), load_p=bool(load_p),
                load_ld=bool(load_ld), load_bt=bool(load_bt))


def feature_labels(load_coords=True, load_gbuffer=True, load_p=True, load_ld=True, load_bt=True):
    """Names of the per-sample feature channels for a choice of feature groups (datasets.py:309-354)."""
    labels = []
    if load_coords:
        labels += ["dx", "dy", "lens_u", "lens_v", "t"]
    labels += ["diffuse_r", "diffuse_g", "diffuse_b", "specular_r", "specular_g", "specular_b"]
    if load_gbuffer:
        labels += ["normal_first_x", "normal_first_y", "normal_first_z", "normal_x", "normal_y", "normal_z",
                   "depth_first", "depth", "visibility", "hasHit",
                   "albedo_first_r", "albedo_first_g", "albedo_first_b", "albedo_r", "albedo_g", "albedo_b"]
    if load_p:
        labels += ["p"] * (PATH_DEPTH * 4)
    if load_ld:
        for i in range(PATH_DEPTH):
            labels += ["ld_theta_%d" % i, "ld_phi_%d" % i]
    if load_bt:
        for txt in ("reflection", "transmisson", "diffuse", "glossy", "specular"):
            labels += ["bt_%s_%d" % (txt, i) for i in range(PATH_DEPTH)]
    return labels


def _kept_channels(flags):
    """Indices, in the file's 93-channel order, of the groups that are switched on (datasets.py:706-717; the
    probabilities, light directions and bounce types the reference does not even copy, :646-701)."""
    keep = list(range(0, 5)) if flags["load_coords"] else []
    keep += list(range(5, 11))                                 # radiance: always
    if flags["load_gbuffer"]:
        keep += list(range(11, SAMPLE_FEATURES))
    o = SAMPLE_FEATURES
    for on, n in ((flags["load_p"], 4 * PATH_DEPTH), (flags["load_ld"], 2 * PATH_DEPTH),
                  (flags["load_bt"], N_BT * PATH_DEPTH)):
        if on:
            keep += list(range(o, o + n))
        o += n
    return keep

_LZ4 = None


def _lz4():
    global _LZ4
    if _LZ4 is None:
        lib = ctypes.CDLL("liblz4.so.1")
        lib.LZ4F_compressFrameBound.restype = ctypes.c_size_t
        lib.LZ4F_compressFrameBound.argtypes = [ctypes.c_size_t, ctypes.c_void_p]
        lib.LZ4F_compressFrame.restype = ctypes.c_size_t
        lib.LZ4F_compressFrame.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                           ctypes.c_size_t, ctypes.c_void_p]
        lib.LZ4F_isError.restype = ctypes.c_uint
        lib.LZ4F_isError.argtypes = [ctypes.c_size_t]
        lib.LZ4F_createDecompressionContext.restype = ctypes.c_size_t
        lib.LZ4F_createDecompressionContext.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        lib.LZ4F_freeDecompressionContext.argtypes = [ctypes.c_void_p]
        lib.LZ4F_decompress.restype = ctypes.c_size_t
        lib.LZ4F_decompress.argtypes = [ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p,
                                        ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p]
        _LZ4 = lib
    return _LZ4


def lz4f_compress(data):
    lib = _lz4()
    data = bytes(data)
    bound = lib.LZ4F_compressFrameBound(len(data), None)
    dst = ctypes.create_string_buffer(bound)
    n = lib.LZ4F_compressFrame(dst, bound, data, len(data), None)
    if lib.LZ4F_isError(n):
        raise RuntimeError("LZ4F_compressFrame failed")
    return dst.raw[:n]


def lz4f_decompress(buf, expected_size=None):
    """Decompresses one LZ4 frame; checks the size when `expected_size` is given."""
    lib = _lz4()
    buf = bytes(buf)
    ctx = ctypes.c_void_p()
    if lib.LZ4F_isError(lib.LZ4F_createDecompressionContext(ctypes.byref(ctx), 100)):
        raise RuntimeError("LZ4F_createDecompressionContext failed")
    try:
        chunk = ctypes.create_string_buffer(1 << 20)
        parts, total, src_off = [], 0, 0
        while True:
            dst_sz = ctypes.c_size_t(len(chunk))
            src_sz = ctypes.c_size_t(len(buf) - src_off)
            rc = lib.LZ4F_decompress(ctx, chunk, ctypes.byref(dst_sz),
                                     ctypes.c_char_p(buf[src_off:]), ctypes.byref(src_sz), None)
            if lib.LZ4F_isError(rc):
                raise RuntimeError("LZ4F_decompress failed (corrupt block?)")
            src_off += src_sz.value
            if dst_sz.value:
                parts.append(chunk.raw[:dst_sz.value])
                total += dst_sz.value
            if rc == 0:
                break
            if src_sz.value == 0 and dst_sz.value == 0:
                raise RuntimeError("truncated LZ4 frame")
            if expected_size is not None and total > expected_size:
                break
        if expected_size is not None and total != expected_size:
            raise RuntimeError("LZ4 frame holds %d bytes, expected %d" % (total, expected_size))
        return b"".join(parts)
    finally:
        lib.LZ4F_freeDecompressionContext(ctx)


def _write_block(fid, payload):
    frame = lz4f_compress(payload)
    fid.write(struct.pack("i", len(frame)))
    fid.write(frame)


def _read_block(fid, expected_size):
    raw = fid.read(4)
    if len(raw) != 4:
        raise RuntimeError("truncated .bin file")
    nbytes = struct.unpack("i", raw)[0]
    buf = fid.read(nbytes)
    if nbytes < 0 or len(buf) != nbytes:
        raise RuntimeError("truncated .bin file")
    return lz4f_decompress(buf, expected_size)


def write_tile(path, block_x, block_y, image_width, image_height, pixel_data, base, probabilities,
               light_dirs, bounce_flags, focus_distance=0.0, aperture_radius=0.0, fov=35.0,
               scene_radius=1.0, gt_sample_count=1, version=20190401):
    """Writes one tile.

    pixel_data [30, ts, ts] f32; base [spp, 27, ts, ts] f32; probabilities [spp, 24, ts, ts] f32;
    light_dirs [spp, 12, ts, ts] f32; bounce_flags [spp, 6, ts, ts] int16.
    """
    spp, ts = base.shape[0], base.shape[-1]
    assert pixel_data.shape == (PIXEL_FEATURES, ts, ts)
    assert base.shape == (spp, SAMPLE_FEATURES, ts, ts)
    assert probabilities.shape == (spp, 4 * PATH_DEPTH, ts, ts)
    assert light_dirs.shape == (spp, 2 * PATH_DEPTH, ts, ts)
    assert bounce_flags.shape == (spp, PATH_DEPTH, ts, ts)
    with open(path, "wb") as fid:
        fid.write(struct.pack("9i", version, ts, image_width, image_height, spp, gt_sample_count,
                              SAMPLE_FEATURES, PIXEL_FEATURES, PATH_DEPTH))
        fid.write(struct.pack("4f", focus_distance, aperture_radius, fov, scene_radius))
        fid.write(struct.pack("2i", block_x, block_y))
        _write_block(fid, np.ascontiguousarray(pixel_data, np.float32).tobytes())
        for s in range(spp):
            _write_block(fid, b"".join((
                np.ascontiguousarray(base[s], np.float32).tobytes(),
                np.ascontiguousarray(probabilities[s], np.float32).tobytes(),
                np.ascontiguousarray(light_dirs[s], np.float32).tobytes(),
                np.ascontiguousarray(bounce_flags[s], np.int16).tobytes())))


def read_header(fid):
    raw = fid.read(36 + 16)
    if len(raw) != 52:
        raise RuntimeError("truncated .bin header")
    (version, ts, width, height, sample_count, gt_count, sample_features, pixel_features,
     path_depth) = struct.unpack("9i", raw[:36])
    focus, aperture, fov, scene_radius = struct.unpack("4f", raw[36:])
    if version not in VERSIONS:
        raise ValueError("Version unsupported: got %s, valid are %s" % (version, VERSIONS))
    if path_depth != PATH_DEPTH:
        raise RuntimeError("Incorrect path depth in the data")
    if sample_features != SAMPLE_FEATURES or pixel_features != PIXEL_FEATURES:
        raise RuntimeError("unexpected feature counts %d / %d" % (sample_features, pixel_features))
    if aperture == 0:
        focus = 0.0  # datasets.py:527-528 (unset focus distance is NaN without depth of field)
    if focus < 0 or aperture < 0 or fov < 0 or scene_radius < 0:
        raise RuntimeError("corrupt global features")
    return dict(version=version, tile_size=ts, image_width=width, image_height=height,
                sample_count=sample_count, gt_sample_count=gt_count, focus_distance=focus,
                aperture_radius=aperture, fov=fov, scene_radius=scene_radius)


def _gradients(buf):
    """[c, h, w] -> [2c, h, w]: backward differences along x then y, zero in the first column / row
    (datasets.py:858-874)."""
    dx = np.pad(buf[:, :, 1:] - buf[:, :, :-1], [[0, 0], [0, 0], [1, 0]], mode="constant")
    dy = np.pad(buf[:, 1:] - buf[:, :-1], [[0, 0], [1, 0], [0, 0]], mode="constant")
    return np.concatenate([dx, dy], 0)


def preprocess_kpcn(tile):
    """A raw tile (read_tile(..., mode="raw")) in the format [Bako2017]'s denoiser expects: per-pixel
    means / variances of the samples, albedo-demodulated diffuse, log specular and their gradients
    (the reference's TilesDataset._preprocess_kpcn, datasets.py:780-856).  27 input channels per branch."""
    f, tgt = tile["features"], tile["image_data"]
    labels = tile["labels"]
    I_DEPTH, I_ALBEDO, I_NORMAL = labels.index("depth"), labels.index("albedo_r"), labels.index("normal_x")
    I_DIFFUSE, I_SPECULAR = labels.index("diffuse_r"), labels.index("specular_r")
    spp = f.shape[0]
    depth = f[:, I_DEPTH:I_DEPTH + 1].mean(0)
    depth_v = f[:, I_DEPTH:I_DEPTH + 1].var(0)
    max_depth = depth.max()
    if max_depth > 0:
        depth = depth / max_depth
        depth_v = depth_v / (max_depth * max_depth * spp)
    depth = np.clip(depth, 0, 1)

    def stats(i):
        return f[:, i:i + 3].mean(0), f[:, i:i + 3].var(0).mean(0, keepdims=True) / spp
    albedo, albedo_v = stats(I_ALBEDO)
    albedo = albedo + 0.00316
    albedo_sqr = (albedo * albedo).mean(0, keepdims=True)
    diffuse, diffuse_v = stats(I_DIFFUSE)
    diffuse = np.maximum(diffuse, 0)
    specular, specular_v = stats(I_SPECULAR)
    specular = np.maximum(specular, 0)
    diffuse = diffuse / albedo
    diffuse_v = diffuse_v / albedo_sqr
    specular = np.log(1 + specular)
    specular_v = specular_v / (((1 + specular) * (1 + specular)).mean(0, keepdims=True) + 1e-5)
    normals, normals_v = stats(I_NORMAL)
    normals_g, depth_g, albedo_g = _gradients(normals), _gradients(depth), _gradients(albedo)
    common = [normals_g, normals_v, depth_g, depth_v, albedo_g, albedo_v]
    out = {
        "kpcn_diffuse_in": np.concatenate([diffuse] + common + [_gradients(diffuse), diffuse_v], 0),
        "kpcn_specular_in": np.concatenate([specular] + common + [_gradients(specular), specular_v], 0),
        "kpcn_diffuse_buffer": diffuse, "kpcn_specular_buffer": specular, "kpcn_albedo": albedo,
    }
    for k in ("target_image", "low_spp", "spp", "block_x", "block_y", "header", "global_features",
              "scene_radius", "path"):
        if k in tile:
            out[k] = tile[k]
    return out


def read_tile(path, spp=None, preprocess=True, mode="sbmc", **flags):
    """One tile as the reference's TilesDataset yields it (datasets.py:395-411).

    mode "sbmc": a dict with block_x, block_y, global_features [3,1,1], image_data [15,ts,ts],
    image_data_var, target_image [3,ts,ts], features [spp,nf,ts,ts] (nf = 93 with every feature group on;
    flags: load_coords, load_gbuffer, load_p, load_ld, load_bt as in the reference, datasets.py:194-215),
    labels (the nf channel names), radiance [spp,3,ts,ts], low_spp [3,ts,ts], spp, scene_radius, header; the
    radiance features log-compressed (`_preprocess_standard`) unless preprocess=False.  mode "raw": radiance +
    g-buffer (22 channels), not preprocessed; mode "kpcn": `preprocess_kpcn` of that.
    """
    if mode not in MODES:
        raise RuntimeError("Unknown dataset loading mode %s" % mode)
    unknown = set(flags) - set(FEATURE_FLAGS)
    if unknown:
        raise TypeError("read_tile: unknown arguments %s" % sorted(unknown))
    flags = feature_flags(mode, **flags)
    if mode != "sbmc":
        preprocess = False
    with open(path, "rb") as fid:
        hdr = read_header(fid)
        ts = hdr["tile_size"]
        n = hdr["sample_count"] if spp is None else spp
        if n > hdr["sample_count"]:
            raise RuntimeError("Requested too many samples.")
        bx, by = struct.unpack("2i", fid.read(8))
        out = dict(header=hdr, block_x=bx, block_y=by, scene_radius=hdr["scene_radius"], path=path)
        out["global_features"] = np.array([hdr[k] for k in GLOBAL_LABELS], np.float32).reshape(3, 1, 1)
        img = np.frombuffer(_read_block(fid, PIXEL_FEATURES * ts * ts * 4), np.float32)
        img = img.reshape(PIXEL_FEATURES, ts, ts)
        half = PIXEL_FEATURES // 2
        out["image_data"], out["image_data_var"] = img[:half], img[half:]
        out["target_image"] = img[:3] + img[3:6]              # diffuse + specular
        out["spp"] = n * np.ones((1, 1, 1), np.int32)
        px = ts * ts
        fsz = (SAMPLE_FEATURES + 4 * PATH_DEPTH + 2 * PATH_DEPTH) * px * 4
        feats = np.zeros((n, NUM_FEATURES, ts, ts), np.float32)
        for s in range(n):
            buf = _read_block(fid, fsz + PATH_DEPTH * px * 2)
            nfl = SAMPLE_FEATURES + 6 * PATH_DEPTH
            feats[s, :nfl] = np.frombuffer(buf[:fsz], np.float32).reshape(nfl, ts, ts)
            bits = np.frombuffer(buf[fsz:], np.int16).reshape(PATH_DEPTH, ts, ts)
            for b in range(N_BT):                             # datasets.py:682-703
                feats[s, nfl + b * PATH_DEPTH: nfl + (b + 1) * PATH_DEPTH] = (bits & (1 << b)) != 0
    labels = feature_labels(**flags)
    keep = _kept_channels(flags)
    if len(keep) != NUM_FEATURES:
        feats = np.ascontiguousarray(feats[:, keep])
    i_d, i_s = labels.index("diffuse_r"), labels.index("specular_r")
    if n > 0:
        out["radiance"] = feats[:, i_d:i_d + 3] + feats[:, i_s:i_s + 3]
        out["low_spp"] = out["radiance"].mean(0)
    else:
        out["low_spp"] = np.zeros_like(out["target_image"])
    if preprocess and n > 0:                                  # _preprocess_standard, :741-778
        diffuse = np.maximum(feats[:, i_d:i_d + 3], 0)
        specular = np.maximum(feats[:, i_s:i_s + 3], 0)
        feats[:, i_d:i_d + 3] = np.log(1 + diffuse + specular) / 10.0
        feats[:, i_s:i_s + 3] = np.log(1 + specular) / 10.0
    out["features"] = feats
    out["labels"] = labels
    return preprocess_kpcn(out) if mode == "kpcn" else out


def read_scene(folder, spp=None, mode="sbmc", **flags):
    """All tiles of one scene folder assembled into full-frame arrays (FullImagesDataset,
    datasets.py:930-964: every tile is preprocessed on its own, then pasted).  flags: the feature-group
    switches of `read_tile`."""
    files = sorted(f for f in os.listdir(folder) if f.endswith(".bin"))
    if not files:
        raise RuntimeError("Empty dataset")
    first = read_tile(os.path.join(folder, files[0]), spp, mode=mode, **flags)
    hdr = first["header"]
    ts, w, h = hdr["tile_size"], hdr["image_width"], hdr["image_height"]
    keys = [k for k, v in first.items() if isinstance(v, np.ndarray) and v.ndim >= 3
            and v.shape[-1] == ts and k not in ("global_features", "spp")]
    frame = {k: np.zeros(first[k].shape[:-2] + (h, w), first[k].dtype) for k in keys}
    frame["global_features"] = first["global_features"]
    frame["scene_radius"] = first["scene_radius"]
    frame["header"] = hdr
    if "labels" in first:
        frame["labels"] = first["labels"]
    for f in files:
        tile = first if f == files[0] else read_tile(os.path.join(folder, f), spp, mode=mode, **flags)
        th_ = tile["header"]
        for k in ("version", "tile_size", "image_width", "image_height", "sample_count"):
            if th_[k] != hdr[k]:
                raise ValueError("Metadata do not match.")
        bx, by = tile["block_x"], tile["block_y"]
        for k in keys:
            frame[k][..., by:by + ts, bx:bx + ts] = tile[k][..., :h - by, :w - bx]
    return frame
