"""Minimal image writers for the denoiser's outputs: OpenEXR (float32 HDR) and PNG (8-bit preview).

The reference's scripts/denoise.py:166-173 writes `pyexr.write(out.exr, img)` and
`skimage.io.imsave(out.png, clip(img) * 255)`; neither package exists in this image, and the two
file formats are simple enough to emit directly:

* EXR: single-part scanline file, float32 channels, no compression (readable by every OpenEXR
  implementation); `read_exr` reads back exactly what `write_exr` writes (tests).
* PNG: 8-bit RGB / grey, zlib-compressed, filter type 0.
"""
import struct
import zlib

import numpy as np

__all__ = ["write_exr", "read_exr", "write_png"]

_EXR_MAGIC = 20000630


def _attr(name, kind, payload):
    return name.encode() + b"\0" + kind.encode() + b"\0" + struct.pack("<i", len(payload)) + payload


def write_exr(path, img):
    """img [h, w, c] (c = 1: Y, 3: RGB, 4: RGBA) float -> uncompressed float32 scanline OpenEXR."""
    img = np.asarray(img, dtype=np.float32)
    if img.ndim == 2:
        img = img[..., None]
    h, w, c = img.shape
    names = {1: ["Y"], 3: ["R", "G", "B"], 4: ["R", "G", "B", "A"]}.get(c)
    if names is None:
        raise ValueError("write_exr: 1, 3 or 4 channels expected, got %d" % c)
    order = sorted(range(c), key=lambda i: names[i])          # channels are stored in alphabetical order
    chlist = b"".join(names[i].encode() + b"\0" + struct.pack("<iBBBBii", 2, 0, 0, 0, 0, 1, 1) for i in order) + b"\0"
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    header = b"".join([
        _attr("channels", "chlist", chlist),
        _attr("compression", "compression", b"\0"),
        _attr("dataWindow", "box2i", box),
        _attr("displayWindow", "box2i", box),
        _attr("lineOrder", "lineOrder", b"\0"),
        _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)),
        _attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0)),
        _attr("screenWindowWidth", "float", struct.pack("<f", 1.0)),
    ]) + b"\0"
    head = struct.pack("<ii", _EXR_MAGIC, 2) + header
    line_bytes = c * w * 4
    table_at = len(head)
    first = table_at + 8 * h
    offsets = struct.pack("<%dQ" % h, *[first + y * (8 + line_bytes) for y in range(h)])
    planar = np.ascontiguousarray(img[:, :, order].transpose(0, 2, 1)).astype("<f4")   # [h, c, w]
    with open(path, "wb") as f:
        f.write(head)
        f.write(offsets)
        for y in range(h):
            f.write(struct.pack("<ii", y, line_bytes))
            f.write(planar[y].tobytes())


def read_exr(path):
    """Reads back a file written by `write_exr` (uncompressed float32 scanlines) -> [h, w, c] float32."""
    buf = open(path, "rb").read()
    magic, version = struct.unpack_from("<ii", buf, 0)
    if magic != _EXR_MAGIC or (version & 0xFF) != 2:
        raise ValueError("%s is not an OpenEXR file" % path)
    pos, attrs = 8, {}
    while buf[pos] != 0:
        end = buf.index(b"\0", pos)
        name = buf[pos:end].decode()
        pos = end + 1
        end = buf.index(b"\0", pos)
        pos = end + 1
        size = struct.unpack_from("<i", buf, pos)[0]
        attrs[name] = buf[pos + 4:pos + 4 + size]
        pos += 4 + size
    pos += 1
    if attrs["compression"] != b"\0":
        raise ValueError("read_exr only reads uncompressed files")
    names, p = [], 0
    ch = attrs["channels"]
    while ch[p] != 0:
        end = ch.index(b"\0", p)
        names.append(ch[p:end].decode())
        if struct.unpack_from("<i", ch, end + 1)[0] != 2:
            raise ValueError("read_exr only reads float32 channels")
        p = end + 1 + 16
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"])
    w, h, c = x1 - x0 + 1, y1 - y0 + 1, len(names)
    offsets = struct.unpack_from("<%dQ" % h, buf, pos)
    out = np.empty((h, c, w), np.float32)
    for y in range(h):
        yy, nbytes = struct.unpack_from("<ii", buf, offsets[y])
        out[yy - y0] = np.frombuffer(buf, "<f4", c * w, offsets[y] + 8).reshape(c, w)
    want = {1: ["Y"], 3: ["R", "G", "B"], 4: ["R", "G", "B", "A"]}[c]
    return np.ascontiguousarray(out[:, [names.index(n) for n in want]].transpose(0, 2, 1))


def write_png(path, img):
    """img [h, w] or [h, w, 1|3] uint8 -> PNG."""
    img = np.asarray(img)
    if img.dtype != np.uint8:
        raise ValueError("write_png expects uint8")
    if img.ndim == 2:
        img = img[..., None]
    h, w, c = img.shape
    if c not in (1, 3):
        raise ValueError("write_png: 1 or 3 channels expected")

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    raw = b"".join(b"\0" + np.ascontiguousarray(img[y]).tobytes() for y in range(h))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2 if c == 3 else 0, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(chunk(b"IEND", b""))
