"""PNG <-> uint8 RGBA array: the part of tf.image.decode_png(image, channels=4) that the sprite loader needs, and a writer
for the preview sheets (the reference saves matplotlib figures, pix2pix_model.py:112-150).  Decoder:
(dataset_utils.py:66-69): 8-bit, non-interlaced images of colour type 0/2/3/4/6.  Chunk parsing and zlib inflate are Python
standard library; the scanline un-filtering (Paeth & co., sequential per byte) is the host function p2p_png_unfilter of the
C-ABI library -- no PIL/libpng dependency."""
import ctypes as C
import struct
import zlib

import numpy as np

from . import _lib as L

_SIGNATURE = b"\x89PNG\r\n\x1a\n"
_CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def decode_png(data):
    """bytes of a PNG file -> uint8 array (H, W, 4)"""
    if data[:8] != _SIGNATURE:
        raise ValueError("not a PNG file")
    pos, idat, plte, trns, hdr = 8, [], None, None, None
    while pos < len(data):
        length, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + length]
        pos += 12 + length
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif kind == b"tRNS":
            trns = np.frombuffer(body, np.uint8)
        elif kind == b"IEND":
            break
    if hdr is None or not idat:
        raise ValueError("PNG without IHDR/IDAT")
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or interlace != 0 or ctype not in _CHANNELS:
        raise ValueError(f"unsupported PNG (bit depth {depth}, colour type {ctype}, interlace {interlace}): 8-bit non-interlaced only")
    bpp = _CHANNELS[ctype]
    raw = zlib.decompress(b"".join(idat))
    if len(raw) != h * (1 + w * bpp):
        raise ValueError("PNG data size does not match its header")
    out = np.empty((h, w * bpp), np.uint8)
    src = np.frombuffer(raw, np.uint8)
    L.call("p2p_png_unfilter", C.c_void_p(src.ctypes.data), h, w * bpp, bpp, C.c_void_p(out.ctypes.data))
    px = out.reshape(h, w, bpp)
    rgba = np.empty((h, w, 4), np.uint8)
    if ctype == 6:
        rgba[:] = px
    elif ctype == 2:
        rgba[..., :3], rgba[..., 3] = px, 255
        if trns is not None and len(trns) >= 6:      # one fully transparent colour (16-bit samples, low byte used)
            key = trns[1:6:2]
            rgba[..., 3][(px == key).all(-1)] = 0
    elif ctype == 0:
        rgba[..., :3], rgba[..., 3] = px, 255
    elif ctype == 4:
        rgba[..., :3], rgba[..., 3] = px[..., :1], px[..., 1]
    else:                                            # palette
        if plte is None:
            raise ValueError("palette PNG without PLTE")
        alpha = np.full(len(plte), 255, np.uint8)
        if trns is not None:
            alpha[:len(trns)] = trns[:len(plte)]
        rgba[..., :3], rgba[..., 3] = plte[px[..., 0]], alpha[px[..., 0]]
    return rgba


def encode_png(rgba):
    """uint8 array (H, W, 4) -> bytes of an 8-bit RGBA PNG (scanline filter "Up", which suits sprites and previews)"""
    rgba = np.ascontiguousarray(rgba, np.uint8)
    if rgba.ndim != 3 or rgba.shape[2] != 4:
        raise ValueError("encode_png takes an (H, W, 4) uint8 array")
    h, w, _ = rgba.shape
    rows = rgba.reshape(h, w * 4)
    up = np.zeros_like(rows)
    up[1:] = rows[:-1]
    filt = np.empty((h, 1 + w * 4), np.uint8)
    filt[:, 0] = 2
    filt[:, 1:] = rows - up            # uint8 arithmetic wraps modulo 256, as the filter is defined

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)
    return (_SIGNATURE + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(filt.tobytes(), 6)) + chunk(b"IEND", b""))


def write_png(path, rgba):
    with open(path, "wb") as f:
        f.write(encode_png(rgba))
    return path


def read_png(path):
    with open(path, "rb") as f:
        return decode_png(f.read())
