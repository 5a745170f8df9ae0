"""Synthetic sprite files for the input-pipeline tests: a small PNG *writer* (every scanline filter type is exercised) and a
dataset folder in the reference's layout (datasets/rpg-maker-xp/{train,test}/{0-back,..}/<n>.png, configuration.py:6-13)."""
import os
import struct
import zlib

import numpy as np

from palette_and_histo_gan_amd.configuration import DIRECTION_FOLDERS


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def encode_png(rgba, filters=(0, 1, 2, 3, 4)):
    """uint8 (H, W, 4) -> PNG bytes (colour type 6, 8 bit); row y uses filter filters[y % len(filters)]"""
    h, w, _ = rgba.shape
    rows = rgba.reshape(h, w * 4).astype(np.int32)
    raw = bytearray()
    for y in range(h):
        f = filters[y % len(filters)]
        cur, up = rows[y], rows[y - 1] if y else np.zeros(w * 4, np.int32)
        out = np.empty(w * 4, np.int32)
        for i in range(w * 4):
            a = cur[i - 4] if i >= 4 else 0
            b = up[i]
            c = up[i - 4] if i >= 4 else 0
            pred = [0, a, b, (a + b) >> 1, _paeth(int(a), int(b), int(c))][f]
            out[i] = (cur[i] - pred) & 255
        raw.append(f)
        raw += bytes(out.astype(np.uint8))

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(bytes(raw), 6)) + chunk(b"IEND", b""))


def synthetic_sprite(rng, size, palette):
    """a blob of palette colours on a transparent background; some transparent pixels keep a NON-black colour (the case
    blacken_transparent_pixels exists for, dataset_utils.py:7-10)"""
    yy, xx = np.mgrid[0:size, 0:size]
    cy, cx, r = rng.uniform(size * 0.35, size * 0.65, 2).tolist() + [rng.uniform(size * 0.2, size * 0.4)]
    inside = (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
    img = np.zeros((size, size, 4), np.uint8)
    img[inside] = palette[rng.integers(0, len(palette), size=int(inside.sum()))]
    dirty = (~inside) & (rng.random((size, size)) < 0.05)
    img[dirty, :3] = rng.integers(1, 256, size=(int(dirty.sum()), 3))
    return img


def write_dataset(root, n_train, n_test, size=64, seed=5, directions=(0, 2)):
    """-> {(split, direction): uint8 (n, S, S, 4)}; files under root/datasets/rpg-maker-xp/..."""
    rng = np.random.default_rng(seed)
    out = {}
    for split, n in (("train", n_train), ("test", n_test)):
        pals = [None] * n
        for k in range(n):
            p = int(rng.integers(4, 12))
            pals[k] = np.concatenate([rng.integers(0, 256, size=(p, 3)), np.full((p, 1), 255)], axis=1).astype(np.uint8)
        for d in directions:
            folder = os.path.join(root, "datasets", "rpg-maker-xp", split, DIRECTION_FOLDERS[d])
            os.makedirs(folder, exist_ok=True)
            imgs = np.stack([synthetic_sprite(rng, size, pals[k]) for k in range(n)])
            for k in range(n):
                with open(os.path.join(folder, f"{k}.png"), "wb") as f:
                    f.write(encode_png(imgs[k], filters=[(0, 1, 2, 3, 4), (4, 3), (1,), (2, 4, 0)][k % 4]))
            out[(split, d)] = imgs
    return out
