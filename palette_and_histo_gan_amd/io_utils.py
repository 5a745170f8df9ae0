"""Palette helpers of the reference's io_utils.py (integer work on the host side of the input pipeline):

    extract_palette(image, palette_ordering, channels)      io_utils.py:25-64
    rgba_to_indexed(image, palette)                         io_utils.py:78-93
    indexed_to_rgba(indexed_image, palette)                 io_utils.py:96-103

All three are bit-exact integer functions.  extract_palette follows tf.raw_ops.UniqueWithCountsV2 (unique rows in order of
first appearance, image swept top-left to bottom-right) and, for "grayness", tf.argsort(..., stable=True) of the float32
product colours x [0.2989, 0.5870, 0.1140, 0] (summed in channel order), then pads to MAX_PALETTE_SIZE with
INVALID_INDEX_COLOR.  indexed_to_rgba accepts numpy arrays or torch tensors (on any device: a gather)."""
import numpy as np
import torch

from .configuration import INVALID_INDEX_COLOR, MAX_PALETTE_SIZE, OUTPUT_CHANNELS


def _unique_rows_first_appearance(flat):
    """rows of an int array [P, C] that are distinct, in order of first appearance (UniqueWithCountsV2, axis 0) + counts"""
    keys = np.zeros(len(flat), dtype=np.int64)
    for c in range(flat.shape[1]):
        keys = keys * 256 + flat[:, c].astype(np.int64)
    _, first, counts = np.unique(keys, return_index=True, return_counts=True)
    order = np.argsort(first, kind="stable")
    return flat[first[order]], counts[order]


def extract_palette(image, palette_ordering="grayness", channels=OUTPUT_CHANNELS, rng=None):
    """io_utils.py:25-64: int32 [MAX_PALETTE_SIZE, channels] palette of an (H, W, channels) image with values 0..255."""
    flat = np.asarray(image).astype(np.int32).reshape(-1, channels)
    if palette_ordering == "top2bottom":
        colors, _ = _unique_rows_first_appearance(flat)
    elif palette_ordering == "bottom2top":
        colors, _ = _unique_rows_first_appearance(flat[::-1])
    elif palette_ordering == "grayness":
        colors, _ = _unique_rows_first_appearance(flat)
        coeff = np.array([0.2989, 0.5870, 0.1140, 0.0], np.float32)[:channels]
        gray = np.zeros(len(colors), np.float32)
        for c in range(channels):               # float32 accumulation in channel order, like the [n,4] x [4,1] matmul
            gray = gray + colors[:, c].astype(np.float32) * coeff[c]
        colors = colors[np.argsort(gray, kind="stable")]
    else:                                       # "shuffled"
        colors, _ = _unique_rows_first_appearance(flat)
        colors = colors[(rng or np.random.default_rng()).permutation(len(colors))]
    if len(colors) > MAX_PALETTE_SIZE:
        raise ValueError(f"image has {len(colors)} distinct colours, more than MAX_PALETTE_SIZE = {MAX_PALETTE_SIZE}")
    fill = np.tile(np.array(INVALID_INDEX_COLOR[:channels], np.int32), (MAX_PALETTE_SIZE - len(colors), 1))
    return np.concatenate([colors.astype(np.int32), fill], axis=0)


def rgba_to_single_int(values_in_rgba):
    """io_utils.py:67-75 (the reference's multipliers: 2^24, 2^16, 2^8, 0)"""
    v = np.asarray(values_in_rgba).astype(np.int64)
    out = np.zeros(v.shape[:-1], np.int64)
    for i, m in enumerate([16777216, 65536, 256, 0]):
        out += v[..., i] * m
    return out.astype(np.int32)


def rgba_to_indexed(image, palette):
    """io_utils.py:78-93: int32 (H, W, 1) index of every pixel's colour in `palette`; a colour that is not in the palette maps
    to 0 (tf.scatter_nd leaves it at its zero initial value); if the colour occurs twice in the palette the reference's
    scatter_nd keeps one of them -- here the highest palette position, as TF's CPU scatter (last write) does."""
    img = np.asarray(image).astype(np.int32)
    pal = np.asarray(palette).astype(np.int32)
    h, w, c = img.shape
    flat = img.reshape(-1, c)
    key = lambda a: sum(a[:, k].astype(np.int64) << (8 * (c - 1 - k)) for k in range(c))
    pk, fk = key(pal), key(flat)
    order = np.argsort(pk, kind="stable")
    spk = pk[order]
    pos = np.searchsorted(spk, fk, side="right") - 1           # last palette entry with that key
    pos = np.clip(pos, 0, len(spk) - 1)
    hit = spk[pos] == fk
    idx = np.where(hit, order[pos], 0).astype(np.int32)
    return idx.reshape(h, w, 1)


def indexed_to_rgba(indexed_image, palette):
    """io_utils.py:96-103: (H, W, 1) indices -> (H, W, channels) colours; torch (any device) or numpy."""
    if isinstance(indexed_image, torch.Tensor) or isinstance(palette, torch.Tensor):
        idx = torch.as_tensor(indexed_image).long()
        pal = torch.as_tensor(palette).to(idx.device)
        out = pal[idx.reshape(-1)]
        return out.reshape(idx.shape[0], idx.shape[1], -1)
    idx = np.asarray(indexed_image).astype(np.int64)
    pal = np.asarray(palette)
    return pal[idx.reshape(-1)].reshape(idx.shape[0], idx.shape[1], -1)
