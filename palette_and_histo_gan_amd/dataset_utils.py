"""Batch sources with the small slice of the tf.data API that the reference's training loop relies on
(side2side_model.py:73 `.repeat().take(n).enumerate()`, pix2pix_model.py:107-108 `.unbatch().take(n).batch(1)`,
`.as_numpy_iterator()`).  The PNG pipeline of the reference (dataset_utils.py:66-246) is out of scope for this round
(SURVEY.md 8f F1); the value contract of what train_step receives is kept: RGBA batches are (source, target) f32
(B,S,S,4) in [-1,1] with transparent pixels at -1, indexed batches are (source_idx, target_idx, palette) int32."""
import itertools

import numpy as np

from .configuration import BATCH_SIZE, IMG_SIZE, MAX_PALETTE_SIZE, SEED, INVALID_INDEX_COLOR


class Dataset:
    """An in-memory, re-iterable sequence of batches (tuples of numpy arrays)."""

    def __init__(self, make_iter):
        self._make_iter = make_iter

    @staticmethod
    def from_batches(batches):
        batches = list(batches)
        return Dataset(lambda: iter(batches))

    def __iter__(self):
        return self._make_iter()

    def repeat(self):
        return Dataset(lambda: itertools.chain.from_iterable(self._make_iter() for _ in itertools.count()))

    def take(self, n):
        return Dataset(lambda: itertools.islice(self._make_iter(), int(n)))

    def enumerate(self):
        return Dataset(lambda: enumerate(self._make_iter()))

    def unbatch(self):
        def gen():
            for batch in self._make_iter():
                for i in range(len(batch[0])):
                    yield tuple(np.asarray(t)[i] for t in batch)
        return Dataset(gen)

    def batch(self, n):
        def gen():
            it = self._make_iter()
            while True:
                chunk = list(itertools.islice(it, n))
                if not chunk:
                    return
                yield tuple(np.stack([c[k] for c in chunk]) for k in range(len(chunk[0])))
        return Dataset(gen)

    def as_numpy_iterator(self):
        return self._make_iter()


def _sprite(rng, size, palette, pixels_transparent=0.835):
    opaque = rng.random((size, size)) >= pixels_transparent
    idx = rng.integers(0, len(palette), size=(size, size))
    return np.where(opaque[..., None], palette[idx], 0).astype(np.uint8)


def synthetic_rgba_ds(n_images, batch_size=BATCH_SIZE, img_size=IMG_SIZE, palette_size=None, seed=SEED):
    """Sprite-like pairs with the statistics of the shipped dataset: per image a palette of opaque
    colours, 83.5 % transparent pixels, normalised x/127.5 - 1 (dataset_utils.py:39-48), batch() without
    drop_remainder (:223) so the last batch may be ragged."""
    rng = np.random.default_rng(seed)
    src, tgt = [], []
    for _ in range(n_images):
        P = palette_size or int(rng.integers(10, 55))
        pal = np.concatenate([rng.integers(0, 256, size=(P, 3)), np.full((P, 1), 255)], axis=1)
        src.append(_sprite(rng, img_size, pal))
        tgt.append(_sprite(rng, img_size, pal))
    norm = lambda a: np.stack(a).astype(np.float32) / 127.5 - 1.0
    src, tgt = norm(src), norm(tgt)
    return Dataset.from_batches((src[i:i + batch_size], tgt[i:i + batch_size]) for i in range(0, n_images, batch_size))


def synthetic_indexed_ds(n_images, batch_size=BATCH_SIZE, img_size=IMG_SIZE, palette_size=24, seed=SEED):
    """Indexed triples (source_idx, target_idx, palette) as load_indexed_ds yields them (dataset_utils.py:232-246):
    int32 (B,S,S,1) indices with transparent black at 0, palette (B,256,4) padded with INVALID_INDEX_COLOR."""
    rng = np.random.default_rng(seed)

    def draw():
        opaque = rng.random((n_images, img_size, img_size, 1)) >= 0.835
        idx = rng.integers(1, palette_size, size=(n_images, img_size, img_size, 1))
        return np.where(opaque, idx, 0).astype(np.int32)

    pal = np.tile(np.array(INVALID_INDEX_COLOR, np.int32), (n_images, MAX_PALETTE_SIZE, 1))
    pal[:, :palette_size, :3] = rng.integers(0, 256, size=(n_images, palette_size, 3))
    pal[:, :palette_size, 3] = 255
    pal[:, 0] = 0
    src, tgt = draw(), draw()
    return Dataset.from_batches((src[i:i + batch_size], tgt[i:i + batch_size], pal[i:i + batch_size])
                                for i in range(0, n_images, batch_size))


def synthetic_rgba_batch(rng, batch, img_size, palette_size=None):
    """One sprite-like RGBA batch (source, target) in [-1,1] for benchmarks: per image a palette of P
    opaque colours (P = palette_size, or uniform in 10..54 like the shipped sprites), pixels transparent (0,0,0,0) with
    probability 0.835, source and target share the palette but draw their pixels independently
    (dataset_utils.py:11-20,39-48 value contract)."""
    src = np.zeros((batch, img_size, img_size, 4), np.uint8)
    tgt = np.zeros_like(src)
    for b in range(batch):
        P = palette_size or int(rng.integers(10, 55))
        pal = np.concatenate([rng.integers(0, 256, size=(P, 3)), np.full((P, 1), 255)], axis=1).astype(np.uint8)
        for out in (src, tgt):
            opaque = rng.random((img_size, img_size)) >= 0.835
            idx = rng.integers(0, P, size=(img_size, img_size))
            out[b] = np.where(opaque[..., None], pal[idx], 0)
    to_f = lambda a: (a.astype(np.float32) / 127.5 - 1.0)
    return to_f(src), to_f(tgt)


def synthetic_indexed_batch(rng, batch, img_size, palette_size=24):
    """One indexed batch (source_idx, target_idx, palette): int32 (B,S,S,1) with 0 (transparent black) with probability 0.835,
    else uniform in 1..P-1; palette (B,256,4) padded with INVALID_INDEX_COLOR (configuration.py:31-32, io_utils.py:50-63)."""
    def draw():
        opaque = rng.random((batch, img_size, img_size, 1)) >= 0.835
        idx = rng.integers(1, palette_size, size=(batch, img_size, img_size, 1))
        return np.where(opaque, idx, 0).astype(np.int32)
    pal = np.tile(np.array(INVALID_INDEX_COLOR, np.int32), (batch, MAX_PALETTE_SIZE, 1))
    pal[:, :palette_size, :3] = rng.integers(0, 256, size=(batch, palette_size, 3))
    pal[:, :palette_size, 3] = 255
    pal[:, 0] = 0
    return draw(), draw(), pal
