"""Batch sources with the small slice of the tf.data API that the reference's training loop relies on
(side2side_model.py:73 `.repeat().take(n).enumerate()`, pix2pix_model.py:107-108 `.unbatch().take(n).batch(1)`,
`.as_numpy_iterator()`), and the sprite input pipeline of the reference (dataset_utils.py:66-246):

    load_rgba_ds(source_direction, target_direction, augment=True)        dataset_utils.py:209-229
    load_indexed_ds(source_direction, target_direction, palette_ordering) dataset_utils.py:232-246

MI355X-first layout: the sprite set is tiny (19 MB decoded), so it is decoded once (png.py) and kept in HBM; a batch is ONE
kernel launch (csrc/sprites.hip: gather of the shuffled pair, blacken, shared hue rotation + translation with probability
0.8, normalise) that writes the f32 tensors train_step takes -- no host worker pool, no per-step PCIe traffic beyond the
B x 6 numbers that select and augment the batch.  The value contract of what train_step receives is the reference's: RGBA
batches are (source, target) f32 (B,S,S,4) in [-1,1] with transparent pixels at -1, indexed batches are (source_idx,
target_idx, palette) int32.  The random streams (shuffle order, augmentation draws) are numpy's, not TensorFlow's."""
import itertools
import os

import numpy as np

from .configuration import (BATCH_SIZE, DATA_FOLDERS, DIRECTION_FOLDERS, IMG_SIZE, INVALID_INDEX_COLOR, MAX_PALETTE_SIZE, SEED,
                            TEST_SIZES, TRAIN_SIZES)


class ShardedBatch(tuple):
    """A batch that already is ONE RANK'S contiguous shard of a global batch (data parallelism): the tensors hold
    rows [offset, offset + len) of the `global_batch` samples every rank agreed on.  Unpacks like the plain tuple train_step
    takes (pix2pix_model.py:64, :297)."""

    def __new__(cls, tensors, global_batch, offset):
        self = super().__new__(cls, tensors)
        self.global_batch, self.offset = int(global_batch), int(offset)
        return self


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
                    yield tuple(_item(t, i) for t in batch)
        return Dataset(gen)

    def batch(self, n):
        def gen():
            it = self._make_iter()
            while True:
                chunk = list(itertools.islice(it, n))
                if not chunk:
                    return
                yield tuple(_stack([c[k] for c in chunk]) for k in range(len(chunk[0])))
        return Dataset(gen)

    def as_numpy_iterator(self):
        return self._make_iter()


def _is_torch(t):
    return type(t).__module__.startswith("torch")


_COPY_STREAMS = {}


def upload_async(tensors, device="cuda:0"):
    """Host batch elements -> device tensors WITHOUT putting the PCIe copy into the compute stream (tf.data's prefetch-to-device for a
    batch source that yields numpy arrays, e.g. a pipeline of the reference fed through Dataset.from_batches).  The copy is issued on a
    stream of its own; the caller's current stream waits for it with an event.  The host runs several steps ahead of the GPU (a
    replayed train step costs it 0.4 ms), so the upload of step k + 1 runs while the GPU computes step k: measured at c2 (33.5 MB per
    step) 2.70 ms/step with the copy in the compute stream, 2.05 ms with this.  Elements that already live on the device pass through."""
    import torch
    dev = torch.device(device)
    if all(_is_torch(t) and t.is_cuda for t in tensors):
        return list(tensors)
    cs = _COPY_STREAMS.get(str(dev))
    if cs is None:
        cs = _COPY_STREAMS[str(dev)] = torch.cuda.Stream(device=dev)
    cur = torch.cuda.current_stream(dev)
    out = []
    with torch.cuda.stream(cs):
        for t in tensors:
            if _is_torch(t) and t.is_cuda:
                out.append(t)
                continue
            h = torch.as_tensor(np.ascontiguousarray(t) if not _is_torch(t) else t)
            if h.dtype == torch.float64:
                h = h.float()
            out.append(h.to(dev, non_blocking=True))
        ev = torch.cuda.Event()
        ev.record(cs)
    cur.wait_event(ev)
    for t in out:
        t.record_stream(cur)          # allocated under the copy stream, consumed under the caller's
    return out


def _item(t, i):
    return t[i] if _is_torch(t) else np.asarray(t)[i]


def _stack(items):
    if _is_torch(items[0]):
        import torch
        return torch.stack(items)
    return np.stack(items)


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


# ---- the sprite pipeline (dataset_utils.py:66-246) ---------------------------------------------------------------------------
AUGMENT_PROBABILITY = 0.8                      # create_augmentation_with_prob(0.8), dataset_utils.py:219
HUE_MAX_DELTA = 0.5                            # tf.image.stateless_random_hue(image_rgb, 0.5, seed), :82
TRANSLATE_HEIGHT = (-0.15, 0.075)              # RandomTranslation((-0.15, 0.075), 0.125, ...), :89
TRANSLATE_WIDTH = (-0.125, 0.125)


def sprite_paths(direction, sizes, split, data_folders=None, root="."):
    """files of one direction in image-number order: image k of the concatenated datasets is file `<k - offset>.png` of the
    dataset it falls in (the tf.while_loop of dataset_utils.py:158-166,184-192)"""
    out = []
    for folder, n in zip(data_folders or DATA_FOLDERS, sizes):
        out += [os.path.join(root, folder, split, DIRECTION_FOLDERS[direction], f"{i}.png") for i in range(n)]
    return out


def load_sprites(paths, img_size=IMG_SIZE):
    """uint8 (n, S, S, 4), as tf.image.decode_png(channels=4) + reshape (dataset_utils.py:67-69)"""
    from . import png
    out = np.empty((len(paths), img_size, img_size, 4), np.uint8)
    for k, p in enumerate(paths):
        img = png.read_png(p)
        if img.shape != (img_size, img_size, 4):
            raise ValueError(f"{p}: expected a {img_size}x{img_size} sprite, got {img.shape[:2]}")
        out[k] = img
    return out


def blacken_transparent_pixels(image):
    """dataset_utils.py:11-20 (host form, used at load time by the indexed pipeline): colour of alpha == 0 pixels -> 0"""
    image = np.array(image)
    image[image[..., 3] == 0] = 0
    return image


class SpriteRGBADataset(Dataset):
    """load_rgba_ds's train or test dataset: range(n).shuffle(n) [reshuffled every iteration] -> load pair -> augment with
    probability 0.8 (train, augment=True) -> normalise -> batch(batch_size) without drop_remainder."""

    def __init__(self, source_sprites, target_sprites, augment, batch_size=BATCH_SIZE, seed=SEED, device=None):
        import torch
        self.torch = torch
        self.device = torch.device(device or "cuda:0")
        self.n, self.S = len(source_sprites), source_sprites.shape[1]
        both = np.concatenate([source_sprites, target_sprites], axis=0)
        self.sprites = torch.from_numpy(both).to(self.device)            # uint8 [2n][S][S][4], resident
        self.augment, self.batch_size, self.seed = augment, batch_size, seed
        self.epoch = 0
        self.shard = (0, 1)              # (rank, world): which rows of every global batch make_batch produces
        super().__init__(self._iterate)

    def set_shard(self, rank, world):
        """data parallelism: every rank draws the SAME shuffle order and augmentation rows (same seed) and materialises only
        its contiguous share of each batch -- nothing is produced eight times to keep an eighth"""
        self.shard = (int(rank), int(world))
        return self

    def unsharded(self):
        """a view that yields whole batches whatever the shard (evaluation reads the first images of the dataset)"""
        return Dataset(lambda: self._iterate(shard=(0, 1)))

    def repeat(self):
        """the TRAINING stream (side2side_model.py:73): epoch k is shuffled by (seed, k), whatever other iterators were taken
        from this dataset in between (evaluation on one rank must not move the order the other ranks see)"""
        return Dataset(lambda: itertools.chain.from_iterable(self._iterate(epoch=k) for k in itertools.count()))

    def batch_parameters(self, rng, picks):
        """host-side draws of one batch: (int32 [2][B] sprite numbers, f32 [B][4] augmentation rows or None)"""
        B = len(picks)
        idx = np.stack([picks, picks + self.n]).astype(np.int32)
        if not self.augment:
            return idx, None
        aug = np.zeros((B, 4), np.float32)
        aug[:, 0] = rng.random(B) < AUGMENT_PROBABILITY
        aug[:, 1] = rng.uniform(-HUE_MAX_DELTA, HUE_MAX_DELTA, B)
        aug[:, 2] = rng.uniform(*TRANSLATE_HEIGHT, B) * self.S
        aug[:, 3] = rng.uniform(*TRANSLATE_WIDTH, B) * self.S
        return idx, aug

    def make_batch(self, idx, aug):
        """one kernel launch: (source, target) f32 (B,S,S,4) device tensors"""
        import ctypes as C
        from . import _lib as L
        torch = self.torch
        B = idx.shape[1]
        if B == 0:       # this rank's share of a ragged batch smaller than the world
            z = torch.empty((0, self.S, self.S, 4), dtype=torch.float32, device=self.device)
            return z, z.clone()
        idx = np.ascontiguousarray(idx)
        if idx.min() < 0 or idx.max() >= 2 * self.n:
            raise IndexError(f"sprite numbers must lie in [0, {2 * self.n})")
        idx_d = torch.from_numpy(idx).to(self.device, non_blocking=True)
        aug_d = torch.from_numpy(np.ascontiguousarray(aug)).to(self.device, non_blocking=True) if aug is not None else None
        src = torch.empty((B, self.S, self.S, 4), dtype=torch.float32, device=self.device)
        tgt = torch.empty_like(src)
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        L.call("p2p_sprites_rgba_batch", C.c_void_p(self.sprites.data_ptr()), 2 * self.n, self.S, C.c_void_p(idx_d[0].data_ptr()),
               C.c_void_p(idx_d[1].data_ptr()), C.c_void_p(aug_d.data_ptr()) if aug_d is not None else None, B, 1,
               C.c_void_p(src.data_ptr()), C.c_void_p(tgt.data_ptr()), st)
        return src, tgt

    def _iterate(self, shard=None, epoch=None):
        if epoch is None:            # an ad-hoc iteration (evaluation, previews): reshuffled each time like tf.data's shuffle
            epoch = 1_000_000 + self.epoch
            self.epoch += 1
        rank, world = shard or self.shard
        rng = np.random.default_rng([self.seed, epoch])
        order = rng.permutation(self.n)
        for i in range(0, self.n, self.batch_size):
            idx, aug = self.batch_parameters(rng, order[i:i + self.batch_size])       # drawn for the GLOBAL batch on every rank
            if world == 1:
                yield self.make_batch(idx, aug)
                continue
            from .parallel import shard_bounds
            Bg = idx.shape[1]
            lo, hi = shard_bounds(Bg, world, rank)
            yield ShardedBatch(self.make_batch(idx[:, lo:hi], aug[lo:hi] if aug is not None else None), Bg, lo)


class SpriteIndexedDataset(Dataset):
    """load_indexed_ds's train or test dataset (dataset_utils.py:123-164,232-246).  The union palette of every pair and both
    index maps are extracted once here (io_utils.py) and live in HBM as int32 tables; a batch is three row gathers.  For the
    deterministic orderings ("grayness", "top2bottom", "bottom2top") nothing on this path is random per step.  "shuffled" is:
    the reference shuffles the colours inside the dataset map (io_utils.py:53-55), i.e. every time a sample is loaded, so the
    tables keep the first-appearance order and every batch draws a fresh permutation per sample, applied by one launch
    (p2p_palette_relabel_batch: palette rows permuted, both index maps re-labelled through the inverse)."""

    def __init__(self, source_sprites, target_sprites, palette_ordering, batch_size=BATCH_SIZE, seed=SEED, device=None):
        import torch
        from . import io_utils
        self.torch = torch
        self.device = torch.device(device or "cuda:0")
        self.n, self.S = len(source_sprites), source_sprites.shape[1]
        self.palette_ordering = palette_ordering
        self.reshuffle = palette_ordering not in ("grayness", "top2bottom", "bottom2top")
        src_idx = np.empty((self.n, self.S, self.S, 1), np.int32)
        tgt_idx = np.empty_like(src_idx)
        pal = np.empty((self.n, MAX_PALETTE_SIZE, 4), np.int32)
        self.ncolors = np.empty(self.n, np.int32)            # distinct colours of each pair (the rest of a palette is padding)
        for k in range(self.n):
            s = blacken_transparent_pixels(source_sprites[k]).astype(np.int32)
            t = blacken_transparent_pixels(target_sprites[k]).astype(np.int32)
            both = np.concatenate([s, t], axis=-1)
            pal[k] = io_utils.extract_palette(both, "top2bottom" if self.reshuffle else palette_ordering)
            self.ncolors[k] = len(io_utils._unique_rows_first_appearance(both.reshape(-1, 4))[0])
            src_idx[k] = io_utils.rgba_to_indexed(s, pal[k])
            tgt_idx[k] = io_utils.rgba_to_indexed(t, pal[k])
        self.tables = [torch.from_numpy(a.reshape(self.n, -1)).to(self.device) for a in (src_idx, tgt_idx, pal)]
        self.shapes = [(self.S, self.S, 1), (self.S, self.S, 1), (MAX_PALETTE_SIZE, 4)]
        self.batch_size, self.seed, self.epoch = batch_size, seed, 0
        self.shard = (0, 1)
        super().__init__(self._iterate)

    set_shard = SpriteRGBADataset.set_shard
    unsharded = SpriteRGBADataset.unsharded
    repeat = SpriteRGBADataset.repeat

    def palette_permutations(self, rng, picks):
        """per sample: perm (position j takes the colour at perm[j]) over its real colours, identity over the padding; and the
        inverse.  int32 [B][MAX_PALETTE_SIZE] each"""
        B = len(picks)
        perm = np.tile(np.arange(MAX_PALETTE_SIZE, dtype=np.int32), (B, 1))
        inv = perm.copy()
        for b, k in enumerate(picks):
            nc = int(self.ncolors[k])
            p = rng.permutation(nc).astype(np.int32)
            perm[b, :nc] = p
            inv[b, p] = np.arange(nc, dtype=np.int32)
        return perm, inv

    def make_batch(self, picks, relabel=None):
        import ctypes as C
        from . import _lib as L
        torch = self.torch
        picks = np.asarray(picks, np.int32)
        B = len(picks)
        if B == 0:
            return tuple(torch.empty((0,) + shape, dtype=torch.int32, device=self.device) for shape in self.shapes)
        if picks.min() < 0 or picks.max() >= self.n:
            raise IndexError(f"sample numbers must lie in [0, {self.n})")
        sel = torch.from_numpy(np.ascontiguousarray(picks)).to(self.device, non_blocking=True)
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        out = []
        for table, shape in zip(self.tables, self.shapes):
            o = torch.empty((B,) + shape, dtype=torch.int32, device=self.device)
            L.call("p2p_gather_rows_i32", C.c_void_p(table.data_ptr()), self.n, table.shape[1], C.c_void_p(sel.data_ptr()), B,
                   C.c_void_p(o.data_ptr()), st)
            out.append(o)
        if relabel is not None:
            perm, inv = (torch.from_numpy(np.ascontiguousarray(a)).to(self.device, non_blocking=True) for a in relabel)
            res = [torch.empty_like(o) for o in out]
            L.call("p2p_palette_relabel_batch", C.c_void_p(out[0].data_ptr()), C.c_void_p(out[1].data_ptr()),
                   C.c_void_p(out[2].data_ptr()), C.c_void_p(perm.data_ptr()), C.c_void_p(inv.data_ptr()), B, self.S * self.S,
                   MAX_PALETTE_SIZE, 4, C.c_void_p(res[0].data_ptr()), C.c_void_p(res[1].data_ptr()), C.c_void_p(res[2].data_ptr()), st)
            out = res
        return tuple(out)

    def _iterate(self, shard=None, epoch=None):
        if epoch is None:
            epoch = 1_000_000 + self.epoch
            self.epoch += 1
        rank, world = shard or self.shard
        rng = np.random.default_rng([self.seed, epoch])
        order = rng.permutation(self.n)
        for i in range(0, self.n, self.batch_size):
            picks = order[i:i + self.batch_size]
            relabel = self.palette_permutations(rng, picks) if self.reshuffle else None      # drawn for the GLOBAL batch on every rank
            if world == 1:
                yield self.make_batch(picks, relabel)
                continue
            from .parallel import shard_bounds
            lo, hi = shard_bounds(len(picks), world, rank)
            rl = (relabel[0][lo:hi], relabel[1][lo:hi]) if relabel is not None else None
            yield ShardedBatch(self.make_batch(picks[lo:hi], rl), len(picks), lo)


def _pair_sprites(source_direction, target_direction, split, sizes, data_folders, root):
    return (load_sprites(sprite_paths(source_direction, sizes, split, data_folders, root)),
            load_sprites(sprite_paths(target_direction, sizes, split, data_folders, root)))


def _default_seed(seed):
    if seed is not None:
        return seed
    from .tf_compat import global_seed          # tf.random.set_seed(SEED) of the notebook (experiments.ipynb cell 3)
    return global_seed()


def load_rgba_ds(source_direction, target_direction, augment=True, *, batch_size=BATCH_SIZE, data_folders=None, root=".",
                 train_sizes=None, test_sizes=None, seed=None, device=None):
    """dataset_utils.py:209-229 -> (train_dataset, test_dataset); keyword arguments are this build's (the reference reads the
    same values from configuration.py)."""
    seed = _default_seed(seed)
    tr = _pair_sprites(source_direction, target_direction, "train", train_sizes or TRAIN_SIZES, data_folders, root)
    te = _pair_sprites(source_direction, target_direction, "test", test_sizes or TEST_SIZES, data_folders, root)
    return (SpriteRGBADataset(*tr, augment=augment, batch_size=batch_size, seed=seed, device=device),
            SpriteRGBADataset(*te, augment=False, batch_size=batch_size, seed=seed + 1, device=device))


def load_indexed_ds(source_direction, target_direction, palette_ordering, *, batch_size=BATCH_SIZE, data_folders=None, root=".",
                    train_sizes=None, test_sizes=None, seed=None, device=None):
    """dataset_utils.py:232-246 -> (train_dataset, test_dataset) of (source_idx, target_idx, palette) batches"""
    seed = _default_seed(seed)
    tr = _pair_sprites(source_direction, target_direction, "train", train_sizes or TRAIN_SIZES, data_folders, root)
    te = _pair_sprites(source_direction, target_direction, "test", test_sizes or TEST_SIZES, data_folders, root)
    return (SpriteIndexedDataset(*tr, palette_ordering=palette_ordering, batch_size=batch_size, seed=seed, device=device),
            SpriteIndexedDataset(*te, palette_ordering=palette_ordering, batch_size=batch_size, seed=seed + 1, device=device))
