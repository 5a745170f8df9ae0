"""Not-gpu: host logic of data-parallel batch production (dataset_utils.set_shard / ShardedBatch): every
rank draws the same shuffle order and augmentation rows and materialises only its contiguous share of each batch.  The device
launch (make_batch) is replaced by its numpy meaning so the algebra runs without a GPU."""
import numpy as np
import torch

from palette_and_histo_gan_amd import dataset_utils as D
from palette_and_histo_gan_amd.parallel import shard_bounds


class _HostRGBA(D.SpriteRGBADataset):
    """batch = (sprite numbers, augmentation rows) as float tensors: enough to see WHICH rows a rank was handed"""

    def make_batch(self, idx, aug):
        B = idx.shape[1]
        a = aug if aug is not None else np.zeros((B, 4), np.float32)
        return torch.as_tensor(idx[0].astype(np.float32).reshape(B, 1)), torch.as_tensor(np.asarray(a, np.float32).reshape(B, 4))


def _ds(n=10, batch=4, augment=True):
    sprites = np.zeros((n, 8, 8, 4), np.uint8)
    return _HostRGBA(sprites, sprites, augment=augment, batch_size=batch, seed=3, device="cpu")


def _epochs(ds, n_batches):
    return [tuple(t.numpy().copy() for t in b) + (getattr(b, "global_batch", len(b[0])), getattr(b, "offset", 0))
            for _, b in zip(range(n_batches), ds.repeat())]


def test_rank_shards_concatenate_to_the_global_batches_including_ragged_and_empty_ones():
    whole = _epochs(_ds(), 7)                                   # 10 samples, batch 4: 4 + 4 + 2 per epoch, two epochs and one batch
    assert [len(b[0]) for b in whole] == [4, 4, 2, 4, 4, 2, 4]
    for world in (2, 3, 8):
        ranks = [_epochs(_ds().set_shard(r, world), 7) for r in range(world)]
        for k, ref in enumerate(whole):
            Bg = len(ref[0])
            parts = [ranks[r][k] for r in range(world)]
            assert all(p[2] == Bg for p in parts)                                   # every rank knows the global batch size
            for r, p in enumerate(parts):
                lo, hi = shard_bounds(Bg, world, r)
                assert p[3] == lo and len(p[0]) == hi - lo                          # ... and where its rows sit (empty shards too)
            assert np.array_equal(np.concatenate([p[0] for p in parts]), ref[0])    # same samples in the same order
            assert np.array_equal(np.concatenate([p[1] for p in parts]), ref[1])    # same augmentation draw per sample


def test_training_order_does_not_depend_on_evaluation_iterations_taken_in_between():
    a, b = _ds(), _ds()
    ita = iter(a.repeat())
    first = [next(ita)[0].numpy().copy() for _ in range(2)]
    for _ in range(3):
        list(a)                                                 # rank 0 evaluates: ad-hoc iterations of the same dataset
        list(a.unsharded())
    rest = [next(ita)[0].numpy().copy() for _ in range(4)]
    want = [x[0] for x in _epochs(b, 6)]
    assert all(np.array_equal(x, y) for x, y in zip(first + rest, want))
    # ad-hoc iterations are still reshuffled each time (tf.data shuffle(reshuffle_each_iteration), dataset_utils.py:217)
    e1, e2 = [x[0].numpy() for x in a], [x[0].numpy() for x in a]
    assert not all(np.array_equal(x, y) for x, y in zip(e1, e2))


def test_unsharded_view_yields_whole_batches_on_every_rank():
    ds = _ds(augment=False).set_shard(1, 2)
    assert [len(b[0]) for b in ds] == [2, 2, 1]
    assert [len(b[0]) for b in ds.unsharded()] == [4, 4, 2]
    assert all(not isinstance(b, D.ShardedBatch) for b in ds.unsharded())


def test_shuffled_palette_permutations_are_inverse_pairs_over_the_real_colours_only():
    ds = D.SpriteIndexedDataset.__new__(D.SpriteIndexedDataset)
    ds.ncolors = np.array([5, 1, 256, 17], np.int32)
    rng = np.random.default_rng(0)
    perm, inv = ds.palette_permutations(rng, [0, 1, 2, 3, 0])
    assert perm.shape == inv.shape == (5, 256) and perm.dtype == np.int32
    for b, k in enumerate([0, 1, 2, 3, 0]):
        nc = ds.ncolors[k]
        assert sorted(perm[b, :nc]) == list(range(nc))
        assert np.array_equal(perm[b, nc:], np.arange(nc, 256)) and np.array_equal(inv[b, nc:], np.arange(nc, 256))
        assert np.array_equal(inv[b][perm[b]], np.arange(256)) and np.array_equal(perm[b][inv[b]], np.arange(256))
    assert not np.array_equal(perm[0], perm[4])                 # the same sample drawn twice gets two permutations
