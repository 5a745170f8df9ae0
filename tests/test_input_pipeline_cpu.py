"""CPU tests of the sprite input pipeline: PNG decoding, the oracle's augmentation known answers, and the
host logic of the two dataset classes (device="cpu": everything except the batch kernels)."""
import os

import numpy as np
import pytest
import torch

from oracle import input_pipeline as ip
from palette_and_histo_gan_amd import dataset_utils as D
from palette_and_histo_gan_amd import io_utils, png
from tests import sprite_fixtures as F


def test_png_decoder_all_filter_types_and_errors(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(16, 12, 4), dtype=np.uint8)
    for filters in [(0,), (1,), (2,), (3,), (4,), (0, 1, 2, 3, 4), (4, 3, 2, 1)]:
        assert (png.decode_png(F.encode_png(img, filters)) == img).all()
    assert (png.decode_png(png.encode_png(img)) == img).all()          # the product's own writer (preview sheets)
    with pytest.raises(ValueError):
        png.decode_png(b"not a png at all")
    bad = bytearray(F.encode_png(img))
    bad[24] = 16                                   # bit depth 16
    with pytest.raises(ValueError):
        png.decode_png(bytes(bad))


@pytest.mark.skipif(not os.path.isdir("/root/reference/datasets/rpg-maker-xp"), reason="reference dataset not on this machine")
def test_png_decoder_equals_pil_on_the_reference_sprites():
    Image = pytest.importorskip("PIL.Image")
    import glob
    files = sorted(glob.glob("/root/reference/datasets/rpg-maker-xp/*/*/*.png"))
    assert len(files) == 1176
    for f in files[::3]:
        assert (png.read_png(f) == np.asarray(Image.open(f).convert("RGBA"))).all(), f


def test_oracle_known_answers():
    red = np.array([[[255.0, 0.0, 0.0]]])
    assert np.allclose(ip.adjust_hue(red, 0.0), red)
    assert np.allclose(ip.adjust_hue(red, 1 / 3), [[[0, 255, 0]]], atol=1e-9)
    assert np.allclose(ip.adjust_hue(red, -1 / 3), [[[0, 0, 255]]], atol=1e-9)
    assert np.allclose(ip.adjust_hue(red, 0.5), [[[0, 255, 255]]], atol=1e-9)
    grey = np.array([[[90.0, 90.0, 90.0], [0.0, 0.0, 0.0]]])
    assert np.array_equal(ip.adjust_hue(grey, 0.37), grey)
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, size=(8, 8, 3)).astype(np.float64)
    out = ip.adjust_hue(rgb, 0.21)
    assert np.allclose(out.max(-1), rgb.max(-1)) and np.allclose(out.min(-1), rgb.min(-1))      # value and chroma kept
    assert np.allclose(ip.adjust_hue(out, -0.21), rgb, atol=1e-9)
    img = np.arange(5 * 5 * 4, dtype=np.float64).reshape(5, 5, 4)
    sh = ip.translate_nearest(img, 2.0, -1.0)      # content moves 2 rows down, 1 column left; the uncovered part is 0
    assert np.array_equal(sh[2:, :4], img[:3, 1:]) and not sh[:2].any() and not sh[:, 4].any()
    half = ip.translate_nearest(img, 0.5, 0.0)     # y - 0.5 rounds away from zero: row 0 samples row -1 (fill), row y >= 1 itself
    assert not half[0].any() and np.array_equal(half[1:], img[1:])
    assert np.array_equal(ip.translate_nearest(img, 0.49, 0.0), img)
    px = np.array([[[10, 20, 30, 0], [10, 20, 30, 255]]], np.uint8)
    assert np.array_equal(ip.blacken_transparent_pixels(px), [[[0, 0, 0, 0], [10, 20, 30, 255]]])
    s, t = ip.make_pair(px, px)
    assert np.allclose(s[0, 0], -1.0) and np.allclose(s[0, 1], np.array([10, 20, 30, 255]) / 127.5 - 1)


def test_dataset_host_logic(tmp_path):
    data = F.write_dataset(str(tmp_path), 7, 3, directions=(0, 2))
    paths = D.sprite_paths(2, [7], "train", root=str(tmp_path))
    assert paths[3].endswith(os.path.join("datasets", "rpg-maker-xp", "train", "2-front", "3.png"))
    two = D.sprite_paths(0, [2, 3], "test", data_folders=["a", "b"], root="r")          # image 3 of the concatenation = b/1.png
    assert two[3] == os.path.join("r", "b", "test", "0-back", "1.png") and len(two) == 5
    assert (D.load_sprites(paths) == data[("train", 2)]).all()
    ds = D.SpriteRGBADataset(data[("train", 0)], data[("train", 2)], augment=True, batch_size=4, seed=3, device="cpu")
    rng = np.random.default_rng(0)
    idx, aug = ds.batch_parameters(rng, np.arange(4))
    assert idx.dtype == np.int32 and (idx[1] == idx[0] + 7).all() and aug.shape == (4, 4) and aug.dtype == np.float32
    many = np.concatenate([ds.batch_parameters(rng, np.arange(4))[1] for _ in range(500)])
    assert 0.75 < many[:, 0].mean() < 0.85 and np.abs(many[:, 1]).max() <= 0.5
    assert -0.15 * 64 <= many[:, 2].min() and many[:, 2].max() <= 0.075 * 64 and np.abs(many[:, 3]).max() <= 0.125 * 64
    assert D.SpriteRGBADataset(data[("test", 0)], data[("test", 2)], augment=False, device="cpu").batch_parameters(rng, np.arange(3))[1] is None
    # the tf.data slice works on torch tensors as well as numpy arrays
    tens = D.Dataset.from_batches([(torch.arange(8).reshape(4, 2), torch.ones(4, 2))])
    ones = list(tens.unbatch().take(3).batch(1))
    assert len(ones) == 3 and isinstance(ones[0][0], torch.Tensor) and tuple(ones[1][0].shape) == (1, 2)


def test_indexed_dataset_tables_round_trip(tmp_path):
    data = F.write_dataset(str(tmp_path), 5, 2, directions=(0, 2))
    ds = D.SpriteIndexedDataset(data[("train", 0)], data[("train", 2)], "grayness", batch_size=2, device="cpu")
    src, tgt, pal = [t.numpy() for t in ds.tables]
    for k in range(5):
        p = pal[k].reshape(256, 4)
        for idx, sprites in ((src, data[("train", 0)]), (tgt, data[("train", 2)])):
            want = D.blacken_transparent_pixels(sprites[k]).astype(np.int32)
            assert np.array_equal(io_utils.indexed_to_rgba(idx[k].reshape(64, 64, 1), p), want)
        used = int(max(src[k].max(), tgt[k].max())) + 1
        gray = (p[:used, :3].astype(np.float32) * np.array([0.2989, 0.5870, 0.1140], np.float32)).sum(-1)
        assert (np.diff(gray) >= -1e-3).all() and (p[used:] == [255, 0, 220, 255]).all()
