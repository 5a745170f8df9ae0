"""CPU restatement (numpy, float64 where the reference computes in float) of the reference's sprite input pipeline for
ONE pair of images -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

    blacken_transparent_pixels   dataset_utils.py:11-20
    normalize                    dataset_utils.py:39-49
    adjust_hue                   dataset_utils.py:80-84  -> tf.image.stateless_random_hue(rgb, 0.5, seed) = adjust_hue(rgb, delta)
    translate_nearest            dataset_utils.py:87-92  -> keras RandomTranslation(fill_mode="constant", interpolation="nearest")
    augment_pair / make_batch    dataset_utils.py:95-120,209-229

PARITY UNPINNED for the two third-party ops: tensorflow 2.9.1's AdjustHue kernel and keras 2.9.0's RandomTranslation (an
ImageProjectiveTransformV3 with the translation matrix [1,0,-dx,0,1,-dy]) are absent from /root/reference; their published
semantics are restated: hue is the HSV hue angle, rotated by `delta` turns with value and chroma kept; the transform samples
input (x - dx, y - dy), rounds half away from zero (std::round) and fills with 0 outside.  The random draws themselves
(TF's stateless/stateful generators) are not reproducible outside TF: the draws are inputs here.
"""
import numpy as np


def blacken_transparent_pixels(image):
    image = np.asarray(image, np.float64).copy()
    image[image[..., 3] == 0] = 0.0
    return image


def normalize(image):
    return image / 127.5 - 1.0


def adjust_hue(rgb, delta):
    rgb = np.asarray(rgb, np.float64)
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    vmax, vmin = rgb.max(-1), rgb.min(-1)
    c = vmax - vmin
    safe = np.where(c > 0, c, 1.0)
    h6 = np.where(vmax == r, (g - b) / safe, np.where(vmax == g, 2.0 + (b - r) / safe, 4.0 + (r - g) / safe))
    h6 = np.mod(h6 + 6.0 * delta, 6.0)
    x = c * (1.0 - np.abs(np.mod(h6, 2.0) - 1.0))
    sector = np.minimum(np.floor(h6).astype(int), 5)
    zero = np.zeros_like(c)
    table = [(c, x, zero), (x, c, zero), (zero, c, x), (zero, x, c), (x, zero, c), (c, zero, x)]
    out = np.zeros_like(rgb)
    for s, (rr, gg, bb) in enumerate(table):
        m = sector == s
        out[..., 0][m], out[..., 1][m], out[..., 2][m] = rr[m], gg[m], bb[m]
    out += vmin[..., None]
    return np.where((c > 0)[..., None], out, rgb)


def _round_half_away(v):
    return (np.sign(v) * np.floor(np.abs(v) + 0.5)).astype(int)


def translate_nearest(image, dy, dx):
    h, w = image.shape[:2]
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    # float32 coordinates, as the transform op computes them
    sy = _round_half_away((ys.astype(np.float32) - np.float32(dy)).astype(np.float64))
    sx = _round_half_away((xs.astype(np.float32) - np.float32(dx)).astype(np.float64))
    inside = (sy >= 0) & (sy < h) & (sx >= 0) & (sx < w)
    out = np.zeros_like(image)
    out[inside] = image[sy[inside], sx[inside]]
    return out


def augment_pair(first, second, delta, dy, dx):
    """augment_two (dataset_utils.py:95-103): the same hue rotation (shared seed) and the same translation for both"""
    res = []
    for img in (first, second):
        img = np.concatenate([adjust_hue(img[..., :3], delta), img[..., 3:]], axis=-1)
        res.append(translate_nearest(img, dy, dx))
    return res


def make_pair(source_u8, target_u8, aug_row=None, should_normalize=True):
    """load_image (dataset_utils.py:66-77) of both sprites + the optional augmentation + normalize_two (:106-107).
    aug_row = (apply, delta, dy, dx) or None.  Translation commutes with the per-pixel steps, hue with blackening (a
    blackened pixel is grey), so the order here equals the reference's load -> augment -> normalise."""
    s, t = blacken_transparent_pixels(source_u8), blacken_transparent_pixels(target_u8)
    if aug_row is not None and aug_row[0]:
        s, t = augment_pair(s, t, float(aug_row[1]), float(aug_row[2]), float(aug_row[3]))
    if should_normalize:
        s, t = normalize(s), normalize(t)
    return s, t
