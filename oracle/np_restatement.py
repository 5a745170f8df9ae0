"""Independent pure-numpy, index-level restatement of every op on the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- PARITY UNPINNED.

Purpose: pin the torch mappings used by ``reference_graph`` (F.conv2d / conv_transpose2d /
padding) to the *published* TensorFlow semantics, written out as explicit index formulas, and
state the closed-form backward passes the HIP kernels implement so they can be checked against
autograd.  Loops are slow: use at toy sizes only.

TF SAME padding (tf.nn.convolution docs): out = ceil(in / s); pad_total = max((out-1)*s + k - in, 0);
pad_before = pad_total // 2; pad_after = pad_total - pad_before  (the odd pixel goes after).
"""
import math
import numpy as np


def same_pads(size, k, s):
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2, total - total // 2


def conv2d_same(x, w, stride):
    """tf.nn.conv2d(x NHWC, w HWIO, strides=stride, padding='SAME') (networks.py:10-16,47-48,75-78)."""
    n, h, wd, ci = x.shape
    kh, kw, _, co = w.shape
    oh, pt, _ = same_pads(h, kh, stride)
    ow, pl, _ = same_pads(wd, kw, stride)
    y = np.zeros((n, oh, ow, co), x.dtype)
    for a in range(oh):
        for b in range(ow):
            for i in range(kh):
                for j in range(kw):
                    ih, iw = a * stride + i - pt, b * stride + j - pl
                    if 0 <= ih < h and 0 <= iw < wd:
                        y[:, a, b, :] += x[:, ih, iw, :] @ w[i, j]
    return y


def conv2d_transpose_same(x, w, stride):
    """keras Conv2DTranspose(padding='same') == tf.nn.conv2d_transpose == gradient of conv2d wrt its input
    (networks.py:26-27).  w has the Keras layout (kh, kw, Cout, Cin); output size = in * stride."""
    n, h, wd, ci = x.shape
    kh, kw, co, _ = w.shape
    oh_full, ow_full = h * stride, wd * stride
    _, pt, _ = same_pads(oh_full, kh, stride)      # pads of the forward conv that maps out -> in
    _, pl, _ = same_pads(ow_full, kw, stride)
    y = np.zeros((n, oh_full, ow_full, co), x.dtype)
    for a in range(h):
        for b in range(wd):
            for i in range(kh):
                for j in range(kw):
                    oh, ow = a * stride + i - pt, b * stride + j - pl
                    if 0 <= oh < oh_full and 0 <= ow < ow_full:
                        y[:, oh, ow, :] += x[:, a, b, :] @ w[i, j].T
    return y


def instance_norm(x, gamma, beta, eps=1e-3):
    """tfa InstanceNormalization == GroupNormalization(groups=C): biased moments over (H, W)."""
    mu = x.mean(axis=(1, 2), keepdims=True)
    var = x.var(axis=(1, 2), keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * gamma + beta


def instance_norm_backward(x, gamma, dy, eps=1e-3):
    """Closed form the HIP kernel implements. Returns dx, dgamma, dbeta."""
    mu = x.mean(axis=(1, 2), keepdims=True)
    r = 1.0 / np.sqrt(x.var(axis=(1, 2), keepdims=True) + eps)
    xh = (x - mu) * r
    dbeta = dy.sum(axis=(0, 1, 2))
    dgamma = (dy * xh).sum(axis=(0, 1, 2))
    m1 = dy.mean(axis=(1, 2), keepdims=True)
    m2 = (dy * xh).mean(axis=(1, 2), keepdims=True)
    dx = gamma * r * (dy - m1 - xh * m2)
    return dx, dgamma, dbeta


def leaky_relu(x, alpha=0.3):
    return np.where(x > 0, x, alpha * x)


def bce_from_logits(x, z):
    return np.mean(np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x))))


def rgbuv_histogram(image, size=64, sigma=0.02):
    """histogram.py:35-81 written as explicit per-pixel accumulation (no matmul)."""
    eps = 1e-6
    b = image.shape[0]
    dom = np.linspace(-3.0, 3.0, size).astype(image.dtype)
    img = image[..., :3] * 0.5 + 0.5
    I = img.reshape(b, -1, 3)
    hist = np.zeros((b, size, size, 3), image.dtype)
    combos = ((0, 1, 2), (1, 0, 2), (2, 0, 1))          # histogram.py:72-74
    for n in range(b):
        for p in range(I.shape[1]):
            px = I[n, p]
            iy = math.sqrt(float(px[0] ** 2 + px[1] ** 2 + px[2] ** 2) + eps)
            for c, (a, p1, p2) in enumerate(combos):
                u = math.log(px[a] + eps) - math.log(px[p1] + eps)
                v = math.log(px[a] + eps) - math.log(px[p2] + eps)
                ku = 1.0 / (1.0 + (u - dom) ** 2 / sigma ** 2)
                kv = 1.0 / (1.0 + (v - dom) ** 2 / sigma ** 2)
                hist[n, :, :, c] += iy * np.outer(ku, kv)
    return hist / hist.sum(axis=(1, 2, 3), keepdims=True)


def hellinger(y_true, y_pred):
    """histogram.py:84-89."""
    return math.sqrt(float(((np.sqrt(y_pred) - np.sqrt(y_true)) ** 2).sum())) / math.sqrt(2.0) / y_true.shape[0]


def hist_hellinger_backward(fake, real_hist, size=64, sigma=0.02, global_sq_sum=None, global_batch=None):
    """Closed-form d hellinger / d fake image -- what the HIP backward kernel computes.

    global_sq_sum / global_batch let a data-parallel shard plug in the all-reduced batch-wide
    sum of (sqrt p - sqrt q)^2 and the global batch size."""
    eps = 1e-6
    s2 = sigma ** 2
    b = fake.shape[0]
    dom = np.linspace(-3.0, 3.0, size).astype(fake.dtype)
    x = fake[..., :3].reshape(b, -1, 3) * 0.5 + 0.5
    iy = np.sqrt((x ** 2).sum(-1) + eps)                                   # (b, HW)
    combos = ((0, 1, 2), (1, 0, 2), (2, 0, 1))
    ku, kv, uu, vv = [], [], [], []
    raw = np.zeros((b, size, size, 3), fake.dtype)
    for c, (a, p1, p2) in enumerate(combos):
        u = np.log(x[..., a] + eps) - np.log(x[..., p1] + eps)
        v = np.log(x[..., a] + eps) - np.log(x[..., p2] + eps)
        k_u = 1.0 / (1.0 + (u[..., None] - dom) ** 2 / s2)                 # (b, HW, size)
        k_v = 1.0 / (1.0 + (v[..., None] - dom) ** 2 / s2)
        raw[..., c] = np.einsum("bp,bpi,bpj->bij", iy, k_u, k_v)
        ku.append(k_u); kv.append(k_v); uu.append(u); vv.append(v)
    T = raw.sum(axis=(1, 2, 3), keepdims=True)
    hn = raw / T
    D = np.sqrt(hn) - np.sqrt(real_hist)
    sq = float((D ** 2).sum()) if global_sq_sum is None else float(global_sq_sum)
    bg = b if global_batch is None else global_batch
    G = D / (2.0 * math.sqrt(2.0) * bg * math.sqrt(sq) * np.sqrt(hn))      # dL/dHn
    GH = (G - (G * hn).sum(axis=(1, 2, 3), keepdims=True)) / T             # dL/draw
    dx = np.zeros_like(x)
    diy = np.zeros_like(iy)
    for c, (a, p1, p2) in enumerate(combos):
        A = np.einsum("bij,bpj->bpi", GH[..., c], kv[c])                   # (b, HW, size)
        Bm = np.einsum("bij,bpi->bpj", GH[..., c], ku[c])
        diy += (A * ku[c]).sum(-1)
        du = (iy[..., None] * A * (-2.0 * (uu[c][..., None] - dom) / s2) * ku[c] ** 2).sum(-1)
        dv = (iy[..., None] * Bm * (-2.0 * (vv[c][..., None] - dom) / s2) * kv[c] ** 2).sum(-1)
        dx[..., a] += (du + dv) / (x[..., a] + eps)
        dx[..., p1] -= du / (x[..., p1] + eps)
        dx[..., p2] -= dv / (x[..., p2] + eps)
    dx += diy[..., None] * x / iy[..., None]
    dimg = np.zeros_like(fake)
    dimg[..., :3] = 0.5 * dx.reshape(fake.shape[:3] + (3,))
    return dimg


def keras_adam_step(theta, g, m, v, t, lr=2e-4, b1=0.5, b2=0.999, eps=1e-7):
    """keras OptimizerV2 Adam, t = iteration count after increment (pix2pix_model.py:28-29)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    return theta - lr_t * m / (np.sqrt(v) + eps), m, v
