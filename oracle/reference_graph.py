"""torch-CPU restatement of the reference's Pix2Pix graphs, losses and train steps.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- PARITY UNPINNED.

Every function cites the reference file:line it follows (paths relative to
/root/reference).  Third-party constants that differ from torch defaults
(SURVEY.md section 7 "hard parts"):
  * tfa InstanceNormalization epsilon = 1e-3, biased variance   (networks.py:18,29)
  * keras LeakyReLU alpha = 0.3                                  (networks.py:19)
  * keras Dropout(0.5) is always on (training=True everywhere)   (pix2pix_model.py:60,67)
  * keras Adam: eps = 1e-7 added to sqrt(v) *outside* the bias correction
  * TF SAME padding for 4x4 stride 1 is (1 before, 2 after)
  * Conv2DTranspose kernel layout (kh, kw, Cout, Cin)
  * discriminator input order [target, source]                   (networks.py:45)

All tensors are NHWC like the reference; conv kernels keep the Keras layouts.
Gradients come from torch autograd over this forward restatement, exactly as
the reference's come from tf.GradientTape over its forward graph.
"""
from collections import OrderedDict
import math

import numpy as np
import torch
import torch.nn.functional as F

IN_EPS = 1e-3          # tfa InstanceNormalization default epsilon
LEAKY_ALPHA = 0.3      # keras LeakyReLU default alpha
ADAM_LR = 2e-4         # pix2pix_model.py:28-29
ADAM_BETA1 = 0.5       # pix2pix_model.py:28-29
ADAM_BETA2 = 0.999     # keras default
ADAM_EPS = 1e-7        # keras default
MAX_PALETTE_SIZE = 256  # configuration.py:31

DOWN_FILTERS = (64, 128, 256, 512, 512, 512)     # networks.py:57-64
UP_FILTERS = (512, 512, 256, 128, 64, 32)        # networks.py:66-73
UP_DROPOUT = (True, True, True, False, False, False)


# --------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------
def generator_param_shapes(in_ch, out_ch):
    """Keras variable order and shapes of UnetGenerator (networks.py:53-98)."""
    shapes = OrderedDict()
    c = in_ch
    skip_ch = []
    for i, f in enumerate(DOWN_FILTERS, start=1):
        shapes[f"down{i}.kernel"] = (4, 4, c, f)              # Conv2D HWIO, no bias (networks.py:10-16)
        if i > 1:                                             # apply_batchnorm=False on down1 (networks.py:58)
            shapes[f"down{i}.gamma"] = (f,)
            shapes[f"down{i}.beta"] = (f,)
        skip_ch.append(f)
        c = f
    skips = list(reversed(skip_ch[:-1])) + [in_ch]            # networks.py:89,92
    for i, (f, s) in enumerate(zip(UP_FILTERS, skips), start=1):
        shapes[f"up{i}.kernel"] = (4, 4, f, c)                # Conv2DTranspose (kh,kw,Cout,Cin) (networks.py:26-27)
        shapes[f"up{i}.gamma"] = (f,)
        shapes[f"up{i}.beta"] = (f,)
        c = f + s                                             # concat [x, skip] (networks.py:94)
    shapes["last.kernel"] = (4, 4, c, out_ch)                 # networks.py:75-78
    shapes["last.bias"] = (out_ch,)
    return shapes


def discriminator_param_shapes(in_ch):
    """PatchDiscriminator variables (networks.py:39-50)."""
    return OrderedDict([
        ("down.kernel", (4, 4, 2 * in_ch, 64)),               # networks.py:46, no IN, no bias
        ("last.kernel", (4, 4, 64, 1)),                       # networks.py:47-48
        ("last.bias", (1,)),
    ])


def init_params(shapes, rng, dtype=torch.float64):
    """kernels N(0, 0.02) (networks.py:7,24,40,54), biases 0, gamma 1, beta 0."""
    p = OrderedDict()
    for name, shp in shapes.items():
        if name.endswith(".kernel"):
            p[name] = torch.tensor(rng.normal(0.0, 0.02, size=shp), dtype=dtype)
        elif name.endswith(".gamma"):
            p[name] = torch.ones(shp, dtype=dtype)
        else:
            p[name] = torch.zeros(shp, dtype=dtype)
    return p


def perturb_affine(p, rng, scale=0.1):
    """Oracle-only helper: move gamma/beta/bias off their init (1/0/0) so parity tests exercise every
    gradient path (at the reference's init, beta6 = 0 makes up1's pre-activation exactly 0 and
    relu'(0) = 0 silences the whole bottleneck)."""
    for k, x in p.items():
        if k.endswith((".gamma", ".beta", ".bias")):
            x += torch.tensor(rng.normal(0.0, scale, size=tuple(x.shape)), dtype=x.dtype)
    return p


def param_count(shapes):
    return int(sum(int(np.prod(s)) for s in shapes.values()))


# --------------------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------------------
_QUANT = None      # oracle-only: emulate the engine's bf16 storage points (straight-through rounding)


class storage_dtype:
    """Context manager: `with storage_dtype(torch.bfloat16): ...` rounds weights, conv outputs, block outputs and
    images through that dtype at the points where the HIP engine stores them, gradients passing straight
    through.  Used to check the bf16 throughput mode against the same graph with the same storage."""

    def __init__(self, dt):
        self.dt = dt

    def __enter__(self):
        global _QUANT
        self.prev, _QUANT = _QUANT, self.dt

    def __exit__(self, *a):
        global _QUANT
        _QUANT = self.prev


def _q(x):
    if _QUANT is None:
        return x
    return x + (x.detach().to(_QUANT).to(x.dtype) - x.detach())


def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


def conv4x4_s2(x, kernel):
    """Conv2D(f, 4, strides=2, padding='same', use_bias=False) (networks.py:10-16).
    TF SAME on an even input with k=4,s=2 pads (1,1); kernel HWIO."""
    return _q(_nhwc(F.conv2d(_nchw(x), _q(kernel).permute(3, 2, 0, 1).contiguous(), stride=2, padding=1)))


def convT4x4_s2(x, kernel):
    """Conv2DTranspose(f, 4, strides=2, padding='same', use_bias=False) (networks.py:26-27).
    Keras kernel (kh,kw,Cout,Cin); out[n,oh,ow,co] = sum_{oh=2ih+kh-1} x[n,ih,iw,ci] W[kh,kw,co,ci]."""
    return _q(_nhwc(F.conv_transpose2d(_nchw(x), _q(kernel).permute(3, 2, 0, 1).contiguous(), stride=2, padding=1)))


def conv4x4_s1_bias(x, kernel, bias):
    """Conv2D(f, 4, padding='same') with bias (networks.py:47-48,75-78). TF SAME pads 1 before, 2 after."""
    xp = F.pad(_nchw(x), (1, 2, 1, 2))
    return _q(_nhwc(F.conv2d(xp, _q(kernel).permute(3, 2, 0, 1).contiguous(), bias=bias)))


def instance_norm(x, gamma, beta):
    """tfa InstanceNormalization (networks.py:18,29): per (n,c) biased moments over H,W, eps=1e-3."""
    mu = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mu) ** 2).mean(dim=(1, 2), keepdim=True)
    return (x - mu) * torch.rsqrt(var + IN_EPS) * gamma + beta


def leaky_relu(x):
    return torch.where(x > 0, x, LEAKY_ALPHA * x)     # keras LeakyReLU() (networks.py:19)


def dropout(x, mask):
    """keras Dropout(0.5) in training mode (networks.py:32): keep w.p. 0.5, scale by 2. mask is 0/1."""
    return x * mask * 2.0


def unet_downsample(x, p, name, apply_norm):
    """networks.py:7-21."""
    x = conv4x4_s2(x, p[f"{name}.kernel"])
    if apply_norm:
        x = instance_norm(x, p[f"{name}.gamma"], p[f"{name}.beta"])
    return _q(leaky_relu(x))


def unet_upsample(x, p, name, mask):
    """networks.py:24-36: convT -> IN -> [Dropout] -> ReLU."""
    x = convT4x4_s2(x, p[f"{name}.kernel"])
    x = instance_norm(x, p[f"{name}.gamma"], p[f"{name}.beta"])
    if mask is not None:
        x = dropout(x, mask)
    return _q(torch.relu(x))


def unet_generator(p, x, masks, last_activation, aux=None):
    """UnetGenerator forward (networks.py:80-98). masks: list of three 0/1 tensors for up1..up3.
    aux (oracle-only, dict): receives "fake_unrounded" = tanh output before the storage rounding (storage_dtype): the
    engine hands the histogram loss that f32 copy in every mode."""
    inputs = x
    skips = []
    for i in range(1, 7):
        x = unet_downsample(x, p, f"down{i}", apply_norm=(i > 1))
        skips.append(x)
    skips = list(reversed(skips[:-1])) + [inputs]
    for i, skip in enumerate(skips, start=1):
        m = masks[i - 1] if UP_DROPOUT[i - 1] else None
        x = unet_upsample(x, p, f"up{i}", m)
        x = torch.cat([x, skip], dim=-1)
    x = conv4x4_s1_bias(x, p["last.kernel"], p["last.bias"])
    if last_activation == "tanh":
        y = torch.tanh(x)
        if aux is not None:
            aux["fake_unrounded"] = y
        return _q(y)
    if last_activation == "softmax":
        return torch.softmax(x, dim=-1)
    if last_activation == "logits":          # oracle-only: pre-softmax values for the log-softmax CCE
        return x
    raise ValueError(last_activation)


def patch_discriminator(p, target, source):
    """PatchDiscriminator forward (networks.py:45-48): concat [target, source] -> down(64, no IN) -> conv s1."""
    x = torch.cat([target, source], dim=-1)
    x = _q(leaky_relu(conv4x4_s2(x, p["down.kernel"])))
    return conv4x4_s1_bias(x, p["last.kernel"], p["last.bias"])


def dropout_mask_shapes(batch, img_size):
    """Shapes of the three dropout masks (up1..up3 outputs, networks.py:67-69)."""
    s = img_size // 64
    return [(batch, 2 * s, 2 * s, 512), (batch, 4 * s, 4 * s, 512), (batch, 8 * s, 8 * s, 256)]


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------
def bce_from_logits(logits, label):
    """keras BinaryCrossentropy(from_logits=True), mean over all elements (pix2pix_model.py:19)."""
    z = float(label)
    return (torch.clamp(logits, min=0) - logits * z + torch.log1p(torch.exp(-torch.abs(logits)))).mean()


def generator_loss(fake_pred, fake, real, lambda_l1):
    """pix2pix_model.py:44-49."""
    adv = bce_from_logits(fake_pred, 1.0)
    l1 = (real - fake).abs().mean()
    return adv + lambda_l1 * l1, adv, l1


def discriminator_loss(real_pred, fake_pred):
    """pix2pix_model.py:51-56."""
    real = bce_from_logits(real_pred, 1.0)
    fake = bce_from_logits(fake_pred, 0.0)
    return fake + real, real, fake


def rgbuv_histogram(image, size=64, sigma=0.02, method="inverse-quadratic"):
    """calculate_rgbuv_histogram (histogram.py:35-81); method "RBF", "inverse-quadratic", or anything else: the reference applies no
    kernel function then (histogram.py:20-27 has two branches)."""
    eps = 1e-6
    sigma_sqr = sigma ** 2
    dom = torch.linspace(-3.0, 3.0, size, dtype=image.dtype)              # histogram.py:55
    img = image * 0.5 + 0.5                                                # :58
    I = img[..., :3].reshape(image.shape[0], -1, 3)                        # :61,64
    Iy = torch.sqrt((I ** 2).sum(-1) + eps).unsqueeze(-1)                  # :65-66

    def component(comp, p1, p2):                                           # histogram.py:4-32
        Iu = (torch.log(comp + eps) - torch.log(p1 + eps)).unsqueeze(-1)
        Iv = (torch.log(comp + eps) - torch.log(p2 + eps)).unsqueeze(-1)
        du, dv = (Iu - dom) ** 2 / sigma_sqr, (Iv - dom) ** 2 / sigma_sqr        # :20-21
        if method == "RBF":                                                       # :22-24
            du, dv = torch.exp(-du), torch.exp(-dv)
        elif method == "inverse-quadratic":                                       # :25-27
            du, dv = 1.0 / (1.0 + du), 1.0 / (1.0 + dv)
        a = (Iy * du).transpose(1, 2)
        return a @ dv

    r, g, b = I[..., 0], I[..., 1], I[..., 2]
    h = torch.stack([component(r, g, b), component(g, r, b), component(b, r, g)], dim=-1)   # :72-75
    return h / h.sum(dim=(1, 2, 3), keepdim=True)                          # :78-79


def hellinger_loss(y_true, y_pred):
    """histogram.py:84-89: sqrt over the WHOLE batch sum, divided by the batch size."""
    b = y_true.shape[0]
    return (1.0 / math.sqrt(2.0)) * torch.sqrt(((torch.sqrt(y_pred) - torch.sqrt(y_true)) ** 2).sum()) / b


def categorical_crossentropy_from_logits(logits, target_idx):
    """keras CategoricalCrossentropy(from_logits=False) on a softmax output (pix2pix_model.py:265,274).
    Keras 2.9 uses the cached logits of a softmax activation -> softmax_cross_entropy_with_logits;
    the documented fallback clip(p,1e-7,1-1e-7) agrees to <1e-6 unless a target prob < 1e-7
    (SURVEY.md section 8a A9).  Mean over all pixels."""
    logp = torch.log_softmax(logits, dim=-1)
    return -(torch.gather(logp, -1, target_idx.long()).squeeze(-1)).mean()


# --------------------------------------------------------------------------------------
# optimizer
# --------------------------------------------------------------------------------------
def keras_adam(params, grads, m, v, t, lr=ADAM_LR, b1=ADAM_BETA1, b2=ADAM_BETA2, eps=ADAM_EPS):
    """One keras OptimizerV2 Adam step (pix2pix_model.py:28-29,81-83); t is the step index AFTER increment.
    theta -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)."""
    lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    new_p, new_m, new_v = OrderedDict(), OrderedDict(), OrderedDict()
    for k in params:
        g = grads[k]
        new_m[k] = b1 * m[k] + (1.0 - b1) * g
        new_v[k] = b2 * v[k] + (1.0 - b2) * g * g
        new_p[k] = params[k] - lr_t * new_m[k] / (torch.sqrt(new_v[k]) + eps)
    return new_p, new_m, new_v


def zeros_like_params(p):
    return OrderedDict((k, torch.zeros_like(x)) for k, x in p.items())


# --------------------------------------------------------------------------------------
# train steps
# --------------------------------------------------------------------------------------
def _leaf(p):
    return OrderedDict((k, x.detach().clone().requires_grad_(True)) for k, x in p.items())


def train_step_rgba(Gp, Dp, source, real, masks, lambda_l1, lambda_hist=None, global_batch=None):
    """Pix2PixModel.train_step (pix2pix_model.py:62-89) and the histogram variant (:242-250).

    Returns dict(g_loss=(total, adv, l1[, hist]), d_loss=(total, real, fake), g_grads, d_grads, fake).
    Both gradients are taken at the same (pre-update) weights; G's gradient flows through D.
    """
    Gl, Dl = _leaf(Gp), _leaf(Dp)
    real_unrounded = real
    source, real = _q(source), _q(real)
    aux = {}
    fake = unet_generator(Gl, source, masks, "tanh", aux)                # :67
    real_pred = patch_discriminator(Dl, real, source)                     # :69
    fake_pred = patch_discriminator(Dl, fake, source)                     # :70  (fake not detached)
    g_total, adv, l1 = generator_loss(fake_pred, fake, real, lambda_l1)  # :72
    g_loss = [g_total, adv, l1]
    if lambda_hist is not None:                                           # pix2pix_model.py:242-250
        # f32 arithmetic on f32 images in every storage mode: identical to (real, fake) without storage_dtype
        hist = hellinger_loss(rgbuv_histogram(real_unrounded), rgbuv_histogram(aux["fake_unrounded"]))
        g_total = g_total + lambda_hist * hist
        g_loss = [g_total, adv, l1, hist]
    d_total, d_real, d_fake = discriminator_loss(real_pred, fake_pred)    # :75
    g_grads = torch.autograd.grad(g_total, list(Gl.values()), retain_graph=True, allow_unused=True)   # :78
    d_grads = torch.autograd.grad(d_total, list(Dl.values()), allow_unused=True)                      # :79
    zg = lambda g, x: torch.zeros_like(x) if g is None else g
    return dict(
        g_loss=tuple(float(x.detach()) for x in g_loss),
        d_loss=(float(d_total.detach()), float(d_real.detach()), float(d_fake.detach())),
        g_grads=OrderedDict((k, zg(g, Gl[k]).detach()) for k, g in zip(Gl, g_grads)),
        d_grads=OrderedDict((k, zg(g, Dl[k]).detach()) for k, g in zip(Dl, d_grads)),
        fake=fake.detach(), real_pred=real_pred.detach(), fake_pred=fake_pred.detach())


def train_step_indexed(Gp, Dp, source_idx, real_idx, masks, lambda_segmentation):
    """Pix2PixIndexedModel.train_step (pix2pix_model.py:295-325).

    source_idx / real_idx: int (B,S,S,1) palette indices.  G = UnetGenerator(1, 256, softmax); the
    discriminator sees argmax indices cast to float un-normalised (0..255), so no gradient flows D->G;
    lambda_l1 is hard-wired to 0 (:263) so G learns only from lambda_seg * CCE.
    """
    dt = next(iter(Gp.values())).dtype
    Gl, Dl = _leaf(Gp), _leaf(Dp)
    src = source_idx.to(dt)
    real = real_idx.to(dt)
    logits = unet_generator(Gl, src, masks, "logits")                     # generate_with_probs :289-293
    probs = torch.softmax(logits, dim=-1)
    fake_idx = torch.argmax(probs, dim=-1, keepdim=True)                  # :292 ties -> lowest index
    real_pred = patch_discriminator(Dl, real, src)                        # :305
    fake_pred = patch_discriminator(Dl, fake_idx.to(dt), src)             # :306
    one_hot = F.one_hot(real_idx.squeeze(-1).long(), MAX_PALETTE_SIZE).to(dt)   # :300-301
    seg = categorical_crossentropy_from_logits(logits, real_idx)          # :274
    adv = bce_from_logits(fake_pred, 1.0)
    l1 = (one_hot - probs).abs().mean()                                   # lambda_l1 = 0 (:263): reported only
    g_total = adv + 0.0 * l1 + lambda_segmentation * seg                  # :273-278
    d_total, d_real, d_fake = discriminator_loss(real_pred, fake_pred)
    g_grads = torch.autograd.grad(g_total, list(Gl.values()), retain_graph=True, allow_unused=True)
    d_grads = torch.autograd.grad(d_total, list(Dl.values()), allow_unused=True)
    zg = lambda g, x: torch.zeros_like(x) if g is None else g
    return dict(
        g_loss=(float(g_total.detach()), float(adv.detach()), float(l1.detach()), float(seg.detach())),
        d_loss=(float(d_total.detach()), float(d_real.detach()), float(d_fake.detach())),
        g_grads=OrderedDict((k, zg(g, Gl[k]).detach()) for k, g in zip(Gl, g_grads)),
        d_grads=OrderedDict((k, zg(g, Dl[k]).detach()) for k, g in zip(Dl, d_grads)),
        fake_idx=fake_idx.to(torch.int32), probs=probs.detach(), logits=logits.detach(),
        real_pred=real_pred.detach(), fake_pred=fake_pred.detach())


# --------------------------------------------------------------------------------------
# synthetic batches
# --------------------------------------------------------------------------------------
def synthetic_rgba_batch(rng, batch, img_size, palette_size=None):
    """Sprite-like RGBA batch in [-1,1]: per image a palette of P opaque colours, pixels transparent
    (0,0,0,0) w.p. 0.835, source/target share the palette (dataset_utils.py:11-20,39-48 value contract)."""
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
    """Indexed batch: int32 (B,S,S,1) with 0 w.p. 0.835 else U{1..P-1}; palette padded with hot pink
    (configuration.py:31-32, io_utils.py:50-63)."""
    def draw():
        opaque = rng.random((batch, img_size, img_size, 1)) >= 0.835
        idx = rng.integers(1, palette_size, size=(batch, img_size, img_size, 1))
        return np.where(opaque, idx, 0).astype(np.int32)
    pal = np.tile(np.array([255, 0, 220, 255], np.int32), (batch, MAX_PALETTE_SIZE, 1))
    pal[:, :palette_size, :3] = rng.integers(0, 256, size=(batch, palette_size, 3))
    pal[:, :palette_size, 3] = 255
    pal[:, 0] = 0
    return draw(), draw(), pal
