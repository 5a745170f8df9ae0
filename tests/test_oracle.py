"""Pins the oracle: structural + analytic known answers, the independent numpy
restatement of TF semantics vs the torch mappings, and closed-form backward passes vs autograd."""
import math

import numpy as np
import pytest
import torch

from oracle import np_restatement as npr
from oracle import reference_graph as rg

F64 = torch.float64


def test_param_counts_match_reference_printout():
    # experiments.ipynb:198-199 prints 29,307,844 / 9,217
    assert rg.param_count(rg.generator_param_shapes(4, 4)) == 29_307_844
    assert rg.param_count(rg.discriminator_param_shapes(4)) == 9_217
    assert rg.param_count(rg.generator_param_shapes(1, 256)) == 29_437_888
    assert rg.param_count(rg.discriminator_param_shapes(1)) == 3_073


def test_layer_shapes_follow_networks_comments():
    rng = np.random.default_rng(0)
    Gp = rg.init_params(rg.generator_param_shapes(4, 4), rng, torch.float32)
    x = torch.zeros(1, 64, 64, 4)
    shapes = []
    h = x
    for i in range(1, 7):
        h = rg.unet_downsample(h, Gp, f"down{i}", i > 1)
        shapes.append(tuple(h.shape[1:]))
    assert shapes == [(32, 32, 64), (16, 16, 128), (8, 8, 256), (4, 4, 512), (2, 2, 512), (1, 1, 512)]
    masks = [torch.ones(s) for s in rg.dropout_mask_shapes(1, 64)]
    assert rg.unet_generator(Gp, x, masks, "tanh").shape == (1, 64, 64, 4)
    Dp = rg.init_params(rg.discriminator_param_shapes(4), rng, torch.float32)
    assert rg.patch_discriminator(Dp, x, x).shape == (1, 32, 32, 1)


@pytest.mark.parametrize("size,cin,cout", [(8, 3, 5), (2, 4, 6), (6, 1, 2)])
def test_conv_s2_matches_tf_same_semantics(size, cin, cout):
    rng = np.random.default_rng(1)
    x = rng.normal(size=(2, size, size, cin))
    w = rng.normal(size=(4, 4, cin, cout))
    ref = npr.conv2d_same(x, w, 2)
    got = rg.conv4x4_s2(torch.tensor(x), torch.tensor(w)).numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("size,cin,cout", [(8, 3, 2), (5, 4, 1)])
def test_conv_s1_pads_one_before_two_after(size, cin, cout):
    rng = np.random.default_rng(2)
    x = rng.normal(size=(2, size, size, cin))
    w = rng.normal(size=(4, 4, cin, cout))
    b = rng.normal(size=(cout,))
    assert npr.same_pads(size, 4, 1) == (size, 1, 2)
    ref = npr.conv2d_same(x, w, 1) + b
    got = rg.conv4x4_s1_bias(torch.tensor(x), torch.tensor(w), torch.tensor(b)).numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("size,cin,cout", [(4, 3, 5), (1, 4, 2), (3, 2, 2)])
def test_conv_transpose_matches_gradient_of_conv(size, cin, cout):
    rng = np.random.default_rng(3)
    x = rng.normal(size=(2, size, size, cin))
    w = rng.normal(size=(4, 4, cout, cin))          # Keras (kh,kw,Cout,Cin)
    ref = npr.conv2d_transpose_same(x, w, 2)
    got = rg.convT4x4_s2(torch.tensor(x), torch.tensor(w)).numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)
    # and it really is the adjoint of the forward conv with HWIO kernel w (I=cout of convT, O=cin)
    y = rng.normal(size=ref.shape)
    lhs = (ref * y).sum()
    rhs = (x * npr.conv2d_same(y, w, 2)).sum()
    assert abs(lhs - rhs) < 1e-9 * max(1.0, abs(lhs))


def test_instance_norm_and_closed_form_backward():
    rng = np.random.default_rng(4)
    x = rng.normal(size=(2, 4, 4, 3))
    gamma, beta = rng.normal(size=3), rng.normal(size=3)
    dy = rng.normal(size=x.shape)
    np.testing.assert_allclose(
        rg.instance_norm(torch.tensor(x), torch.tensor(gamma), torch.tensor(beta)).numpy(),
        npr.instance_norm(x, gamma, beta), rtol=1e-12, atol=1e-12)
    xt = torch.tensor(x, requires_grad=True)
    gt = torch.tensor(gamma, requires_grad=True)
    bt = torch.tensor(beta, requires_grad=True)
    (rg.instance_norm(xt, gt, bt) * torch.tensor(dy)).sum().backward()
    dx, dg, db = npr.instance_norm_backward(x, gamma, dy)
    np.testing.assert_allclose(dx, xt.grad.numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(dg, gt.grad.numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(db, bt.grad.numpy(), rtol=1e-10, atol=1e-12)


def test_instance_norm_1x1_is_beta_with_zero_input_grad():
    x = torch.randn(2, 1, 1, 5, dtype=F64, requires_grad=True)
    beta = torch.randn(5, dtype=F64)
    y = rg.instance_norm(x, torch.ones(5, dtype=F64), beta)
    assert torch.equal(y, beta.expand_as(y))
    (y * torch.randn_like(y)).sum().backward()
    assert torch.count_nonzero(x.grad) == 0


def test_bce_known_answers():
    assert abs(float(rg.bce_from_logits(torch.zeros(3, 4, 4, 1, dtype=F64), 1.0)) - math.log(2)) < 1e-15
    x = np.random.default_rng(5).normal(size=(2, 3, 3, 1)) * 4
    assert abs(float(rg.bce_from_logits(torch.tensor(x), 0.0)) - npr.bce_from_logits(x, 0.0)) < 1e-14


def test_discriminator_zero_kernels_gives_bias():
    Dp = rg.init_params(rg.discriminator_param_shapes(4), np.random.default_rng(0), F64)
    Dp["last.kernel"].zero_()
    Dp["last.bias"].fill_(0.37)
    out = rg.patch_discriminator(Dp, torch.randn(1, 8, 8, 4, dtype=F64), torch.randn(1, 8, 8, 4, dtype=F64))
    assert torch.allclose(out, torch.full_like(out, 0.37))


def test_cce_uniform_is_ln256():
    logits = torch.zeros(1, 2, 2, 256, dtype=F64)
    idx = torch.randint(0, 256, (1, 2, 2, 1))
    assert abs(float(rg.categorical_crossentropy_from_logits(logits, idx)) - math.log(256)) < 1e-12


def test_histogram_matches_numpy_and_transparent_known_answer():
    rng = np.random.default_rng(6)
    src, _ = rg.synthetic_rgba_batch(rng, 2, 8, palette_size=6)
    h_t = rg.rgbuv_histogram(torch.tensor(src, dtype=F64)).numpy()
    h_n = npr.rgbuv_histogram(src.astype(np.float64))
    np.testing.assert_allclose(h_t, h_n, rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(h_t.sum(axis=(1, 2, 3)), 1.0, rtol=1e-12)
    # all-transparent image: u=v=0 everywhere -> three identical planes, maxima at bins {31,32}^2
    h = rg.rgbuv_histogram(torch.full((1, 4, 4, 4), -1.0, dtype=F64))[0]
    assert torch.allclose(h[..., 0], h[..., 1]) and torch.allclose(h[..., 0], h[..., 2])
    peak = h[..., 0].max()
    for i in (31, 32):
        for j in (31, 32):
            assert abs(float(h[i, j, 0] - peak)) < 1e-15
    k = 1.0 / (1.0 + np.linspace(-3, 3, 64) ** 2 / 0.02 ** 2)
    np.testing.assert_allclose(h[..., 0].numpy() / float(h[..., 0].sum()), np.outer(k, k) / np.outer(k, k).sum(),
                               rtol=1e-6)


def test_hellinger_known_answers():
    a = torch.zeros(1, 64, 64, 3, dtype=F64); a[0, 0, 0, 0] = 1.0
    b = torch.zeros(1, 64, 64, 3, dtype=F64); b[0, 5, 5, 1] = 1.0
    assert abs(float(rg.hellinger_loss(a, b)) - 1.0) < 1e-15
    assert float(rg.hellinger_loss(a, a)) == 0.0
    assert abs(npr.hellinger(a.numpy(), b.numpy()) - 1.0) < 1e-15


def test_histogram_closed_form_backward_matches_autograd():
    rng = np.random.default_rng(7)
    src, tgt = rg.synthetic_rgba_batch(rng, 2, 8, palette_size=5)
    fake = np.clip(src + rng.normal(scale=0.05, size=src.shape), -1, 1).astype(np.float64)
    real_h = rg.rgbuv_histogram(torch.tensor(tgt, dtype=F64))
    ft = torch.tensor(fake, requires_grad=True)
    rg.hellinger_loss(real_h, rg.rgbuv_histogram(ft)).backward()
    got = npr.hist_hellinger_backward(fake, real_h.numpy())
    np.testing.assert_allclose(got, ft.grad.numpy(), rtol=1e-8, atol=1e-12)
    assert np.count_nonzero(got[..., 3]) == 0


def test_keras_adam_first_step_known_answer():
    g = np.array([1e-2, -3.0, 5e-4])
    th, m, v = npr.keras_adam_step(np.zeros(3), g, np.zeros(3), np.zeros(3), 1)
    a = math.sqrt(1 - 0.999) * np.abs(g)
    np.testing.assert_allclose(th, -2e-4 * np.sign(g) * a / (a + 1e-7), rtol=1e-12)   # eps NOT bias-corrected
    p = {"w": torch.zeros(3, dtype=F64)}
    newp, m2, v2 = rg.keras_adam(p, {"w": torch.tensor(g)}, rg.zeros_like_params(p), rg.zeros_like_params(p), 1)
    np.testing.assert_allclose(newp["w"].numpy(), th, rtol=1e-13)


def test_train_step_structure_g_grad_flows_through_d_and_down6_dead():
    rng = np.random.default_rng(8)
    Gp = rg.perturb_affine(rg.init_params(rg.generator_param_shapes(4, 4), rng, torch.float32), rng)
    Dp = rg.perturb_affine(rg.init_params(rg.discriminator_param_shapes(4), rng, torch.float32), rng)
    src, tgt = rg.synthetic_rgba_batch(rng, 2, 64)
    masks = [torch.tensor(rng.integers(0, 2, size=s).astype(np.float32)) for s in rg.dropout_mask_shapes(2, 64)]
    out = rg.train_step_rgba(Gp, Dp, torch.tensor(src), torch.tensor(tgt), masks, lambda_l1=100.0)
    assert len(out["g_loss"]) == 3 and len(out["d_loss"]) == 3
    assert abs(out["g_loss"][0] - (out["g_loss"][1] + 100.0 * out["g_loss"][2])) < 1e-4
    assert abs(out["d_loss"][0] - (out["d_loss"][1] + out["d_loss"][2])) < 1e-6
    # IN on a 1x1 map kills the gradient of down6's conv
    assert float(out["g_grads"]["down6.kernel"].abs().max()) == 0.0
    assert float(out["g_grads"]["down6.beta"].abs().max()) > 0.0
    assert float(out["g_grads"]["down1.kernel"].abs().max()) > 0.0
