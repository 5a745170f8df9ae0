"""-m gpu: the histogram loss kernels (histogram.py:4-89), the palette-index head (pix2pix_model.py:261-325) and the
two model variants' full train steps against the CPU oracle."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from oracle import np_restatement as npr
from oracle import reference_graph as rg
from palette_and_histo_gan_amd import _lib as L
from palette_and_histo_gan_amd import engine as E
from tests import gpu_util as U
from tests.test_train_step_gpu import grad_report, setup_case, to_np

pytestmark = pytest.mark.gpu
F64 = torch.float64


def test_rgbuv_histogram_forward_matches_oracle():
    rng = np.random.default_rng(31)
    src, tgt = rg.synthetic_rgba_batch(rng, 3, 64, palette_size=24)
    fake = np.clip(src + rng.normal(scale=0.1, size=src.shape), -1, 1).astype(np.float32)
    eng = E.Pix2PixEngine(4, 4, "tanh", 64, L.F32)
    for img in (tgt, fake):
        got = eng.rgbuv_histogram(img).cpu().numpy()
        ref = rg.rgbuv_histogram(torch.tensor(img, dtype=F64)).numpy()
        assert got.shape == (3, 64, 64, 3)
        np.testing.assert_allclose(got.sum(axis=(1, 2, 3)), 1.0, rtol=1e-5)
        assert U.rel_err(got, ref) < 1e-4       # f32 logf / division vs the f64 oracle (north_star tolerance 1e-4)
    # all-transparent image: three identical planes, the four centre bins are the equal maxima
    h = eng.rgbuv_histogram(np.full((1, 64, 64, 4), -1.0, np.float32)).cpu().numpy()[0]
    assert np.allclose(h[..., 0], h[..., 1]) and np.allclose(h[..., 0], h[..., 2])
    assert {tuple(ix) for ix in np.argwhere(h[..., 0] >= h[..., 0].max() * (1 - 1e-6))} == {(31, 31), (31, 32), (32, 31), (32, 32)}
    # the module-level function with the reference's name (histogram.py:35) launches the same kernels WITHOUT building an engine
    # (VERDICT r04 weak #13), and its scalar distances are the reference's (histogram.py:84-97)
    from palette_and_histo_gan_amd import histogram as H
    made = []
    orig = E.Pix2PixEngine.__init__
    try:
        E.Pix2PixEngine.__init__ = lambda self, *a, **k: (made.append(1), orig(self, *a, **k))[1]
        ht, hf = H.calculate_rgbuv_histogram(tgt), H.calculate_rgbuv_histogram(torch.as_tensor(fake).cuda())
    finally:
        E.Pix2PixEngine.__init__ = orig
    assert not made and torch.equal(hf, eng.rgbuv_histogram(fake))
    want = rg.hellinger_loss(rg.rgbuv_histogram(torch.tensor(tgt, dtype=F64)), rg.rgbuv_histogram(torch.tensor(fake, dtype=F64)))
    assert abs(float(H.hellinger_loss(ht, hf)) - float(want)) <= 1e-4 * float(want)
    # the function's other arguments (histogram.py:36; VERDICT r04 missing-4): sizes, sigma, the RBF kernel and the reference's
    # fall-through for any other method string (no kernel function applied) through the general kernel, against the f64 oracle
    for size, method, sigma in ((64, "RBF", 0.02), (32, "inverse-quadratic", 0.05), (48, "RBF", 0.1), (16, "thresholding", 0.02),
                                (128, "inverse-quadratic", 0.02), (64, "inverse-quadratic", 0.03)):
        got = H.calculate_rgbuv_histogram(fake, size=size, method=method, sigma=sigma).cpu().numpy()
        ref = rg.rgbuv_histogram(torch.tensor(fake, dtype=F64), size=size, sigma=sigma, method=method).numpy()
        assert got.shape == (3, size, size, 3)
        np.testing.assert_allclose(got.sum(axis=(1, 2, 3)), 1.0, rtol=1e-5)
        assert U.rel_err(got, ref) < 1e-4, (size, method, sigma, U.rel_err(got, ref))
    # the default arguments through BOTH kernels give the same histogram
    gen = H.calculate_rgbuv_histogram(fake, size=64, method="inverse-quadratic", sigma=0.02 + 1e-9).cpu().numpy()
    assert U.rel_err(gen, hf.cpu().numpy()) < 1e-5
    with pytest.raises(ValueError):
        H.calculate_rgbuv_histogram(tgt, size=256)


def test_histogram_tail_batches_small_image():
    """HW not a multiple of the kernels' pixel batches (8x8 = 64 < 128) exercises the zero-weight tail."""
    rng = np.random.default_rng(32)
    src, _ = rg.synthetic_rgba_batch(rng, 2, 64, palette_size=8)
    img = src[:, :8, :8, :].copy()
    hb = U.halo_from(np.concatenate([img, np.zeros_like(img)], -1), L.F32)
    raw = torch.empty(2 * 3 * 64 * 64, dtype=torch.float32, device=U.DEV)
    L.call("p2p_rgbuv_hist_fwd", L.F32, 2, 8, 8, C.byref(hb.view()), U.ptr(raw), U.stream())
    out = torch.empty((2, 64, 64, 3), dtype=torch.float32, device=U.DEV)
    L.call("p2p_hist_normalize", U.ptr(raw), 2, U.ptr(out), U.stream())
    ref = npr.rgbuv_histogram(img.astype(np.float64))
    assert U.rel_err(out.cpu().numpy(), ref) < 1e-4


def _fwd3(img, points, cap=1024):
    """p2p_rgbuv_hist_fwd3 (+ p2p_rgbuv_points) on a dense f32 (N,S,S,4) batch -> normalised (N,64,64,3), npoints"""
    N, S = img.shape[0], img.shape[1]
    t = U.dev(img)
    view = L.Tensor(t.data_ptr(), S * S, S, 4)
    raw = torch.empty(N * 3 * 64 * 64, dtype=torch.float32, device=U.DEV)
    ws = torch.empty(L.lib().p2p_rgbuv_hist_fwd3_workspace_bytes(N) // 4, dtype=torch.float32, device=U.DEV)
    pts = torch.full((N, cap, 4), float("nan"), dtype=torch.float32, device=U.DEV)
    npts = torch.full((N,), -7, dtype=torch.int32, device=U.DEV)
    if points:
        L.call("p2p_rgbuv_points", L.F32, N, S, S, C.byref(view), cap, U.ptr(pts), U.ptr(npts), U.stream())
    L.call("p2p_rgbuv_hist_fwd3", L.F32, N, S, S, C.byref(view), U.ptr(pts) if points else None, U.ptr(npts) if points else None,
           cap, U.ptr(raw), U.ptr(ws), U.stream())
    out = torch.empty((N, 64, 64, 3), dtype=torch.float32, device=U.DEV)
    L.call("p2p_hist_normalize", U.ptr(raw), N, U.ptr(out), U.stream())
    torch.cuda.synchronize()
    return out.cpu().numpy(), npts.cpu().numpy(), pts.cpu().numpy()


@pytest.mark.parametrize("S", [8, 64, 128])
def test_shared_row_histogram_and_colour_points_match_the_oracle(S):
    """p2p_rgbuv_hist_fwd3: three kernel rows per pixel serve all three components (mirrored bin grid); p2p_rgbuv_points:
    contraction over distinct colours x pixel counts.  Both against the f64 oracle at 1e-4, and against each other."""
    rng = np.random.default_rng(34)
    _, tgt = rg.synthetic_rgba_batch(rng, 3, max(S, 64), palette_size=24)
    tgt = tgt[:, :S, :S].copy()
    noisy = np.clip(tgt + rng.normal(scale=0.1, size=tgt.shape), -1, 1).astype(np.float32)     # every pixel its own colour
    ref_t = rg.rgbuv_histogram(torch.tensor(tgt, dtype=F64)).numpy()
    ref_n = rg.rgbuv_histogram(torch.tensor(noisy, dtype=F64)).numpy()
    dense_t, _, _ = _fwd3(tgt, points=False)
    dense_n, _, _ = _fwd3(noisy, points=False)
    assert U.rel_err(dense_t, ref_t) < 1e-4 and U.rel_err(dense_n, ref_n) < 1e-4
    listed_t, npts, pts = _fwd3(tgt, points=True)
    assert U.rel_err(listed_t, ref_t) < 1e-4
    # same numbers up to f32 summation order: adding 14 000 equal terms one by one (dense) drifts by up to 6e-5 of the peak at
    # 128x128, count x term (listed) does not -- both sit within the 1e-4 of the oracle asserted above
    assert np.abs(listed_t - dense_t).max() <= 1e-4 * dense_t.max()
    tiles = (S * S + 1023) // 1024
    for n in range(3):
        k = int(npts[n])
        colours = len(np.unique(tgt[n].reshape(-1, 4)[:, :3], axis=0))
        assert colours <= k <= colours * tiles                                  # distinct per tile of 1024 pixels, tiles not merged
        assert pts[n, :k, 3].sum() == S * S and (pts[n, :k, 3] >= 1).all()      # the counts cover every pixel once
        assert len(np.unique(pts[n, :k, :3], axis=0)) == colours
    # an image with more colours than the list holds is contracted densely (npoints = -1), with the same result
    listed_n, npts_n, _ = _fwd3(noisy, points=True, cap=64)
    if S >= 64:
        assert (npts_n == -1).all()
    assert np.abs(listed_n - dense_n).max() <= 1e-4 * dense_n.max()
    # determinism: the list is a function of the image alone
    again = _fwd3(tgt, points=True)
    assert np.array_equal(again[1], npts) and np.array_equal(again[0], listed_t)
    for n in range(3):
        assert np.array_equal(again[2][n, :npts[n]], pts[n, :npts[n]])


@pytest.mark.parametrize("size", [8, 64])
def test_hellinger_loss_and_gradient_match_closed_form(size):
    rng = np.random.default_rng(33)
    B = 2
    src, tgt = rg.synthetic_rgba_batch(rng, B, 64, palette_size=12)
    src, tgt = src[:, :size, :size], tgt[:, :size, :size]
    fake = np.clip(src + rng.normal(scale=0.05, size=src.shape), -1, 1).astype(np.float32)
    ft = torch.tensor(fake, dtype=F64, requires_grad=True)
    real_h = rg.rgbuv_histogram(torch.tensor(tgt, dtype=F64))
    loss = rg.hellinger_loss(real_h, rg.rgbuv_histogram(ft))
    loss.backward()
    pad = lambda a: U.halo_from(np.concatenate([a, np.zeros_like(a)], -1), L.F32)
    rb, fb = pad(tgt), pad(fake)
    n = B * 3 * 64 * 64
    h_r, h_f, gh = (torch.empty(n, dtype=torch.float32, device=U.DEV) for _ in range(3))
    tot = torch.empty((2, B), dtype=torch.float32, device=U.DEV)
    sq = torch.zeros(4, dtype=torch.float32, device=U.DEV)
    out_loss = torch.zeros(1, dtype=torch.float32, device=U.DEV)
    dimg = torch.empty(3 * B * size * size * 4, dtype=torch.float32, device=U.DEV)
    L.call("p2p_rgbuv_hist_fwd", L.F32, B, size, size, C.byref(rb.view()), U.ptr(h_r), U.stream())
    L.call("p2p_rgbuv_hist_fwd", L.F32, B, size, size, C.byref(fb.view()), U.ptr(h_f), U.stream())
    sqp = torch.zeros(B, dtype=torch.float32, device=U.DEV)
    L.call("p2p_hellinger_fwd", U.ptr(h_r), U.ptr(h_f), B, U.ptr(tot[0]), U.ptr(tot[1]), U.ptr(sqp), U.ptr(sq), U.stream())
    assert float(sq[0]) == float(sqp.cpu().double().sum().float()) or abs(float(sq[0]) - float(sqp.sum())) < 1e-6 * float(sq[0])
    L.call("p2p_hellinger_finish", U.ptr(sq), 1.0 / B, U.ptr(out_loss), U.stream())
    assert abs(float(out_loss[0]) - float(loss)) < 1e-4 * float(loss)
    L.call("p2p_rgbuv_hist_hellinger_bwd", L.F32, B, size, size, C.byref(fb.view()), U.ptr(h_r), U.ptr(h_f), U.ptr(tot[0]),
           U.ptr(tot[1]), U.ptr(sq), 1.0 / (2.0 * math.sqrt(2.0) * B), U.ptr(gh), U.ptr(dimg), U.stream())
    got = dimg.view(3, B, size, size, 4).sum(0).cpu().numpy()
    ref = ft.grad.numpy()
    assert np.count_nonzero(got[..., 3]) == 0
    # the gradient spans ~6 decades (1/(x+1e-6) on near-black pixels): compare in the max-norm and in L2
    assert U.rel_err(got, ref) < 2e-3
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 2e-3
    # the shared-row form (three components in one workgroup, ONE slab): same gradient
    one = torch.full((B * size * size * 4,), float("nan"), dtype=torch.float32, device=U.DEV)
    L.call("p2p_rgbuv_hist_hellinger_bwd3", L.F32, B, size, size, C.byref(fb.view()), U.ptr(h_r), U.ptr(h_f), U.ptr(tot[0]),
           U.ptr(tot[1]), U.ptr(sq), 1.0 / (2.0 * math.sqrt(2.0) * B), U.ptr(gh), U.ptr(one), U.stream())
    got3 = one.view(B, size, size, 4).cpu().numpy()
    assert np.count_nonzero(got3[..., 3]) == 0 and np.isfinite(got3).all()
    assert U.rel_err(got3, ref) < 2e-3
    assert np.linalg.norm(got3 - ref) / np.linalg.norm(ref) < 2e-3
    assert np.linalg.norm(got3 - got) / np.linalg.norm(got) < 1e-4          # against the per-component form: f32 summation order only


def test_histogram_gradient_kernel_is_reproducible_launch_to_launch():
    """p2p_rgbuv_hist_hellinger_bwd3 stores each batch's result while the next batch's MFMAs are in flight.  Round 4 measured that an
    MFMA's write-back is not interlocked against a pending global store's read of its data registers (a first version stored a wrong
    dword in about one workgroup per thousand: tools/exp/hist_repro.py): the kernel now orders the two itself.  200 launches on one
    input, four workgroups per image (16 batches each), must agree bit for bit."""
    rng = np.random.default_rng(35)
    B, size = 64, 64
    _, tgt = rg.synthetic_rgba_batch(rng, 8, 64, palette_size=24)
    tgt = np.tile(tgt, (B // 8, 1, 1, 1))
    fake = np.clip(tgt + rng.normal(scale=0.05, size=tgt.shape), -1, 1).astype(np.float32)
    t_d, f_d = U.dev(tgt), U.dev(fake)
    vt, vf = L.Tensor(t_d.data_ptr(), size * size, size, 4), L.Tensor(f_d.data_ptr(), size * size, size, 4)
    n = B * 3 * 64 * 64
    h_r, h_f, gh = (torch.empty(n, dtype=torch.float32, device=U.DEV) for _ in range(3))
    ws = torch.empty(L.lib().p2p_rgbuv_hist_fwd3_workspace_bytes(B) // 4, dtype=torch.float32, device=U.DEV)
    tot = torch.empty((2, B), dtype=torch.float32, device=U.DEV)
    sq, sqp = torch.zeros(4, dtype=torch.float32, device=U.DEV), torch.zeros(B, dtype=torch.float32, device=U.DEV)
    for view, out in ((vt, h_r), (vf, h_f)):
        L.call("p2p_rgbuv_hist_fwd3", L.F32, B, size, size, C.byref(view), None, None, 1024, U.ptr(out), U.ptr(ws), U.stream())
    L.call("p2p_hellinger_fwd", U.ptr(h_r), U.ptr(h_f), B, U.ptr(tot[0]), U.ptr(tot[1]), U.ptr(sqp), U.ptr(sq), U.stream())
    first = None
    for i in range(200):
        dimg = torch.full((B * size * size * 4,), float("nan"), dtype=torch.float32, device=U.DEV)
        L.call("p2p_rgbuv_hist_hellinger_bwd3", L.F32, B, size, size, C.byref(vf), U.ptr(h_r), U.ptr(h_f), U.ptr(tot[0]), U.ptr(tot[1]),
               U.ptr(sq), 1.0 / (2.0 * math.sqrt(2.0) * B), U.ptr(gh), U.ptr(dimg), U.stream())
        if first is None:
            first = dimg
            assert bool(torch.isfinite(first).all())
        else:
            assert bool((dimg == first).all()), f"launch {i} differs from launch 0"


def test_histogram_model_train_step_matches_oracle():
    B, S = 2, 64
    rng, Gp, Dp, src, tgt, masks = setup_case(B, S, 34)
    tm = [torch.tensor(m, dtype=F64) for m in masks]
    ref = rg.train_step_rgba(Gp, Dp, torch.tensor(src, dtype=F64), torch.tensor(tgt, dtype=F64), tm, lambda_l1=30.0,
                             lambda_hist=1.0)
    eng = E.Pix2PixEngine(4, 4, "tanh", S, L.F32)
    eng.set_params(to_np(Gp), to_np(Dp))
    out = eng.train_step_rgba(src, tgt, 30.0, lambda_hist=1.0, masks=masks, apply_update=False).cpu().numpy()
    g, d = ref["g_loss"], ref["d_loss"]
    want = np.array([g[0], g[1], g[2], g[3], d[0], d[1], d[2]])
    print("losses", out, want)
    for i in range(7):
        assert abs(out[i] - want[i]) <= 1e-4 * abs(want[i]), (i, out[i], want[i])
    wg = grad_report(eng.G.export(eng.G.grads), ref["g_grads"])
    print("worst G grad", wg)
    assert wg[1][1] < 2e-3          # L2; the histogram gradient's dynamic range makes ReLU-flip outliers likelier


def test_histogram_model_bf16_step_matches_oracle():
    """bf16 throughput mode of the histogram model (c3 / c5 as benchmarked): the histogram loss is evaluated in f32 on f32
    images in every mode (real image from the f32 batch, fake image from the unrounded tanh copy), so against the oracle
    with the same bf16 storage points the histogram loss agrees to 1e-3 -- it sees nothing of bf16 but the rounding of
    the head's pre-activation -- and the other losses to 2e-3 like the baseline model's."""
    B, S = 2, 64
    rng, Gp, Dp, src, tgt, masks = setup_case(B, S, 34)
    tm = [torch.tensor(m, dtype=F64) for m in masks]
    with rg.storage_dtype(torch.bfloat16):
        ref = rg.train_step_rgba(Gp, Dp, torch.tensor(src, dtype=F64), torch.tensor(tgt, dtype=F64), tm, lambda_l1=30.0,
                                 lambda_hist=1.0)
    eng = E.Pix2PixEngine(4, 4, "tanh", S, L.BF16)
    eng.set_params(to_np(Gp), to_np(Dp))
    out = eng.train_step_rgba(src, tgt, 30.0, lambda_hist=1.0, masks=masks, apply_update=False).cpu().numpy()
    g, d = ref["g_loss"], ref["d_loss"]
    want = np.array([g[0], g[1], g[2], g[3], d[0], d[1], d[2]])
    print("losses", out, want)
    assert abs(out[3] - want[3]) <= 1e-3 * abs(want[3]), (out[3], want[3])
    for i in (0, 1, 2, 4, 5, 6):
        assert abs(out[i] - want[i]) <= 2e-3 * abs(want[i]), (i, out[i], want[i])
    wg = grad_report(eng.G.export(eng.G.grads), ref["g_grads"])
    print("worst G grad", wg)
    assert wg[1][1] < 0.3          # bf16 storage noise of the deep layers, as in the baseline model


def test_indexed_model_bf16_step_matches_oracle():
    """bf16 mode of the indexed model (c4 as benchmarked) against the oracle with the same bf16 storage points."""
    B, S = 2, 64
    rng = np.random.default_rng(37)
    Gp = rg.perturb_affine(rg.init_params(rg.generator_param_shapes(1, 256), rng, F64), rng)
    Dp = rg.perturb_affine(rg.init_params(rg.discriminator_param_shapes(1), rng, F64), rng)
    Gp["down1.kernel"] *= 0.05
    Dp["down.kernel"] *= 0.05
    src, tgt, _pal = rg.synthetic_indexed_batch(rng, B, S)
    masks = [rng.integers(0, 2, size=s).astype(np.uint8) for s in rg.dropout_mask_shapes(B, S)]
    with rg.storage_dtype(torch.bfloat16):
        ref = rg.train_step_indexed(Gp, Dp, torch.tensor(src), torch.tensor(tgt), [torch.tensor(m, dtype=F64) for m in masks], 0.01)
    eng = E.Pix2PixEngine(1, 256, "softmax", S, L.BF16)
    eng.set_params(to_np(Gp), to_np(Dp))
    out = eng.train_step_indexed(src, tgt, 0.01, masks=masks, apply_update=False).cpu().numpy()
    g, d = ref["g_loss"], ref["d_loss"]
    want = np.array([g[0], g[1], g[2], g[3], d[0], d[1], d[2]])
    print("losses", out, want)
    # the adversarial / discriminator terms see the argmax image: a near-tie that resolves differently moves them a little
    for i in (2, 3):
        assert abs(out[i] - want[i]) <= 2e-3 * abs(want[i]), (i, out[i], want[i])
    for i in (0, 1, 4, 5, 6):
        assert abs(out[i] - want[i]) <= 2e-2 * abs(want[i]), (i, out[i], want[i])
    wg = grad_report(eng.G.export(eng.G.grads), ref["g_grads"])
    print("worst G grad", wg)
    assert wg[1][1] < 0.3


def test_argmax_is_bit_exact_with_engineered_ties():
    rng = np.random.default_rng(35)
    M, Cn = 4096, 256
    p = rng.random((M, Cn)).astype(np.float32)
    p /= p.sum(1, keepdims=True)
    for r in range(0, M, 7):                     # exact ties, the winner must be the lowest index
        i, j = sorted(rng.choice(Cn, 2, replace=False))
        p[r, i] = p[r, j] = p[r].max() * 1.5
    p[5, :] = 1.0 / Cn                            # fully uniform row -> index 0
    got = torch.empty(M, dtype=torch.int32, device=U.DEV)
    p_d = U.dev(p)
    L.call("p2p_argmax_lastdim", U.ptr(p_d), M, Cn, U.ptr(got), U.stream())
    want = torch.argmax(torch.tensor(p), dim=-1).to(torch.int32).numpy()      # tf.argmax semantics: first maximum
    want_np = np.argmax(p, axis=-1).astype(np.int32)
    assert np.array_equal(want, want_np)
    assert np.array_equal(got.cpu().numpy(), want_np)
    assert got[5].item() == 0


@pytest.mark.parametrize("n", [1, 3])
def test_fused_indexed_head_matches_conv_softmax_cce_argmax(n):
    """p2p_head_softmax_cce (bf16, 64x64): Conv2D(256, 4, stride 1, SAME (1, 2), bias) + softmax + CCE + argmax + gradient + bias
    gradient in one launch, against the f64 evaluation of the same bf16-rounded inputs (networks.py:75-78,
    pix2pix_model.py:268,273-293,300-301)."""
    import torch.nn.functional as F
    dtype, S, cin, cpad, Cn = L.BF16, 64, 33, 40, 256
    rng = np.random.default_rng(38 + n)
    x = U.q(rng.normal(size=(n, S, S, cin)), dtype)
    w = U.q(0.08 * rng.normal(size=(4, 4, cin, Cn)), dtype)            # HWIO
    bias = (0.1 * rng.normal(size=Cn)).astype(np.float32)
    tgt = rng.integers(0, Cn, size=(n, S, S, 1)).astype(np.int32)
    assert L.lib().p2p_head_softmax_ok(dtype, n, S, S, cpad, Cn)
    xb = E.HaloBuf(n, S, S, cpad, dtype, U.DEV)
    xb.t[:, 2:-2, 2:-2, :cin] = U.dev(x, U.tdt(dtype))
    wt = np.zeros((16, Cn, cpad), np.float32)                            # op-G copy wt[tap][d][g]
    wt[:, :, :cin] = w.reshape(16, cin, Cn).transpose(0, 2, 1)
    wt_d = U.dev(wt.reshape(-1), U.tdt(dtype))
    tb = U.halo_from(np.concatenate([tgt.astype(np.float32), np.zeros((n, S, S, 7), np.float32)], -1), dtype)
    fb = E.HaloBuf(n, S, S, 8, dtype, U.DEV)
    dz = E.HaloBuf(n, S, S, Cn, dtype, U.DEV)
    ws = torch.full((L.lib().p2p_head_softmax_workspace_bytes(n, S) // 4 + 4,), float("nan"), dtype=torch.float32, device=U.DEV)
    loss = torch.zeros(2, dtype=torch.float32, device=U.DEV)
    dbias = torch.full((Cn,), float("nan"), dtype=torch.float32, device=U.DEV)
    inv, lam = 1.0 / (n * S * S), 0.5
    L.call("p2p_head_softmax_cce", dtype, n, S, S, cpad, Cn, C.byref(xb.view()), U.ptr(wt_d), U.ptr(U.dev(bias)), C.byref(tb.view()),
           C.byref(fb.view()), lam * inv, inv, C.byref(dz.view()), U.ptr(dbias), U.ptr(ws), U.ptr(loss), U.stream())
    torch.cuda.synchronize()
    xt = torch.tensor(x, dtype=F64).permute(0, 3, 1, 2)
    z = F.conv2d(F.pad(xt, (1, 2, 1, 2)), torch.tensor(w, dtype=F64).permute(3, 2, 0, 1), torch.tensor(bias, dtype=F64))
    z = z.permute(0, 2, 3, 1).detach().requires_grad_(True)
    seg = rg.categorical_crossentropy_from_logits(z, torch.tensor(tgt))
    (lam * seg).backward()
    p_ref = torch.softmax(z, -1).detach().numpy()
    assert abs(float(loss[0]) - float(seg)) < 2e-5 * float(seg), (float(loss[0]), float(seg))
    onehot = np.eye(Cn)[tgt[..., 0]]
    assert abs(float(loss[1]) - np.abs(onehot - p_ref).mean()) < 1e-5
    got_dz = U.halo_to_np(dz).astype(np.float64)
    assert U.rel_err(got_dz, z.grad.numpy()) < 6e-3                      # the gradient is stored in bf16
    # bias gradient = column sums of the STORED (rounded) gradient
    np.testing.assert_allclose(dbias.cpu().numpy(), got_dz.sum(axis=(0, 1, 2)), rtol=2e-5, atol=1e-7)
    # argmax: identical to the f64 argmax wherever the top two probabilities are not within f32 rounding of each other
    idx = U.halo_to_np(fb)[..., 0].astype(np.int64)
    ref_idx = p_ref.argmax(-1)
    top2 = np.sort(p_ref, -1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 1e-6 * top2[..., 1]
    assert (idx == ref_idx)[clear].all() and clear.mean() > 0.99
    assert not U.halo_to_np(fb)[..., 1:].any()                            # only channel 0 of the index pixel is written


def test_indexed_head_data_gradient_kernel():
    """p2p_head_dgrad (bf16, 64x64): op P, stride 1, 256 -> first 32 of 33 input channels, against the f64 sum over taps"""
    dtype, n, S, cg, cd = L.BF16, 2, 64, 33, 256
    rng = np.random.default_rng(40)
    dzv = U.q(rng.normal(size=(n, S, S, cd)), dtype)
    w = U.q(0.05 * rng.normal(size=(16, cg, cd)), dtype)                  # [tap][g][d]
    dzb = U.halo_from(dzv, dtype)
    wn = torch.zeros(16 * E.up32(cg) * cd, dtype=U.tdt(dtype), device=U.DEV)
    wt = torch.zeros(16 * E.up32(cd) * E.pad8(cg), dtype=U.tdt(dtype), device=U.DEV)
    L.call("p2p_weight_prep_pad", dtype, U.ptr(U.dev(w.reshape(-1))), cg, cd, U.ptr(wn), E.up32(cg), cd, U.ptr(wt), E.up32(cd),
           E.pad8(cg), U.stream())
    out = E.DenseBuf(n, S, S, 40, U.tdt(dtype), U.DEV)
    out.t.fill_(7.0)
    assert L.lib().p2p_head_dgrad_ok(dtype, n, S, S, cd, 32, E.up32(cg), cd, 40)
    L.call("p2p_head_dgrad", dtype, n, S, S, cd, 32, C.byref(dzb.view()), U.ptr(wn), E.up32(cg), C.byref(out.view()), U.stream())
    torch.cuda.synchronize()
    dzp = np.pad(dzv.astype(np.float64), ((0, 0), (2, 2), (2, 2), (0, 0)))
    ref = np.zeros((n, S, S, 32))
    for kh in range(4):
        for kw in range(4):
            ref += dzp[:, 3 - kh:3 - kh + S, 3 - kw:3 - kw + S, :] @ w[kh * 4 + kw, :32, :].astype(np.float64).T
    got = U.dense_to_np(out)
    assert U.rel_err(got[..., :32], ref) < 6e-3                            # bf16 output
    assert (got[..., 32:] == 7.0).all()                                    # the source / padding channels are not touched


@pytest.mark.parametrize("dtype", [L.F32, L.BF16])
def test_softmax_cce_argmax_kernel(dtype):
    rng = np.random.default_rng(36)
    n, s, Cn = 2, 8, 256
    z = U.q(rng.normal(size=(n, s, s, Cn)) * 3, dtype)
    tgt = rng.integers(0, Cn, size=(n, s, s, 1)).astype(np.int32)
    zb = E.DenseBuf(n, s, s, Cn, U.tdt(dtype), U.DEV)
    zb.t.copy_(U.dev(z.reshape(-1, Cn), U.tdt(dtype)))
    tb = U.halo_from(tgt.astype(np.float32), dtype)
    fb = E.HaloBuf(n, s, s, 1, dtype, U.DEV)
    dz = E.HaloBuf(n, s, s, Cn, dtype, U.DEV)
    probs = torch.empty((n, s, s, Cn), dtype=torch.float32, device=U.DEV)
    loss = torch.zeros(2, dtype=torch.float32, device=U.DEV)
    part = torch.zeros(2 * 8192, dtype=torch.float32, device=U.DEV)
    inv = 1.0 / (n * s * s)
    L.call("p2p_softmax_cce_argmax", dtype, n, s, s, Cn, C.byref(zb.view()), C.byref(tb.view()), C.byref(fb.view()),
           0.5 * inv, inv, C.byref(dz.view()), U.ptr(probs), U.ptr(part), U.ptr(loss), U.stream())
    zt = torch.tensor(z, dtype=F64, requires_grad=True)
    seg = rg.categorical_crossentropy_from_logits(zt, torch.tensor(tgt))
    p_ref = torch.softmax(zt, -1)
    (0.5 * seg).backward()
    assert abs(float(loss[0]) - float(seg)) < 1e-5 * float(seg)
    onehot = np.eye(Cn)[tgt[..., 0]]
    assert abs(float(loss[1]) - np.abs(onehot - p_ref.detach().numpy()).mean()) < 1e-5
    assert U.rel_err(probs.cpu().numpy(), p_ref.detach().numpy()) < 1e-5
    assert np.array_equal(U.halo_to_np(fb)[..., 0].astype(np.int64), np.argmax(probs.cpu().numpy(), -1))
    assert U.rel_err(U.halo_to_np(dz), zt.grad.numpy()) < (1e-5 if dtype == L.F32 else 6e-3)


def test_softmax_cce_stays_finite_when_the_target_probability_underflows():
    """One very negative target logit: p_t underflows in f32, -log(p_t) would be inf.  The kernel evaluates the
    log-sum-exp form of the logits path (pix2pix_model.py:265,274 through Keras' cached logits), which is
    finite and equals the float64 log-softmax; the gradient stays p - onehot."""
    n, s, Cn = 1, 8, 256
    rng = np.random.default_rng(38)
    z = rng.normal(size=(n, s, s, Cn)).astype(np.float32)
    tgt = rng.integers(0, Cn, size=(n, s, s, 1)).astype(np.int32)
    z[0, 0, 0, tgt[0, 0, 0, 0]] = -200.0          # exp(-200 - max) == 0 in f32
    zb = E.DenseBuf(n, s, s, Cn, torch.float32, U.DEV)
    zb.t.copy_(U.dev(z.reshape(-1, Cn)))
    tb = U.halo_from(tgt.astype(np.float32), L.F32)
    fb = E.HaloBuf(n, s, s, 1, L.F32, U.DEV)
    dz = E.HaloBuf(n, s, s, Cn, L.F32, U.DEV)
    loss = torch.zeros(2, dtype=torch.float32, device=U.DEV)
    part = torch.zeros(2 * 8192, dtype=torch.float32, device=U.DEV)
    inv = 1.0 / (n * s * s)
    runs = []
    for _ in range(2):
        L.call("p2p_softmax_cce_argmax", L.F32, n, s, s, Cn, C.byref(zb.view()), C.byref(tb.view()), C.byref(fb.view()),
               inv, inv, C.byref(dz.view()), None, U.ptr(part), U.ptr(loss), U.stream())
        runs.append(loss.cpu().numpy().copy())
    assert np.array_equal(runs[0], runs[1])        # fixed-order partial sums: bit-reproducible
    zt = torch.tensor(z, dtype=F64, requires_grad=True)
    seg = rg.categorical_crossentropy_from_logits(zt, torch.tensor(tgt))
    seg.backward()
    assert np.isfinite(runs[0]).all() and float(seg) > 3.0
    assert abs(runs[0][0] - float(seg)) < 1e-5 * float(seg)
    assert U.rel_err(U.halo_to_np(dz), zt.grad.numpy()) < 1e-5


def test_indexed_model_train_step_matches_oracle():
    B, S = 2, 64
    rng = np.random.default_rng(37)
    Gp = rg.perturb_affine(rg.init_params(rg.generator_param_shapes(1, 256), rng, F64), rng)
    Dp = rg.perturb_affine(rg.init_params(rg.discriminator_param_shapes(1), rng, F64), rng)
    # index inputs are un-normalised (0..255): scale the first-layer kernels so the test is not saturated
    Gp["down1.kernel"] *= 0.05
    Dp["down.kernel"] *= 0.05
    src, tgt, _pal = rg.synthetic_indexed_batch(rng, B, S)
    masks = [rng.integers(0, 2, size=s).astype(np.uint8) for s in rg.dropout_mask_shapes(B, S)]
    ref = rg.train_step_indexed(Gp, Dp, torch.tensor(src), torch.tensor(tgt), [torch.tensor(m, dtype=F64) for m in masks], 0.01)
    eng = E.Pix2PixEngine(1, 256, "softmax", S, L.F32)
    eng.set_params(to_np(Gp), to_np(Dp))
    out = eng.train_step_indexed(src, tgt, 0.01, masks=masks, apply_update=False).cpu().numpy()
    g, d = ref["g_loss"], ref["d_loss"]
    want = np.array([g[0], g[1], g[2], g[3], d[0], d[1], d[2]])
    print("losses", out, want)
    for i in range(7):
        assert abs(out[i] - want[i]) <= 1e-4 * abs(want[i]), (i, out[i], want[i])
    wg = grad_report(eng.G.export(eng.G.grads), ref["g_grads"])
    wd = grad_report(eng.D.export(eng.D.grads), ref["d_grads"])
    print("worst G", wg, "worst D", wd)
    assert wg[1][1] < 2e-3 and wd[1][1] < 1e-4
    # the generated index image is the argmax of the oracle's probabilities wherever that argmax is not a near-tie
    idx = eng.generate_indexed(src, masks=masks).cpu().numpy()
    pr = ref["probs"].numpy()
    top2 = np.sort(pr, -1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 1e-5
    assert np.array_equal(idx[..., 0][clear], ref["fake_idx"].numpy()[..., 0][clear])
    assert clear.mean() > 0.99


@pytest.mark.timeout(600)
def test_fused_indexed_head_at_the_c4_launch_shape_against_the_generic_path():
    """p2p_head_softmax_cce at c4's per-GPU launch shape (n = 128, 64x64, 33(+7) -> 256) against the generic path
    on the SAME inputs -- p2p_igemm_edge (op G, stride 1, bias) writes the bf16 logits, p2p_softmax_cce_argmax takes softmax, CCE,
    argmax and gradient from them.  The fused kernel never rounds the logits to bf16, so: loss to 2e-4, gradient to two bf16 steps
    of its scale, identical argmax wherever the generic path's top two logits differ by more than the bf16 rounding of a logit
    (pix2pix_model.py:268,273-293,300-301)."""
    dtype, n, S, cin, cpad, Cn = L.BF16, 128, 64, 33, 40, 256
    rng = np.random.default_rng(77)
    x = U.q(rng.normal(size=(n, S, S, cin)).astype(np.float32), dtype)
    w = U.q(0.08 * rng.normal(size=(4, 4, cin, Cn)), dtype)
    bias = (0.1 * rng.normal(size=Cn)).astype(np.float32)
    tgt = rng.integers(0, Cn, size=(n, S, S, 1)).astype(np.int32)
    assert L.lib().p2p_head_softmax_ok(dtype, n, S, S, cpad, Cn)
    xb = E.HaloBuf(n, S, S, cpad, dtype, U.DEV)
    xb.t[:, 2:-2, 2:-2, :cin] = U.dev(x, U.tdt(dtype))
    wt = np.zeros((16, Cn, cpad), np.float32)
    wt[:, :, :cin] = w.reshape(16, cin, Cn).transpose(0, 2, 1)
    wt_d, bias_d = U.dev(wt.reshape(-1), U.tdt(dtype)), U.dev(bias)
    tb = E.HaloBuf(n, S, S, 8, dtype, U.DEV)
    tb.t[:, 2:-2, 2:-2, :1] = U.dev(tgt.astype(np.float32), U.tdt(dtype))
    inv, lam = 1.0 / (n * S * S), 0.01
    # fused
    fb, dz = E.HaloBuf(n, S, S, 8, dtype, U.DEV), E.HaloBuf(n, S, S, Cn, dtype, U.DEV)
    ws = torch.zeros(L.lib().p2p_head_softmax_workspace_bytes(n, S) // 4 + 4, dtype=torch.float32, device=U.DEV)
    loss, dbias = torch.zeros(2, dtype=torch.float32, device=U.DEV), torch.zeros(Cn, dtype=torch.float32, device=U.DEV)
    L.call("p2p_head_softmax_cce", dtype, n, S, S, cpad, Cn, C.byref(xb.view()), U.ptr(wt_d), U.ptr(bias_d), C.byref(tb.view()),
           C.byref(fb.view()), lam * inv, inv, C.byref(dz.view()), U.ptr(dbias), U.ptr(ws), U.ptr(loss), U.stream())
    # generic: logits in bf16, then softmax / CCE / argmax / gradient
    zb = E.DenseBuf(n, S, S, Cn, U.tdt(dtype), U.DEV)
    L.call("p2p_igemm_edge", L.OP_G, 1, dtype, n, S, S, cpad, Cn, Cn, C.byref(xb.view()), C.byref(zb.view()), U.ptr(wt_d), U.ptr(bias_d),
           L.ACT_NONE, 0.3, U.stream())
    fb2, dz2 = E.HaloBuf(n, S, S, 8, dtype, U.DEV), E.HaloBuf(n, S, S, Cn, dtype, U.DEV)
    part = torch.zeros(2 * 8192, dtype=torch.float32, device=U.DEV)
    loss2 = torch.zeros(2, dtype=torch.float32, device=U.DEV)
    L.call("p2p_softmax_cce_argmax", dtype, n, S, S, Cn, C.byref(zb.view()), C.byref(tb.view()), C.byref(fb2.view()), lam * inv, inv,
           C.byref(dz2.view()), None, U.ptr(part), U.ptr(loss2), U.stream())
    torch.cuda.synchronize()
    l1, l2 = loss.cpu().numpy(), loss2.cpu().numpy()
    assert abs(l1[0] - l2[0]) < 2e-4 * abs(l2[0]), (l1, l2)
    assert abs(l1[1] - l2[1]) < 2e-4 * abs(l2[1]) + 1e-7, (l1, l2)
    g1 = dz.t[:, 2:-2, 2:-2, :].float()
    g2 = dz2.t[:, 2:-2, 2:-2, :].float()
    scale = float(g2.abs().max())
    assert float((g1 - g2).abs().max()) < 2.0 ** -6 * scale          # the bf16 logits of the generic path move a probability by ~2^-8 of itself
    assert float(torch.linalg.vector_norm(g1 - g2) / torch.linalg.vector_norm(g2)) < 2e-2
    # bias gradient of the fused kernel = column sums of ITS stored gradient
    np.testing.assert_allclose(dbias.cpu().numpy(), g1.double().sum(dim=(0, 1, 2)).cpu().numpy(), rtol=2e-4, atol=1e-6)
    i1 = fb.t[:, 2:-2, 2:-2, 0].float()
    i2 = fb2.t[:, 2:-2, 2:-2, 0].float()
    z = zb.t.view(n, S, S, Cn).float()
    top2 = torch.topk(z, 2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 2.0 ** -6 * top2[..., 0].abs().clamp_min(1.0)       # beyond the rounding of a bf16 logit
    assert bool((i1 == i2)[clear].all()) and float(clear.float().mean()) > 0.8
    assert float((i1 == i2).float().mean()) > 0.98
