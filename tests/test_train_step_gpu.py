"""-m gpu: the whole Pix2Pix train step (pix2pix_model.py:62-89) on the GPU against the CPU oracle:
losses, every gradient tensor, and the post-Adam weights over two consecutive steps."""
import numpy as np
import pytest
import torch

from oracle import reference_graph as rg
from palette_and_histo_gan_amd import _lib as L
from palette_and_histo_gan_amd import engine as E

pytestmark = pytest.mark.gpu
F64 = torch.float64


def setup_case(B, S, seed, in_ch=4, out_ch=4):
    rng = np.random.default_rng(seed)
    Gp = rg.perturb_affine(rg.init_params(rg.generator_param_shapes(in_ch, out_ch), rng, F64), rng)
    Dp = rg.perturb_affine(rg.init_params(rg.discriminator_param_shapes(in_ch), rng, F64), rng)
    src, tgt = rg.synthetic_rgba_batch(rng, B, S)
    masks = [rng.integers(0, 2, size=s).astype(np.uint8) for s in rg.dropout_mask_shapes(B, S)]
    return rng, Gp, Dp, src, tgt, masks


def to_np(p):
    return {k: v.numpy() for k, v in p.items()}


def grad_report(eng_grads, ref_grads):
    worst = ("", 0.0)
    for k, ref in ref_grads.items():
        ref = ref.numpy()
        got = eng_grads[k]
        scale = np.abs(ref).max()
        if scale == 0.0:
            err = float(np.abs(got).max())
        else:
            err = float(np.abs(got - ref).max() / scale)
        if err > worst[1]:
            worst = (k, err)
    return worst


@pytest.mark.parametrize("use_mfma", [False, True])
@pytest.mark.parametrize("dtype,loss_tol,grad_tol", [(L.F32, 1e-4, 2e-3), (L.BF16, 3e-2, 0.25)])
def test_train_step_matches_oracle(dtype, loss_tol, grad_tol, use_mfma):
    B, S = 2, 64
    rng, Gp, Dp, src, tgt, masks = setup_case(B, S, 21)
    ref = rg.train_step_rgba(Gp, Dp, torch.tensor(src, dtype=F64), torch.tensor(tgt, dtype=F64),
                             [torch.tensor(m, dtype=F64) for m in masks], lambda_l1=100.0)
    eng = E.Pix2PixEngine(4, 4, "tanh", S, dtype, use_mfma=use_mfma)
    eng.set_params(to_np(Gp), to_np(Dp))
    out = eng.train_step_rgba(src, tgt, 100.0, masks=masks, apply_update=False).cpu().numpy()
    g_ref, d_ref = ref["g_loss"], ref["d_loss"]
    want = np.array([g_ref[0], g_ref[1], g_ref[2], 0.0, d_ref[0], d_ref[1], d_ref[2]])
    print("losses got", out, "want", want)
    for i in (0, 1, 2, 4, 5, 6):
        assert abs(out[i] - want[i]) <= loss_tol * abs(want[i]), (i, out[i], want[i])
    wg = grad_report(eng.G.export(eng.G.grads), ref["g_grads"])
    wd = grad_report(eng.D.export(eng.D.grads), ref["d_grads"])
    print("worst G grad", wg, "worst D grad", wd)
    assert wg[1] < grad_tol and wd[1] < grad_tol


def test_two_adam_steps_match_oracle_f32():
    B, S = 2, 64
    rng, Gp, Dp, src, tgt, masks = setup_case(B, S, 22)
    eng = E.Pix2PixEngine(4, 4, "tanh", S, L.F32)
    eng.set_params(to_np(Gp), to_np(Dp))
    gm, gv, dm, dv = (rg.zeros_like_params(Gp), rg.zeros_like_params(Gp), rg.zeros_like_params(Dp), rg.zeros_like_params(Dp))
    tm = [torch.tensor(m, dtype=F64) for m in masks]
    for t in (1, 2):
        ref = rg.train_step_rgba(Gp, Dp, torch.tensor(src, dtype=F64), torch.tensor(tgt, dtype=F64), tm, lambda_l1=100.0)
        Gp, gm, gv = rg.keras_adam(Gp, ref["g_grads"], gm, gv, t)
        Dp, dm, dv = rg.keras_adam(Dp, ref["d_grads"], dm, dv, t)
        eng.train_step_rgba(src, tgt, 100.0, masks=masks)
        got_g, got_d = eng.G.export(), eng.D.export()
        # one Adam step moves a weight by at most ~lr = 2e-4; demand agreement to 2% of that
        for k in Gp:
            assert np.abs(got_g[k] - Gp[k].numpy()).max() < 4e-6 * t, (t, k)
        for k in Dp:
            assert np.abs(got_d[k] - Dp[k].numpy()).max() < 4e-6 * t, (t, k)


def test_generate_is_forward_of_train_step():
    B, S = 2, 64
    rng, Gp, Dp, src, tgt, masks = setup_case(B, S, 23)
    eng = E.Pix2PixEngine(4, 4, "tanh", S, L.F32)
    eng.set_params(to_np(Gp), to_np(Dp))
    fake = eng.generate(src, masks=masks).cpu().numpy()
    ref = rg.unet_generator(Gp, torch.tensor(src, dtype=F64), [torch.tensor(m, dtype=F64) for m in masks], "tanh").numpy()
    assert np.abs(fake - ref).max() < 1e-4
