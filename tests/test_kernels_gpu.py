"""-m gpu: every HIP kernel against the CPU oracle on seeded inputs, through the C ABI."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import reference_graph as rg
from oracle import np_restatement as npr
from palette_and_histo_gan_amd import _lib as L
from palette_and_histo_gan_amd import engine as E
from tests import gpu_util as U

pytestmark = pytest.mark.gpu
F64 = torch.float64
DTYPES = [L.F32, L.BF16]
OUT_TOL = {L.F32: 2e-5, L.BF16: 6e-3}      # max-norm relative; bf16 = output rounding (2^-8) of f32-accumulated sums


def oracle_ops(hi, lo, w, stride):
    """(G, P, W) results of the (hi, lo, W[4,4,Cg,Cd]) layer description in float64."""
    hi_t = torch.tensor(hi, dtype=F64, requires_grad=True)
    w_t = torch.tensor(w, dtype=F64, requires_grad=True)
    lo_t = torch.tensor(lo, dtype=F64)
    if stride == 2:
        g = rg.conv4x4_s2(hi_t, w_t)
    else:
        g = rg.conv4x4_s1_bias(hi_t, w_t, None)
    (g * lo_t).sum().backward()
    return g.detach().numpy(), hi_t.grad.numpy(), w_t.grad.numpy()


def make_case(rng, n, lh, cg, cd, stride, dtype):
    hi = U.q(rng.normal(size=(n, stride * lh, stride * lh, cg)), dtype)
    lo = U.q(rng.normal(size=(n, lh, lh, cd)), dtype)
    w = U.q(rng.normal(scale=0.05, size=(4, 4, cg, cd)), dtype)
    return hi, lo, w


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,lh,cg,cd,stride", [(2, 4, 4, 64, 2), (1, 8, 8, 3, 2), (2, 5, 36, 4, 1), (3, 4, 64, 1, 1),
                                                (2, 1, 5, 7, 2), (1, 3, 2, 2, 2), (2, 32, 32, 128, 2), (2, 8, 128, 512, 2)])
def test_conv_direct(dtype, n, lh, cg, cd, stride):
    rng = np.random.default_rng(10)
    hi, lo, w = make_case(rng, n, lh, cg, cd, stride, dtype)
    bias = rng.normal(size=cd).astype(np.float32)
    g_ref, p_ref, w_ref = oracle_ops(hi, lo, w, stride)
    hi_b, lo_b = U.halo_from(hi, dtype), U.halo_from(lo, dtype)
    w_d = U.dev(w.reshape(-1), U.tdt(dtype))
    bias_d = U.dev(bias)
    out_g = E.HaloBuf(n, lh, lh, cd, dtype, U.DEV)
    L.call("p2p_conv_direct", L.OP_G, stride, dtype, n, lh, lh, cg, cd, C.byref(hi_b.view()), C.byref(out_g.view()),
           U.ptr(w_d), U.ptr(bias_d), None, None, U.stream())
    assert U.rel_err(U.halo_to_np(out_g), g_ref + bias) < OUT_TOL[dtype]
    out_p = E.DenseBuf(n, stride * lh, stride * lh, cg, U.tdt(dtype), U.DEV)
    L.call("p2p_conv_direct", L.OP_P, stride, dtype, n, lh, lh, cg, cd, C.byref(out_p.view()), C.byref(lo_b.view()),
           U.ptr(w_d), None, None, None, U.stream())
    assert U.rel_err(U.dense_to_np(out_p), p_ref) < OUT_TOL[dtype]
    dw = torch.empty(16 * cg * cd, dtype=torch.float32, device=U.DEV)
    db = torch.empty(cd, dtype=torch.float32, device=U.DEV)
    L.call("p2p_conv_direct", L.OP_W, stride, dtype, n, lh, lh, cg, cd, C.byref(hi_b.view()), C.byref(lo_b.view()),
           None, None, U.ptr(dw), U.ptr(db), U.stream())
    assert U.rel_err(dw.cpu().numpy().reshape(4, 4, cg, cd), w_ref) < 2e-5
    assert U.rel_err(db.cpu().numpy(), lo.astype(np.float64).sum(axis=(0, 1, 2))) < 2e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,lh,cg,cd,splitk", [(2, 4, 64, 128, 1), (3, 8, 32, 64, 1), (2, 2, 128, 256, 2), (1, 1, 512, 512, 4),
                                                (7, 1, 64, 128, 1), (130, 1, 128, 64, 2),
                                                (5, 16, 32, 128, 1), (2, 4, 256, 32, 1), (2, 8, 64, 64, 2)])
def test_igemm_ops_G_and_P(dtype, n, lh, cg, cd, splitk):
    rng = np.random.default_rng(11)
    hi, lo, w = make_case(rng, n, lh, cg, cd, 2, dtype)
    g_ref, p_ref, _ = oracle_ops(hi, lo, w, 2)
    hi_b, lo_b = U.halo_from(hi, dtype), U.halo_from(lo, dtype)
    wn = torch.empty(16 * cg * cd, dtype=U.tdt(dtype), device=U.DEV)
    wt = torch.empty(16 * cg * cd, dtype=U.tdt(dtype), device=U.DEV)
    w_d = U.dev(w.reshape(-1))
    L.call("p2p_weight_prep", dtype, U.ptr(w_d), cg, cd, U.ptr(wn), U.ptr(wt), U.stream())
    assert np.array_equal(wn.float().cpu().numpy().reshape(16, cg, cd), w.reshape(16, cg, cd))
    assert np.array_equal(wt.float().cpu().numpy().reshape(16, cd, cg), w.reshape(16, cg, cd).transpose(0, 2, 1))
    esz = 2 if dtype == L.BF16 else 4
    for op, ref, shape in ((L.OP_G, g_ref, (n, lh, lh, cd)), (L.OP_P, p_ref, (n, 2 * lh, 2 * lh, cg))):
        ntaps = 16 if op == L.OP_G else 4
        if lh == 1:     # 1x1 maps contract only the taps that meet real pixels
            ntaps = 4 if op == L.OP_G else 1
        cc = cg if op == L.OP_G else cd
        sk = splitk
        while sk > 1 and (ntaps % sk or ((ntaps // sk) * cc * esz) % 128):
            sk //= 2
        out = E.DenseBuf(*shape, U.tdt(dtype), U.DEV)
        out.t.fill_(float("nan"))
        slabs = torch.full((max(sk, 1) * int(np.prod(shape)),), float("nan"), dtype=torch.float32, device=U.DEV)
        hv, lv = (hi_b.view(), out.view()) if op == L.OP_G else (out.view(), lo_b.view())
        L.call("p2p_igemm", op, dtype, n, lh, lh, cg, cd, C.byref(hv), C.byref(lv), U.ptr(wt if op == L.OP_G else wn),
               sk, U.ptr(slabs) if sk > 1 else None, None, U.stream())
        got = U.dense_to_np(out) if sk == 1 else slabs.view(sk, *shape).sum(0).cpu().numpy()
        tol = OUT_TOL[dtype] if sk == 1 else 2e-5
        assert U.rel_err(got, ref) < tol, (op, sk)


@pytest.mark.parametrize("n,lh", [(2, 32), (3, 64), (5, 4), (300, 8)])
def test_conv_strip_up6(n, lh):
    """up6 (32 <-> 128 channels) through p2p_conv_strip (LDS strip, weights in registers), forward and data gradient,
    against the oracle and bit-for-bit-level close to p2p_igemm on the same operands."""
    dtype, cg, cd = L.BF16, 32, 128
    lw = lh if lh >= 32 else 32            # rows must be >= 32 pixels wide
    rng = np.random.default_rng(21)
    hi = U.q(rng.normal(size=(n, 2 * lh, 2 * lw, cg)), dtype)
    lo = U.q(rng.normal(size=(n, lh, lw, cd)), dtype)
    w = U.q(rng.normal(scale=0.05, size=(4, 4, cg, cd)), dtype)
    hi_t, lo_t, w_t = (torch.tensor(v, dtype=F64) for v in (hi, lo, w))
    g_ref = rg.conv4x4_s2(hi_t, w_t).numpy()
    p_ref = rg.convT4x4_s2(lo_t, w_t).numpy()
    hi_b, lo_b = U.halo_from(hi, dtype), U.halo_from(lo, dtype)
    wn = torch.empty(16 * cg * cd, dtype=U.tdt(dtype), device=U.DEV)
    wt = torch.empty(16 * cg * cd, dtype=U.tdt(dtype), device=U.DEV)
    w_d = U.dev(w.reshape(-1))
    L.call("p2p_weight_prep", dtype, U.ptr(w_d), cg, cd, U.ptr(wn), U.ptr(wt), U.stream())
    assert not L.lib().p2p_conv_strip_ok(L.OP_P, L.F32, n, lh, lw, cg, cd)
    assert not L.lib().p2p_conv_strip_ok(L.OP_P, dtype, n, lh, lw, 64, cd)
    for op, ref, shape in ((L.OP_G, g_ref, (n, lh, lw, cd)), (L.OP_P, p_ref, (n, 2 * lh, 2 * lw, cg))):
        assert L.lib().p2p_conv_strip_ok(op, dtype, n, lh, lw, cg, cd)
        out = E.DenseBuf(*shape, U.tdt(dtype), U.DEV)
        out.t.fill_(float("nan"))
        hv, lv = (hi_b.view(), out.view()) if op == L.OP_G else (out.view(), lo_b.view())
        slots = L.lib().p2p_conv_strip_stat_slots(op, dtype, n, lh, lw, cg, cd)
        assert (slots > 0) == (op == L.OP_P)
        spart = torch.full((max(n * slots * cg * 2, 4),), float("nan"), dtype=torch.float32, device=U.DEV)
        L.call("p2p_conv_strip", op, dtype, n, lh, lw, cg, cd, C.byref(hv), C.byref(lv), U.ptr(wt if op == L.OP_G else wn),
               U.ptr(spart) if slots else None, U.stream())
        got = U.dense_to_np(out)
        assert U.rel_err(got, ref) < OUT_TOL[dtype], op
        if slots:       # pooled slot statistics == per-(image, channel) mean / variance of the stored (rounded) output
            sp = spart.view(n, slots, cg, 2).cpu().numpy().astype(np.float64)
            cnt = got.shape[1] * got.shape[2] / slots
            mean = sp[..., 0].mean(axis=1)
            m2 = (sp[..., 1] + cnt * (sp[..., 0] - mean[:, None, :]) ** 2).sum(axis=1)
            g64 = got.astype(np.float64)
            np.testing.assert_allclose(mean, g64.mean(axis=(1, 2)), atol=2e-5 * np.abs(g64).max())
            np.testing.assert_allclose(m2 / (cnt * slots), g64.var(axis=(1, 2)), rtol=2e-4)
        out2 = E.DenseBuf(*shape, U.tdt(dtype), U.DEV)
        hv, lv = (hi_b.view(), out2.view()) if op == L.OP_G else (out2.view(), lo_b.view())
        L.call("p2p_igemm", op, dtype, n, lh, lw, cg, cd, C.byref(hv), C.byref(lv), U.ptr(wt if op == L.OP_G else wn), 1, None, None,
               U.stream())
        assert U.rel_err(got, U.dense_to_np(out2)) < 1e-2 * OUT_TOL[dtype] + 4e-3, op     # same products, different f32 summation order


def test_conv_strip_is_reproducible_launch_to_launch():
    """p2p_conv_strip stores a tile and starts the next tile's MFMAs in the same wave.  Round 4 found that an MFMA's write-back is not
    interlocked against a pending global store's read of its data registers; the kernel zeroes its accumulators
    with vector moves since.  150 launches per direction at the c2 launch shape (B = 256, persistent workgroups, memory path loaded),
    bit for bit."""
    dtype, n, lh, cg, cd = L.BF16, 256, 32, 32, 128
    g = torch.Generator(device=U.DEV).manual_seed(5)
    hi_b, lo_b = E.HaloBuf(n, 2 * lh, 2 * lh, cg, dtype, U.DEV), E.HaloBuf(n, lh, lh, cd, dtype, U.DEV)
    hi_b.t[:, 2:-2, 2:-2, :] = torch.randn((n, 2 * lh, 2 * lh, cg), device=U.DEV, generator=g).to(U.tdt(dtype))
    lo_b.t[:, 2:-2, 2:-2, :] = torch.randn((n, lh, lh, cd), device=U.DEV, generator=g).to(U.tdt(dtype))
    w_d = (0.05 * torch.randn(16 * cg * cd, device=U.DEV, generator=g)).float()
    wn = torch.empty(16 * cg * cd, dtype=U.tdt(dtype), device=U.DEV)
    wt = torch.empty(16 * cg * cd, dtype=U.tdt(dtype), device=U.DEV)
    L.call("p2p_weight_prep", dtype, U.ptr(w_d), cg, cd, U.ptr(wn), U.ptr(wt), U.stream())
    for op, shape in ((L.OP_G, (n, lh, lh, cd)), (L.OP_P, (n, 2 * lh, 2 * lh, cg))):
        slots = L.lib().p2p_conv_strip_stat_slots(op, dtype, n, lh, lh, cg, cd)
        spart = torch.empty((max(n * slots * cg * 2, 4),), dtype=torch.float32, device=U.DEV)
        first = None
        for i in range(150):
            out = E.DenseBuf(*shape, U.tdt(dtype), U.DEV)
            out.t.fill_(float("nan"))
            hv, lv = (hi_b.view(), out.view()) if op == L.OP_G else (out.view(), lo_b.view())
            L.call("p2p_conv_strip", op, dtype, n, lh, lh, cg, cd, C.byref(hv), C.byref(lv), U.ptr(wt if op == L.OP_G else wn),
                   U.ptr(spart) if slots else None, U.stream())
            if first is None:
                first = out.t.clone()
                assert bool(torch.isfinite(first.float()).all())
            else:
                assert bool((out.t == first).all()), (op, i)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,lh,cg,cd,msplit", [(2, 4, 32, 128, 1), (2, 8, 64, 128, 2), (3, 4, 128, 256, 1), (1, 1, 128, 128, 1),
                                                (2, 16, 32, 128, 4), (5, 2, 64, 256, 1),
                                                # whole 64-pixel stages, 128-multiple channels: the software-pipelined bf16 kernel
                                                # (maps wider / narrower than a stage, 1x1 maps with dead taps, workgroups without pixels)
                                                (4, 4, 128, 256, 1), (8, 8, 128, 128, 2), (64, 1, 128, 128, 1), (16, 2, 256, 128, 2),
                                                (2, 16, 128, 128, 1), (1, 32, 128, 128, 4), (20, 4, 128, 128, 3)])
def test_wgemm(dtype, n, lh, cg, cd, msplit):
    rng = np.random.default_rng(12)
    hi, lo, w = make_case(rng, n, lh, cg, cd, 2, dtype)
    _, _, w_ref = oracle_ops(hi, lo, w, 2)
    hi_b, lo_b = U.halo_from(hi, dtype), U.halo_from(lo, dtype)
    dw = torch.full((16 * cg * cd,), float("nan"), dtype=torch.float32, device=U.DEV)
    ws_bytes = L.lib().p2p_wgemm_workspace_bytes(n, lh, lh, cg, cd, msplit)
    ws = torch.empty(max(ws_bytes // 4, 4), dtype=torch.float32, device=U.DEV)
    L.call("p2p_wgemm", dtype, n, lh, lh, cg, cd, C.byref(hi_b.view()), C.byref(lo_b.view()), U.ptr(dw), msplit,
           U.ptr(ws), U.stream())
    assert U.rel_err(dw.cpu().numpy().reshape(4, 4, cg, cd), w_ref) < 2e-5


# (dtype, channels in front of the slice): 4 = not a whole bf16 vector (scalar forward kernel), 8 = the vector forms; f32: 4 is a vector
@pytest.mark.parametrize("dtype,pad", [(L.F32, 4), (L.BF16, 4), (L.BF16, 8)])
@pytest.mark.parametrize("n,h,c,act,use_mask,norm,nsplit", [
    (2, 4, 64, L.ACT_LEAKY, False, True, 1), (3, 8, 32, L.ACT_RELU, True, True, 1), (2, 1, 512, L.ACT_LEAKY, False, True, 4),
    (2, 4, 36, L.ACT_RELU, True, True, 1), (2, 8, 64, L.ACT_LEAKY, False, False, 2), (2, 32, 32, L.ACT_RELU, False, True, 4),
    (3, 16, 128, L.ACT_LEAKY, True, True, 8),
    (5, 2, 64, L.ACT_RELU, True, True, 1), (3, 3, 16, L.ACT_LEAKY, False, True, 1), (2, 2, 8, L.ACT_RELU, False, False, 1),
    (256, 8, 128, L.ACT_LEAKY, True, True, 4),       # >= 1024 (image, 32-channel) groups: the backward keeps one launch
    # the register-resident forms at 1, 2, 4 and 8 pixels per thread: narrow channel groups (what a small batch picks), and with
    # nsplit | 0x200 the wide groups of batch 256 (a large batch in f32 would meet ReLU gates within an ulp of zero: one flipped
    # gate against the f64 oracle is an 8 % error on its element)
    (2, 16, 128, L.ACT_RELU, False, True, 2), (2, 32, 64, L.ACT_LEAKY, False, True, 4), (3, 8, 256, L.ACT_RELU, True, True, 1),
    (2, 16, 128, L.ACT_RELU, False, True, 0x202), (2, 32, 64, L.ACT_LEAKY, False, True, 0x204), (3, 8, 256, L.ACT_RELU, True, True, 0x201),
    (3, 16, 64, L.ACT_LEAKY, True, True, 0x201), (2, 32, 32, L.ACT_RELU, False, True, 0x201),
    # nsplit | 0x100 (the engine's f32 parity mode): the two-pass forms, whatever the map
    (2, 16, 128, L.ACT_RELU, False, True, 0x102), (3, 8, 256, L.ACT_RELU, True, True, 0x101)])
def test_norm_act_fwd_bwd(dtype, pad, n, h, c, act, use_mask, norm, nsplit):
    rng = np.random.default_rng(13)
    nws = torch.empty(n * 16 * c * 2, dtype=torch.float32, device=U.DEV)
    x = U.q(rng.normal(size=(n, h, h, c)) * 2 + 0.3, dtype)
    gamma = (1 + 0.2 * rng.normal(size=c)).astype(np.float32)
    beta = (0.2 * rng.normal(size=c)).astype(np.float32)
    mask = rng.integers(0, 2, size=(n, h, h, c)).astype(np.uint8) if use_mask else None
    dy1 = U.q(rng.normal(size=(n, h, h, c + 8)), dtype)       # gradient sources with channel offset / f32 slabs
    dy2 = rng.normal(size=(5, n, h, h, c)).astype(np.float32)       # five f32 slabs: one four-slab trip of the loader + its remainder loop
    xt = torch.tensor(x, dtype=F64, requires_grad=True)
    gt = torch.tensor(gamma, dtype=F64, requires_grad=True)
    bt = torch.tensor(beta, dtype=F64, requires_grad=True)
    y = rg.instance_norm(xt, gt, bt) if norm else xt
    if use_mask:
        y = rg.dropout(y, torch.tensor(mask, dtype=F64))
    y = rg.leaky_relu(y) if act == L.ACT_LEAKY else torch.relu(y)
    dy = torch.tensor(dy1[..., 8:], dtype=F64) + torch.tensor(dy2.sum(0), dtype=F64)
    (y * dy).sum().backward()

    raw = E.DenseBuf(n, h, h, c, U.tdt(dtype), U.DEV)
    raw.t.copy_(U.dev(x.reshape(-1, c), U.tdt(dtype)))
    out = E.HaloBuf(n, h, h, c + pad, dtype, U.DEV)
    stats = torch.empty((n, c, 2), dtype=torch.float32, device=U.DEV)
    g_d, b_d = U.dev(gamma), U.dev(beta)
    mask_d = U.dev(mask.reshape(-1, c), torch.uint8) if use_mask else None
    L.call("p2p_norm_act_fwd", dtype, n, h, h, c, raw.ptr(), 1, 1, 0, U.ptr(g_d) if norm else None,
           U.ptr(b_d) if norm else None, 1e-3, act, 0.3, U.ptr(mask_d) if use_mask else None, C.byref(out.view(coff=pad)),
           None, U.ptr(stats) if norm else None, U.ptr(nws), nws.numel() * 4, nsplit, U.stream())
    got = U.halo_to_np(out)
    assert np.count_nonzero(got[..., :pad]) == 0
    assert U.rel_err(got[..., pad:], y.detach().numpy()) < OUT_TOL[dtype]
    assert float(out.t.float().abs().sum()) == pytest.approx(float(np.abs(got).sum()), rel=1e-6)   # halo untouched

    g1 = E.DenseBuf(n, h, h, c + 8, U.tdt(dtype), U.DEV)
    g1.t.copy_(U.dev(dy1.reshape(-1, c + 8), U.tdt(dtype)))
    g2 = U.dev(dy2.reshape(-1))
    gs2 = L.GSrc(g2.data_ptr(), 2, 5, n * h * h * c, c, 0)
    draw = E.HaloBuf(n, h, h, c, dtype, U.DEV)
    part = torch.zeros((2, n, c), dtype=torch.float32, device=U.DEV)
    L.call("p2p_norm_act_bwd", dtype, n, h, h, c, raw.ptr(), U.ptr(stats) if norm else None,
           U.ptr(g_d) if norm else None, U.ptr(b_d) if norm else None, act, 0.3, U.ptr(mask_d) if use_mask else None,
           C.byref(g1.gsrc(coff=8)), C.byref(gs2), C.byref(draw.view()), U.ptr(part[1]) if norm else None,
           U.ptr(part[0]) if norm else None, U.ptr(nws), nws.numel() * 4, nsplit, U.stream())
    ref_dx = xt.grad.numpy()
    scale = np.abs(ref_dx).max() + 1e-30
    assert np.abs(U.halo_to_np(draw) - ref_dx).max() / scale < (1e-4 if dtype == L.F32 else 1e-2)
    if norm:
        dgam = torch.empty(c, dtype=torch.float32, device=U.DEV)
        dbet = torch.empty(c, dtype=torch.float32, device=U.DEV)
        L.call("p2p_colsum", U.ptr(part[1]), n, c, 1.0, U.ptr(dgam), U.stream())
        L.call("p2p_colsum", U.ptr(part[0]), n, c, 1.0, U.ptr(dbet), U.stream())
        assert U.rel_err(dgam.cpu().numpy(), gt.grad.numpy()) < 1e-4
        assert U.rel_err(dbet.cpu().numpy(), bt.grad.numpy()) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,h,c,nslabs", [(2, 4, 64, 16), (3, 2, 512, 7), (2, 8, 256, 5), (3, 16, 128, 4), (2, 32, 64, 2), (2, 64, 32, 3)])
def test_norm_act_fwd_from_split_k_slabs(dtype, n, h, c, nslabs):
    """The forward kernels fed by a K-split convolution (raw_kind 2): the f32 slabs are summed in slab order, rounded through the
    activation dtype (written to raw_out: what the backward pass reads) and normalised -- the lane-group form (4x4, 2x2), the
    register-resident form with the four-slabs-per-trip loader and its remainder loop (8x8 ... 32x32) and the two-pass form (64x64)."""
    rng = np.random.default_rng(17)
    slabs = (rng.normal(size=(nslabs, n, h, h, c)) * 0.8 + 0.1).astype(np.float32)
    acc = np.zeros((n, h, h, c), np.float32)
    for k in range(nslabs):
        acc = acc + slabs[k]                      # f32, slab order: bit for bit what the loader computes
    x = U.q(acc, dtype)
    gamma = (1 + 0.2 * rng.normal(size=c)).astype(np.float32)
    beta = (0.2 * rng.normal(size=c)).astype(np.float32)
    y = rg.leaky_relu(rg.instance_norm(torch.tensor(x, dtype=F64), torch.tensor(gamma, dtype=F64), torch.tensor(beta, dtype=F64))).numpy()
    slabs_d = U.dev(slabs.reshape(-1))
    raw_out = E.DenseBuf(n, h, h, c, U.tdt(dtype), U.DEV)
    raw_out.t.fill_(float("nan"))
    out = E.HaloBuf(n, h, h, c + 8, dtype, U.DEV)
    stats = torch.empty((n, c, 2), dtype=torch.float32, device=U.DEV)
    nws = torch.empty(n * 16 * c * 2, dtype=torch.float32, device=U.DEV)
    g_d, b_d = U.dev(gamma), U.dev(beta)
    L.call("p2p_norm_act_fwd", dtype, n, h, h, c, U.ptr(slabs_d), 2, nslabs, n * h * h * c, U.ptr(g_d), U.ptr(b_d), 1e-3, L.ACT_LEAKY, 0.3,
           None, C.byref(out.view(coff=8)), raw_out.ptr(), U.ptr(stats), U.ptr(nws), nws.numel() * 4, 1, U.stream())
    assert np.array_equal(U.dense_to_np(raw_out), x)
    got = U.halo_to_np(out)
    assert np.count_nonzero(got[..., :8]) == 0
    assert U.rel_err(got[..., 8:], y) < OUT_TOL[dtype]
    mean = x.astype(np.float64).mean(axis=(1, 2))
    assert np.abs(stats[..., 0].cpu().numpy() - mean).max() < 1e-5 * max(1.0, np.abs(mean).max())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,h,c,nsplit,stats_mode", [(3, 8, 32, 1, False), (2, 32, 32, 4, False), (2, 16, 64, 1, False), (2, 8, 128, 1, False),
                                                     (2, 64, 32, 4, False)])
def test_norm_act_fwd_tail_writes_whole_concat_pixels(dtype, n, h, c, nsplit, stats_mode):
    """p2p_norm_act_fwd_tail = p2p_norm_act_fwd + the copy of the following 8 channels from another view (the last concat of the
    generator is [up6 | input image], networks.py:92-94): bit-identical to the two separate writes at the product's shape (64x64 maps),
    halo untouched.  Smaller maps: p2p_norm_act_fwd alone is served by the register-resident form (another order of the sums), so the
    normalised channels agree to the last bits instead; the copied channels and the halo stay exact."""
    rng = np.random.default_rng(31)
    nws = torch.empty(n * 16 * c * 2, dtype=torch.float32, device=U.DEV)
    x = U.q(rng.normal(size=(n, h, h, c)) * 2 + 0.3, dtype)
    tail = U.q(rng.normal(size=(n, h, h, 8)), dtype)
    gamma, beta = U.dev((1 + 0.2 * rng.normal(size=c)).astype(np.float32)), U.dev((0.2 * rng.normal(size=c)).astype(np.float32))
    raw = E.DenseBuf(n, h, h, c, U.tdt(dtype), U.DEV)
    raw.t.copy_(U.dev(x.reshape(-1, c), U.tdt(dtype)))
    tb = U.halo_from(tail, dtype)
    outs = []
    for with_tail in (False, True):
        out = E.HaloBuf(n, h, h, c + 8, dtype, U.DEV)
        stats = torch.empty((n, c, 2), dtype=torch.float32, device=U.DEV)
        args = (dtype, n, h, h, c, raw.ptr(), 1, 1, 0, U.ptr(gamma), U.ptr(beta), 1e-3, L.ACT_RELU, 0.3, None, C.byref(out.view(coff=0)),
                None, U.ptr(stats), U.ptr(nws), nws.numel() * 4, nsplit)
        if with_tail:
            L.call("p2p_norm_act_fwd_tail", *args, C.byref(tb.view()), 8, U.stream())
        else:
            L.call("p2p_norm_act_fwd", *args, U.stream())
            out.t[:, E.HALO:E.HALO + h, E.HALO:E.HALO + h, c:] = tb.t[:, E.HALO:E.HALO + h, E.HALO:E.HALO + h, :]
        outs.append(out.t.clone())
    if h * h > 2048:
        assert torch.equal(outs[0], outs[1])
    else:
        assert torch.equal(outs[0][..., c:], outs[1][..., c:])
        halo = torch.ones_like(outs[0][..., 0], dtype=torch.bool)
        halo[:, E.HALO:E.HALO + h, E.HALO:E.HALO + h] = False
        assert torch.equal(outs[0][halo], outs[1][halo])
        assert (outs[0][..., :c].float() - outs[1][..., :c].float()).abs().max().item() <= (1e-5 if dtype == L.F32 else 0.04)
    inner = outs[1][:, E.HALO:E.HALO + h, E.HALO:E.HALO + h, c:].float().cpu().numpy()
    assert np.array_equal(inner, tail.astype(np.float32))
    # small maps and ragged channel counts are refused, not silently served without the tail
    out = E.HaloBuf(n, 4, 4, c + 8, dtype, U.DEV)
    with pytest.raises(L.P2PError):
        L.call("p2p_norm_act_fwd_tail", dtype, n, 4, 4, c, raw.ptr(), 1, 1, 0, U.ptr(gamma), U.ptr(beta), 1e-3, L.ACT_RELU, 0.3, None,
               C.byref(out.view(coff=0)), None, U.ptr(stats), U.ptr(nws), nws.numel() * 4, 1, C.byref(tb.view()), 8, U.stream())


@pytest.mark.parametrize("dtype", DTYPES)
def test_tanh_l1_fwd_pair_equals_the_two_partial_writes(dtype):
    """p2p_tanh_l1_fwd_pair writes whole [tanh(z) | source] pixels of the discriminator's fake input (networks.py:45); the
    4-channel form + a separate copy of the source half give the same buffer bit for bit, the same f32 copy and the same L1."""
    rng = np.random.default_rng(32)
    n, s = 3, 32
    z = U.q(rng.normal(size=(n, s, s, 4)), dtype)
    pair = U.q(rng.uniform(-1, 1, size=(n, s, s, 8)), dtype)           # [target | source]
    zb, rb = U.halo_from(z, dtype), U.halo_from(pair, dtype)
    inv = 1.0 / (n * s * s * 4)
    res = []
    for whole in (False, True):
        fb = E.HaloBuf(n, s, s, 8, dtype, U.DEV)
        part = torch.zeros(256, dtype=torch.float32, device=U.DEV)
        f32copy = torch.empty((n, s, s, 4), dtype=torch.float32, device=U.DEV)
        loss = torch.zeros(1, dtype=torch.float32, device=U.DEV)
        if whole:
            L.call("p2p_tanh_l1_fwd_pair", dtype, n, s, s, C.byref(zb.view()), C.byref(rb.view()), C.byref(fb.view()), inv, U.ptr(part),
                   U.ptr(f32copy), U.stream())
        else:
            L.call("p2p_tanh_l1_fwd", dtype, n, s, s, 4, C.byref(zb.view()), C.byref(rb.view()), C.byref(fb.view()), inv, U.ptr(part),
                   U.ptr(f32copy), U.stream())
            fb.t[..., 4:] = rb.t[..., 4:]
        L.call("p2p_loss_partials_sum", U.ptr(part), 1, U.ptr(loss), U.stream())
        res.append((fb.t.clone(), f32copy.clone(), float(loss[0])))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert abs(res[0][2] - res[1][2]) <= 2e-6 * abs(res[0][2])          # same terms, another (fixed) summation order


@pytest.mark.parametrize("dtype", DTYPES)
def test_losses(dtype):
    rng = np.random.default_rng(14)
    n2, nr, h = 6, 3, 8
    logits = U.q(rng.normal(size=(n2, h, h, 1)) * 3, dtype)
    lb = U.halo_from(logits, dtype)
    dld, dlg = E.HaloBuf(n2, h, h, 1, dtype, U.DEV), E.HaloBuf(n2 - nr, h, h, 1, dtype, U.DEV)
    loss = torch.zeros(8, dtype=torch.float32, device=U.DEV)
    inv = 1.0 / (nr * h * h)
    part = torch.full((4 * 256,), float("nan"), dtype=torch.float32, device=U.DEV)     # rows 0..2 BCE, row 3 L1
    L.call("p2p_bce_logits", dtype, n2, nr, h, h, C.byref(lb.view()), inv, C.byref(dld.view()), C.byref(dlg.view()),
           U.ptr(part), U.stream())
    L.call("p2p_loss_partials_sum", U.ptr(part), 3, U.ptr(loss), U.stream())
    lt = torch.tensor(logits, dtype=F64, requires_grad=True)
    real, fake, adv = rg.bce_from_logits(lt[:nr], 1.0), rg.bce_from_logits(lt[nr:], 0.0), rg.bce_from_logits(lt[nr:], 1.0)
    got = loss.cpu().numpy()
    np.testing.assert_allclose(got[:3], [float(real), float(fake), float(adv)], rtol=1e-5)
    gd = torch.autograd.grad(real + fake, lt, retain_graph=True)[0].numpy()
    gg = torch.autograd.grad(adv, lt)[0].numpy()[nr:]
    tol = 1e-5 if dtype == L.F32 else 6e-3
    assert U.rel_err(U.halo_to_np(dld), gd) < tol
    assert U.rel_err(U.halo_to_np(dlg), gg) < tol

    n, s, c = 3, 16, 4
    z = U.q(rng.normal(size=(n, s, s, c)), dtype)
    real_img = U.q(rng.uniform(-1, 1, size=(n, s, s, c)), dtype)
    gd_src = U.q(rng.normal(size=(n, s, s, 8)), dtype)
    zb, rb = U.halo_from(z, dtype), U.halo_from(real_img, dtype)
    fb, dzb = E.HaloBuf(n, s, s, c, dtype, U.DEV), E.HaloBuf(n, s, s, c, dtype, U.DEV)
    inv = 1.0 / (n * s * s * c)
    l1_row = part[3 * 256:]
    f32copy = torch.empty((n, s, s, c), dtype=torch.float32, device=U.DEV)
    L.call("p2p_tanh_l1_fwd", dtype, n, s, s, c, C.byref(zb.view()), C.byref(rb.view()), C.byref(fb.view()), inv,
           U.ptr(l1_row), U.ptr(f32copy), U.stream())
    assert np.abs(f32copy.cpu().numpy() - np.tanh(z.astype(np.float64))).max() < 1e-6       # unrounded copy for the histogram loss
    L.call("p2p_loss_partials_sum", U.ptr(part), 4, U.ptr(loss), U.stream())
    zt = torch.tensor(z, dtype=F64, requires_grad=True)
    fake_t = torch.tanh(zt)
    l1 = (torch.tensor(real_img, dtype=F64) - fake_t).abs().mean()
    assert U.rel_err(U.halo_to_np(fb), fake_t.detach().numpy()) < OUT_TOL[dtype]
    assert abs(float(loss[3]) - float(l1)) < (1e-5 if dtype == L.F32 else 3e-3) * float(l1)
    gsrc = E.DenseBuf(n, s, s, 8, U.tdt(dtype), U.DEV)
    gsrc.t.copy_(U.dev(gd_src.reshape(-1, 8), U.tdt(dtype)))
    lam = 100.0
    L.call("p2p_tanh_l1_bwd", dtype, n, s, s, c, C.byref(fb.view()), C.byref(rb.view()), C.byref(gsrc.gsrc()), None,
           lam * inv, C.byref(dzb.view()), U.stream())
    (lam * l1 + (fake_t * torch.tensor(gd_src[..., :c], dtype=F64)).sum()).backward()
    # sign(fake-real) flips where bf16 rounding of fake crosses real: compare where |fake-real| is not tiny
    ok = np.abs(fake_t.detach().numpy() - real_img) > 1e-2
    err = np.abs(U.halo_to_np(dzb) - zt.grad.numpy())[ok].max() / np.abs(zt.grad.numpy()).max()
    assert err < (1e-5 if dtype == L.F32 else 1e-2)


def test_adam_matches_keras_formulation():
    rng = np.random.default_rng(15)
    n = 10007
    p, g = rng.normal(size=n).astype(np.float32), (rng.normal(size=n) * 1e-3).astype(np.float32)
    m, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
    pd, gd, md, vd = U.dev(p), U.dev(g), U.dev(m), U.dev(v)
    pr, mr, vr = p.astype(np.float64), m.astype(np.float64), v.astype(np.float64)
    for t in (1, 2, 3):
        L.call("p2p_adam_flat", U.ptr(pd), U.ptr(gd), U.ptr(md), U.ptr(vd), n, t, 2e-4, 0.5, 0.999, 1e-7, 1.0, U.stream())
        pr, mr, vr = npr.keras_adam_step(pr, g.astype(np.float64), mr, vr, t)
    np.testing.assert_allclose(pd.cpu().numpy(), pr, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(vd.cpu().numpy(), vr, rtol=5e-5)   # (1 - 0.999f) in f32, as keras does


def test_adam_device_step_state_matches_keras_formulation():
    """The variant the engine runs: p2p_adam_tick advances t and the step size in device memory, p2p_adam_flat_dev reads it
    (tf.keras.optimizers.Adam(2e-4, beta_1=0.5), pix2pix_model.py:28-29,81-83)."""
    rng = np.random.default_rng(14)
    n = 5003
    p0 = rng.normal(size=n).astype(np.float32)
    pd, md, vd = U.dev(p0), torch.zeros(n, device=U.DEV), torch.zeros(n, device=U.DEV)
    t_dev = torch.zeros(1, dtype=torch.int32, device=U.DEV)
    lr_dev = torch.zeros(1, dtype=torch.float32, device=U.DEV)
    pr, mr, vr = p0.astype(np.float64), np.zeros(n), np.zeros(n)
    for t in range(1, 5):
        g = (rng.normal(size=n) * 10.0 ** rng.integers(-6, 1, size=n)).astype(np.float32)
        gd = U.dev(g)
        L.call("p2p_adam_tick", U.ptr(t_dev), U.ptr(lr_dev), 2e-4, 0.5, 0.999, U.stream())
        L.call("p2p_adam_flat_dev", U.ptr(pd), U.ptr(gd), U.ptr(md), U.ptr(vd), n, U.ptr(lr_dev), 0.5, 0.999, 1e-7, 1.0, U.stream())
        pr, mr, vr = npr.keras_adam_step(pr, g.astype(np.float64), mr, vr, t)
        assert int(t_dev[0]) == t
        assert abs(float(lr_dev[0]) - 2e-4 * np.sqrt(1 - 0.999 ** t) / (1 - 0.5 ** t)) < 1e-9
    np.testing.assert_allclose(pd.cpu().numpy(), pr, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(vd.cpu().numpy(), vr, rtol=5e-5)


def test_dropout_mask_of_a_shard_is_the_slice_of_the_global_mask():
    """Data parallelism: rank r draws the keep-mask of its samples from the global element index, so N ranks
    together reproduce the single-process mask of the global batch."""
    per, Bg = 2 * 2 * 512, 6                     # elements per sample of up1's mask, global batch
    cnt = torch.tensor([3], dtype=torch.int64, device=U.DEV)
    whole = torch.empty(Bg * per, dtype=torch.uint8, device=U.DEV)
    L.call("p2p_dropout_mask_dev", U.ptr(whole), Bg * per, 47, U.ptr(cnt), 1, 0, U.stream())
    for lo, hi in ((0, 2), (2, 5), (5, 6)):      # ragged shards
        part = torch.empty((hi - lo) * per, dtype=torch.uint8, device=U.DEV)
        L.call("p2p_dropout_mask_dev", U.ptr(part), (hi - lo) * per, 47, U.ptr(cnt), 1, lo * per, U.stream())
        assert torch.equal(part, whole[lo * per:hi * per])
    other = torch.empty(Bg * per, dtype=torch.uint8, device=U.DEV)
    L.call("p2p_dropout_mask_dev", U.ptr(other), Bg * per, 47, U.ptr(cnt), 2, 0, U.stream())     # another layer: another stream
    assert abs(float((other == whole).float().mean()) - 0.5) < 2e-2


def test_dropout_mask_is_fair_and_reproducible():
    n = 1 << 20
    a = torch.empty(n, dtype=torch.uint8, device=U.DEV)
    b = torch.empty(n, dtype=torch.uint8, device=U.DEV)
    L.call("p2p_dropout_mask", U.ptr(a), n, 47, 1, U.stream())
    L.call("p2p_dropout_mask", U.ptr(b), n, 47, 1, U.stream())
    assert torch.equal(a, b) and int(a.max()) == 1
    assert abs(float(a.float().mean()) - 0.5) < 5e-3
    L.call("p2p_dropout_mask", U.ptr(b), n, 47, 2, U.stream())
    assert abs(float((a == b).float().mean()) - 0.5) < 5e-3


def test_bad_arguments_fail_loudly():
    t = L.Tensor(0, 0, 0, 0)
    with pytest.raises(L.P2PError):
        L.call("p2p_igemm", L.OP_G, L.BF16, 1, 4, 4, 48, 64, C.byref(t), C.byref(t), None, 1, None, None, None)
    with pytest.raises(L.P2PError):
        L.call("p2p_conv_direct", 7, 2, L.F32, 1, 4, 4, 4, 4, C.byref(t), C.byref(t), None, None, None, None, None)


@pytest.mark.parametrize("dtype", DTYPES)
def test_weight_prep_batched_equals_per_layer(dtype):
    """One batched launch (64x64 vector tiles where the shape allows, 32x32 tiles and padding elsewhere) == the per-layer
    kernel, bit for bit."""
    rng = np.random.default_rng(23)
    specs = [(64, 128, 64, 128, 128, 64), (128, 64, 128, 64, 64, 128), (32, 128, 32, 128, 128, 32),
             (36, 4, 64, 8, 32, 40), (4, 64, 32, 64, 64, 8), (512, 512, 512, 512, 512, 512)]
    tasks = (L.PrepTask * len(specs))()
    keep, first = [], 0
    for k, (cg, cd, wn_r, wn_c, wt_r, wt_c) in enumerate(specs):
        w = U.dev(rng.normal(size=16 * cg * cd).astype(np.float32))
        wn = torch.full((16 * wn_r * wn_c,), float("nan"), dtype=U.tdt(dtype), device=U.DEV)
        wt = torch.full((16 * wt_r * wt_c,), float("nan"), dtype=U.tdt(dtype), device=U.DEV)
        wn_ref, wt_ref = torch.empty_like(wn), torch.empty_like(wt)
        L.call("p2p_weight_prep_pad", dtype, U.ptr(w), cg, cd, U.ptr(wn_ref), wn_r, wn_c, U.ptr(wt_ref), wt_r, wt_c, U.stream())
        tg, td = C.c_int(0), C.c_int(0)
        nb = L.lib().p2p_weight_prep_task_blocks(cg, cd, wn_r, wn_c, wt_r, wt_c, 1, 1, C.byref(tg), C.byref(td))
        t = tasks[k]
        t.w, t.wn, t.wt = w.data_ptr(), wn.data_ptr(), wt.data_ptr()
        t.Cg, t.Cd, t.wn_rows, t.wn_cols, t.wt_rows, t.wt_cols = cg, cd, wn_r, wn_c, wt_r, wt_c
        t.tiles_g, t.tiles_d, t.first_block = tg.value, td.value, first
        first += nb
        keep.append((w, wn, wt, wn_ref, wt_ref))
    table = torch.frombuffer(bytearray(bytes(tasks)), dtype=torch.uint8).to(U.DEV)
    L.call("p2p_weight_prep_batched", dtype, U.ptr(table), len(specs), first, U.stream())
    for w, wn, wt, wn_ref, wt_ref in keep:
        assert torch.equal(wn.view(torch.int16 if dtype == L.BF16 else torch.int32), wn_ref.view(torch.int16 if dtype == L.BF16 else torch.int32))
        assert torch.equal(wt.view(torch.int16 if dtype == L.BF16 else torch.int32), wt_ref.view(torch.int16 if dtype == L.BF16 else torch.int32))


def _pad_view_input(x, cpad, dtype):
    """numpy (N,H,W,C) -> HaloBuf with C padded (zeros) to cpad channels."""
    n, h, w, c = x.shape
    xp = np.zeros((n, h, w, cpad), np.float32)
    xp[..., :c] = x
    return U.halo_from(xp, dtype)


def _pick_edge_entry(entry, op, stride, dtype, n, lh, cin_pad, nc):
    """The specialised edge kernels cover part of the shape space; like the engine, fall back to p2p_igemm_edge."""
    if entry == "p2p_igemm_edge":
        return entry
    ok = getattr(L.lib(), entry + "_ok")(op, stride, dtype, n, lh, lh, cin_pad, nc)
    return entry if ok else "p2p_igemm_edge"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,lh,cg,cd,stride,g_ok,p_ok", [
    (2, 64, 36, 4, 1, True, False),      # generator head 36(+4) -> 4, stride 1
    (3, 32, 64, 1, 1, True, False),      # discriminator head 64 -> 1
    (2, 32, 4, 64, 2, False, True),      # d(discriminator first conv)/d(fake image): 64 -> 4, transposed stride 2
    (2, 16, 1, 64, 2, False, True),      # indexed model: 64 -> 1
    (1, 128, 33, 3, 1, True, False),     # 128-pixel rows, 3 real output channels
    (5, 8, 60, 2, 1, True, False)])
def test_edge_layers_few_outputs(dtype, n, lh, cg, cd, stride, g_ok, p_ok):
    """Same layers through p2p_conv_fewout (tap-major GEMM + shifted sum out of LDS)."""
    hi_pad, lo_pad = E.pad8(cg), E.pad8(cd)
    assert bool(L.lib().p2p_conv_fewout_ok(L.OP_G, stride, dtype, n, lh, lh, hi_pad, cd)) == g_ok
    assert bool(L.lib().p2p_conv_fewout_ok(L.OP_P, stride, dtype, n, lh, lh, lo_pad, min(cg, 32))) == p_ok
    test_edge_layers_on_mfma(dtype, n, lh, cg, cd, stride, entry="p2p_conv_fewout")


@pytest.mark.parametrize("n,lh,cg,cd,stride,g_ok,p_ok", [
    (2, 32, 4, 64, 2, True, False),      # down1: RGBA(+4) -> 64, stride 2
    (3, 32, 8, 64, 2, True, False),      # discriminator first conv: 8 -> 64
    (2, 64, 36, 4, 1, False, True),      # d(generator head)/d(concat): 4(+4) -> 32 of 36, transposed stride 1
    (3, 32, 64, 1, 1, False, True),      # d(discriminator head)/d(features): 1(+7) -> 64 (32 columns here)
    (2, 16, 1, 64, 2, True, False),      # indexed input, 16-pixel rows: MFMA tiles straddle rows
    (5, 8, 3, 48, 1, True, False),       # stride 1, 8-pixel rows, 48 outputs (two masked chunks)
    (2, 8, 5, 20, 1, True, False)])      # ragged last chunk
def test_edge_layers_few_inputs(n, lh, cg, cd, stride, g_ok, p_ok):
    """Same layers through p2p_conv_fewin (bf16 only: weights in registers, strip in LDS, one K step = two taps)."""
    dtype = L.BF16
    hi_pad, lo_pad = E.pad8(cg), E.pad8(cd)
    assert bool(L.lib().p2p_conv_fewin_ok(L.OP_G, stride, dtype, n, lh, lh, hi_pad, cd)) == g_ok
    assert bool(L.lib().p2p_conv_fewin_ok(L.OP_P, stride, dtype, n, lh, lh, lo_pad, min(cg, 32))) == p_ok
    assert not L.lib().p2p_conv_fewin_ok(L.OP_G, stride, L.F32, n, lh, lh, hi_pad, cd)
    test_edge_layers_on_mfma(dtype, n, lh, cg, cd, stride, entry="p2p_conv_fewin")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,lh,cg,cd,stride", [(2, 8, 4, 64, 2), (3, 8, 8, 64, 2), (2, 16, 36, 4, 1), (2, 8, 64, 1, 1),
                                                (1, 4, 33, 256, 1), (2, 8, 1, 64, 2)])
def test_edge_layers_on_mfma(dtype, n, lh, cg, cd, stride, entry="p2p_igemm_edge"):
    """The edge layers (networks.py:46-48,57,75-78) through p2p_igemm_edge / p2p_wgemm_edge: padded channel
    counts, stride 1 and 2, bias + LeakyReLU epilogue, masked columns."""
    rng = np.random.default_rng(16)
    hi, lo, w = make_case(rng, n, lh, cg, cd, stride, dtype)
    bias = rng.normal(size=cd).astype(np.float32)
    g_ref, p_ref, w_ref = oracle_ops(hi, lo, w, stride)
    hi_pad, lo_pad = E.pad8(cg), E.pad8(cd)
    hi_b, lo_b = _pad_view_input(hi, hi_pad, dtype), _pad_view_input(lo, lo_pad, dtype)
    wt = torch.zeros(16 * E.up32(cd) * hi_pad, dtype=U.tdt(dtype), device=U.DEV)
    wn = torch.zeros(16 * E.up32(cg) * lo_pad, dtype=U.tdt(dtype), device=U.DEV)
    w_d = U.dev(w.reshape(-1))
    L.call("p2p_weight_prep_pad", dtype, U.ptr(w_d), cg, cd, U.ptr(wn), E.up32(cg), lo_pad,
           U.ptr(wt), E.up32(cd), hi_pad, U.stream())
    # op G with bias + LeakyReLU into a channel slice of a wider haloed buffer
    out_g = E.HaloBuf(n, lh, lh, cd + 8, dtype, U.DEV)
    bias_d = U.dev(bias)
    e_g = _pick_edge_entry(entry, L.OP_G, stride, dtype, n, lh, hi_pad, cd)
    L.call(e_g, L.OP_G, stride, dtype, n, lh, lh, hi_pad, cd, E.up32(cd), C.byref(hi_b.view()),
           C.byref(out_g.view(coff=8)), U.ptr(wt), U.ptr(bias_d), L.ACT_LEAKY, 0.3, U.stream())
    want = g_ref + bias
    want = np.where(want > 0, want, 0.3 * want)
    got = U.halo_to_np(out_g)
    assert np.count_nonzero(got[..., :8]) == 0
    assert U.rel_err(got[..., 8:], want) < OUT_TOL[dtype]
    # op P restricted to the first ncols output channels
    ncols = min(cg, 32)
    out_p = E.DenseBuf(n, stride * lh, stride * lh, hi_pad, U.tdt(dtype), U.DEV)
    out_p.t.zero_()
    e_p = _pick_edge_entry(entry, L.OP_P, stride, dtype, n, lh, lo_pad, ncols)
    L.call(e_p, L.OP_P, stride, dtype, n, lh, lh, lo_pad, ncols, E.up32(cg), C.byref(lo_b.view()),
           C.byref(out_p.view()), U.ptr(wn), None, L.ACT_NONE, 0.0, U.stream())
    gp = U.dense_to_np(out_p)
    assert U.rel_err(gp[..., :ncols], p_ref[..., :ncols]) < OUT_TOL[dtype] * max(1.0, np.abs(p_ref).max() / (np.abs(p_ref[..., :ncols]).max() + 1e-30))
    assert np.count_nonzero(gp[..., ncols:]) == 0
    # op W with masked stores
    for msplit in (1, 2):
        dw = torch.full((16 * cg * cd,), float("nan"), dtype=torch.float32, device=U.DEV)
        ws = torch.empty(max(L.lib().p2p_wgemm_workspace_bytes(n, lh, lh, cg, cd, msplit) // 4, 4), dtype=torch.float32, device=U.DEV)
        L.call("p2p_wgemm_edge", dtype, stride, n, lh, lh, cg, cd, C.byref(hi_b.view()), C.byref(lo_b.view()), U.ptr(dw),
               msplit, U.ptr(ws), U.stream())
        assert U.rel_err(dw.cpu().numpy().reshape(4, 4, cg, cd), w_ref) < 2e-5, msplit
    db = torch.empty(cd, dtype=torch.float32, device=U.DEV)
    lv = lo_b.view()
    cs_ws = torch.empty(max(16, L.lib().p2p_view_colsum_workspace_bytes(dtype, n, lh, lh, cd, C.byref(lv)) // 4), dtype=torch.float32, device=U.DEV)
    L.call("p2p_view_colsum", dtype, n, lh, lh, cd, C.byref(lv), U.ptr(db), U.ptr(cs_ws), U.stream())
    assert U.rel_err(db.cpu().numpy(), lo.astype(np.float64).sum(axis=(0, 1, 2))) < 2e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("op,n,lh,cg,cd", [(L.OP_G, 8, 8, 64, 128), (L.OP_P, 8, 8, 64, 128), (L.OP_G, 4, 16, 32, 64),
                                            (L.OP_P, 64, 4, 128, 256), (L.OP_P, 128, 32, 32, 128)])
def test_igemm_fused_instance_norm_statistics(dtype, op, n, lh, cg, cd):
    """InstanceNorm statistics produced by the GEMM epilogue (slot partials pooled by the parallel-variance rule) and
    consumed by the apply-only norm pass == the unfused path == the oracle."""
    rng = np.random.default_rng(17)
    hi, lo, w = make_case(rng, n, lh, cg, cd, 2, dtype)
    hi_b, lo_b = U.halo_from(hi, dtype), U.halo_from(lo, dtype)
    wn = torch.empty(16 * cg * cd, dtype=U.tdt(dtype), device=U.DEV)
    wt = torch.empty(16 * cg * cd, dtype=U.tdt(dtype), device=U.DEV)
    w_d = U.dev(w.reshape(-1))
    L.call("p2p_weight_prep", dtype, U.ptr(w_d), cg, cd, U.ptr(wn), U.ptr(wt), U.stream())
    ncols = cd if op == L.OP_G else cg
    res = lh if op == L.OP_G else 2 * lh
    slots = L.lib().p2p_igemm_layer_stat_slots(op, dtype, n, lh, lh, cg, cd)     # of whichever kernel p2p_igemm uses for the layer
    assert slots > 0
    assert L.lib().p2p_igemm_stat_slots(L.OP_P, n, 2, 2, ncols) == 0      # <= 16-pixel output maps take their own statistics
    out = E.DenseBuf(n, res, res, ncols, U.tdt(dtype), U.DEV)
    spart = torch.full((n * slots * ncols * 2,), float("nan"), dtype=torch.float32, device=U.DEV)
    hv, lv = (hi_b.view(), out.view()) if op == L.OP_G else (out.view(), lo_b.view())
    L.call("p2p_igemm", op, dtype, n, lh, lh, cg, cd, C.byref(hv), C.byref(lv), U.ptr(wt if op == L.OP_G else wn), 1, None,
           U.ptr(spart), U.stream())
    x = U.dense_to_np(out).astype(np.float64)                 # the rounded values the statistics must describe
    sp = spart.view(n, slots, ncols, 2).cpu().numpy().astype(np.float64)
    cnt = res * res / slots
    mean = sp[..., 0].mean(1)
    m2 = sp[..., 1].sum(1) + cnt * ((sp[..., 0] - mean[:, None]) ** 2).sum(1)
    np.testing.assert_allclose(mean, x.mean(axis=(1, 2)), rtol=1e-4, atol=1e-5 * np.abs(x).max())
    np.testing.assert_allclose(m2 / (res * res), x.var(axis=(1, 2)), rtol=2e-4)
    gamma = (1 + 0.2 * rng.normal(size=ncols)).astype(np.float32)
    beta = (0.2 * rng.normal(size=ncols)).astype(np.float32)
    y = E.HaloBuf(n, res, res, ncols, dtype, U.DEV)
    stats = torch.empty((n, ncols, 2), dtype=torch.float32, device=U.DEV)
    g_d, b_d = U.dev(gamma), U.dev(beta)          # keep the device tensors alive across the launch
    L.call("p2p_norm_act_fwd", dtype, n, res, res, ncols, out.ptr(), 1, 1, 0, U.ptr(g_d), U.ptr(b_d), 1e-3,
           L.ACT_RELU, 0.3, None, C.byref(y.view()), None, U.ptr(stats), U.ptr(spart), spart.numel() * 4, -slots, U.stream())
    want = torch.relu(rg.instance_norm(torch.tensor(x), torch.tensor(gamma, dtype=F64), torch.tensor(beta, dtype=F64))).numpy()
    assert U.rel_err(U.halo_to_np(y), want) < OUT_TOL[dtype]
    np.testing.assert_allclose(stats[..., 0].cpu().numpy(), x.mean(axis=(1, 2)), rtol=1e-4, atol=1e-5 * np.abs(x).max())


@pytest.mark.parametrize("cbw", [1, 2])
@pytest.mark.parametrize("op,n,lh,cg,cd", [
    (L.OP_P, 3, 8, 64, 64), (L.OP_P, 5, 8, 128, 32), (L.OP_P, 9, 8, 64, 96), (L.OP_P, 2, 16, 64, 96), (L.OP_P, 1, 16, 128, 64),
    (L.OP_P, 2, 32, 64, 32), (L.OP_P, 1, 64, 64, 32),
    (L.OP_G, 3, 8, 32, 256), (L.OP_G, 5, 8, 64, 256), (L.OP_G, 2, 16, 64, 256), (L.OP_G, 1, 16, 96, 512), (L.OP_G, 2, 32, 32, 256),
    (L.OP_G, 1, 64, 32, 256), (L.OP_G, 2, 16, 64, 128)])
def test_igemm_block_resident_wide_maps(op, n, lh, cg, cd, cbw, monkeypatch):
    """bf16 wide maps: p2p_igemm runs the block-resident kernel (csrc/brig.hip: input block in LDS once, phases merged /
    parity planes walked, weights streamed); outputs and fused InstanceNorm statistics against the oracle, incl. ragged
    image groups (n = 3, 5, 9 on 8x8 maps: 4 images per workgroup), several K chunks and output-channel tiles, and both
    wave tilings (64 or 32 output channels per wave)."""
    if cbw == 2 and (cg if op == L.OP_P else cd) % (64 if op == L.OP_P else 256):
        pytest.skip("the 64-channels-per-wave form needs whole 64 / 256 channel tiles")
    monkeypatch.setenv("P2P_BRIG_CBW", str(cbw))
    monkeypatch.setenv("P2P_BRIG_MIN_WG", "1")       # (the product keeps this kernel for launches that fill the chip)
    dtype = L.BF16
    assert L.lib().p2p_brig_ok(op, dtype, n, lh, lh, cg, cd) == 1
    rng = np.random.default_rng(23)
    hi, lo, w = make_case(rng, n, lh, cg, cd, 2, dtype)
    g_ref, p_ref, _ = oracle_ops(hi, lo, w, 2)
    hi_b, lo_b = U.halo_from(hi, dtype), U.halo_from(lo, dtype)
    wn = torch.empty(16 * cg * cd, dtype=U.tdt(dtype), device=U.DEV)
    wt = torch.empty(16 * cg * cd, dtype=U.tdt(dtype), device=U.DEV)
    w_d = U.dev(w.reshape(-1))
    L.call("p2p_weight_prep", dtype, U.ptr(w_d), cg, cd, U.ptr(wn), U.ptr(wt), U.stream())
    ref, shape = (g_ref, (n, lh, lh, cd)) if op == L.OP_G else (p_ref, (n, 2 * lh, 2 * lh, cg))
    ncols, res = shape[3], shape[1]
    slots = L.lib().p2p_igemm_layer_stat_slots(op, dtype, n, lh, lh, cg, cd)
    assert slots == L.lib().p2p_brig_stat_slots(op, dtype, n, lh, lh, cg, cd) and slots >= 1
    out = E.DenseBuf(*shape, U.tdt(dtype), U.DEV)
    out.t.fill_(float("nan"))
    spart = torch.full((n * slots * ncols * 2,), float("nan"), dtype=torch.float32, device=U.DEV)
    hv, lv = (hi_b.view(), out.view()) if op == L.OP_G else (out.view(), lo_b.view())
    L.call("p2p_igemm", op, dtype, n, lh, lh, cg, cd, C.byref(hv), C.byref(lv), U.ptr(wt if op == L.OP_G else wn), 1, None,
           U.ptr(spart), U.stream())
    got = U.dense_to_np(out)
    assert np.isfinite(got).all()
    assert U.rel_err(got, ref) < OUT_TOL[dtype]
    x = got.astype(np.float64)
    sp = spart.view(n, slots, ncols, 2).cpu().numpy().astype(np.float64)
    cnt = res * res / slots
    mean = sp[..., 0].mean(1)
    m2 = sp[..., 1].sum(1) + cnt * ((sp[..., 0] - mean[:, None]) ** 2).sum(1)
    np.testing.assert_allclose(mean, x.mean(axis=(1, 2)), rtol=1e-4, atol=1e-5 * np.abs(x).max())
    np.testing.assert_allclose(m2 / (res * res), x.var(axis=(1, 2)), rtol=5e-4)
    # without statistics, into a channel slice of a wider haloed buffer (concat-by-slice, networks.py:94)
    wide = E.HaloBuf(n, res, res, ncols + 64, dtype, U.DEV)
    hv, lv = (hi_b.view(), wide.view(coff=64)) if op == L.OP_G else (wide.view(coff=64), lo_b.view())
    L.call("p2p_igemm", op, dtype, n, lh, lh, cg, cd, C.byref(hv), C.byref(lv), U.ptr(wt if op == L.OP_G else wn), 1, None,
           None, U.stream())
    back = U.halo_to_np(wide)
    assert np.array_equal(back[..., 64:], got) and not back[..., :64].any()


@pytest.mark.parametrize("cbw", [1, 2])
@pytest.mark.parametrize("op,n,lh,cg,cd,act", [(L.OP_P, 5, 8, 128, 64, L.ACT_RELU), (L.OP_P, 2, 16, 64, 96, L.ACT_RELU),
                                               (L.OP_G, 3, 8, 64, 256, L.ACT_LEAKY), (L.OP_G, 2, 16, 32, 256, L.ACT_LEAKY)])
def test_fused_block_conv_instance_norm_activation(op, n, lh, cg, cd, act, cbw, monkeypatch):
    """One launch for the whole block of networks.py:7-21 / 24-36 (without dropout): the activated output, the raw
    convolution result and the (mean, rstd) statistics equal convolution + p2p_norm_act_fwd in two launches and the oracle."""
    dtype = L.BF16
    monkeypatch.setenv("P2P_BRIG_CBW", str(cbw))
    monkeypatch.setenv("P2P_BRIG_MIN_WG", "1")
    assert L.lib().p2p_igemm_norm_act_ok(op, dtype, n, lh, lh, cg, cd) == 1
    assert L.lib().p2p_igemm_norm_act_ok(L.OP_P, dtype, n, 32, 32, 64, 64) == 0        # strips of an image: not fusable
    rng = np.random.default_rng(29)
    hi, lo, w = make_case(rng, n, lh, cg, cd, 2, dtype)
    g_ref, p_ref, _ = oracle_ops(hi, lo, w, 2)
    hi_b, lo_b = U.halo_from(hi, dtype), U.halo_from(lo, dtype)
    wn = torch.empty(16 * cg * cd, dtype=U.tdt(dtype), device=U.DEV)
    wt = torch.empty(16 * cg * cd, dtype=U.tdt(dtype), device=U.DEV)
    w_d = U.dev(w.reshape(-1))
    L.call("p2p_weight_prep", dtype, U.ptr(w_d), cg, cd, U.ptr(wn), U.ptr(wt), U.stream())
    ref, shape = (g_ref, (n, lh, lh, cd)) if op == L.OP_G else (p_ref, (n, 2 * lh, 2 * lh, cg))
    ncols, res = shape[3], shape[1]
    gamma = (1 + 0.2 * rng.normal(size=ncols)).astype(np.float32)
    beta = (0.2 * rng.normal(size=ncols)).astype(np.float32)
    g_d, b_d = U.dev(gamma), U.dev(beta)
    raw = E.DenseBuf(*shape, U.tdt(dtype), U.DEV)
    raw.t.fill_(float("nan"))
    y = E.HaloBuf(n, res, res, ncols + 32, dtype, U.DEV)
    stats = torch.full((n, ncols, 2), float("nan"), dtype=torch.float32, device=U.DEV)
    hv, lv = (hi_b.view(), raw.view()) if op == L.OP_G else (raw.view(), lo_b.view())
    L.call("p2p_igemm_norm_act", op, dtype, n, lh, lh, cg, cd, C.byref(hv), C.byref(lv), U.ptr(wt if op == L.OP_G else wn),
           U.ptr(g_d), U.ptr(b_d), 1e-3, act, 0.3, C.byref(y.view(coff=32)), U.ptr(stats), U.stream())
    x = U.dense_to_np(raw).astype(np.float64)
    assert U.rel_err(x, ref) < OUT_TOL[dtype]
    mean, var = x.mean(axis=(1, 2)), x.var(axis=(1, 2))
    np.testing.assert_allclose(stats[..., 0].cpu().numpy(), mean, rtol=1e-4, atol=1e-5 * np.abs(x).max())
    np.testing.assert_allclose(stats[..., 1].cpu().numpy(), 1.0 / np.sqrt(var + 1e-3), rtol=3e-4)
    z = rg.instance_norm(torch.tensor(x), torch.tensor(gamma, dtype=F64), torch.tensor(beta, dtype=F64))
    want = (torch.relu(z) if act == L.ACT_RELU else rg.leaky_relu(z)).numpy()
    got = U.halo_to_np(y)
    assert U.rel_err(got[..., 32:], want) < OUT_TOL[dtype]
    assert not got[..., :32].any()           # the other slice of the concat buffer is untouched


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,lh,cg,cd,stride", [(3, 32, 4, 64, 2), (2, 32, 8, 64, 2), (2, 64, 36, 4, 1), (3, 32, 64, 1, 1),
                                                (2, 16, 1, 64, 2), (20, 64, 33, 8, 1), (3, 32, 32, 128, 2), (2, 16, 64, 128, 2),
                                                (2, 16, 64, 256, 2), (2, 32, 64, 64, 2), (2, 128, 36, 4, 1), (2, 128, 64, 1, 1)])
def test_wgrad_small_lds_resident(dtype, n, lh, cg, cd, stride):
    """Edge-layer weight gradients through the LDS-resident kernel (all 16 taps out of one staged strip)."""
    rng = np.random.default_rng(18)
    hi, lo, w = make_case(rng, n, lh, cg, cd, stride, dtype)
    _, _, w_ref = oracle_ops(hi, lo, w, stride)
    hi_b, lo_b = _pad_view_input(hi, E.pad8(cg), dtype), _pad_view_input(lo, E.pad8(cd), dtype)
    nb = L.lib().p2p_wgrad_small_blocks(dtype, stride, n, lh, lh, cg, cd, E.pad8(cg), E.pad8(cd))
    assert nb > 0
    ws = torch.full((nb * 16 * cg * cd,), float("nan"), dtype=torch.float32, device=U.DEV)
    dw = torch.full((16 * cg * cd,), float("nan"), dtype=torch.float32, device=U.DEV)
    L.call("p2p_wgrad_small", dtype, stride, n, lh, lh, cg, cd, C.byref(hi_b.view()), C.byref(lo_b.view()), U.ptr(dw), U.ptr(ws),
           U.stream())
    assert U.rel_err(dw.cpu().numpy().reshape(4, 4, cg, cd), w_ref) < 2e-5


def test_conv_fewin_actbwd_equals_conv_then_act_bwd():
    """d(D.last)/d(features) with the LeakyReLU backward of D.down fused into its epilogue (p2p_conv_fewin_actbwd) against the
    two launches it replaces (p2p_conv_fewin, p2p_act_bwd): bit for bit, into a channel-sliced haloed output view"""
    dtype, n, lh, cg, cd = L.BF16, 3, 32, 64, 1
    rng = np.random.default_rng(23)
    lo = rng.normal(size=(n, lh, lh, cd)).astype(np.float32)
    w = (0.05 * rng.normal(size=(16, cg, cd))).astype(np.float32)
    gate = rng.normal(size=(n, lh, lh, cg)).astype(np.float32)
    lo_pad = E.pad8(cd)
    lo_b = _pad_view_input(lo, lo_pad, dtype)
    wn = torch.zeros(16 * E.up32(cg) * lo_pad, dtype=U.tdt(dtype), device=U.DEV)
    wt = torch.zeros(16 * E.up32(cd) * E.pad8(cg), dtype=U.tdt(dtype), device=U.DEV)
    L.call("p2p_weight_prep_pad", dtype, U.ptr(U.dev(w.reshape(-1))), cg, cd, U.ptr(wn), E.up32(cg), lo_pad, U.ptr(wt), E.up32(cd),
           E.pad8(cg), U.stream())
    assert L.lib().p2p_conv_fewin_ok(L.OP_P, 1, dtype, n, lh, lh, lo_pad, cg)
    gate_b = U.halo_from(gate, dtype)
    g_mid = E.HaloBuf(n, lh, lh, cg, dtype, U.DEV)
    two = E.HaloBuf(n, lh, lh, cg + 8, dtype, U.DEV)
    one = E.HaloBuf(n, lh, lh, cg + 8, dtype, U.DEV)
    L.call("p2p_conv_fewin", L.OP_P, 1, dtype, n, lh, lh, lo_pad, cg, E.up32(cg), C.byref(lo_b.view()), C.byref(g_mid.view()),
           U.ptr(wn), None, L.ACT_NONE, 0.0, U.stream())
    gs = L.GSrc(g_mid.view().ptr, 1, 1, 0, cg, 0)
    # the gradient source indexes dense pixels: use a dense copy of the haloed intermediate
    dense = g_mid.t[:, E.HALO:E.HALO + lh, E.HALO:E.HALO + lh, :].contiguous()
    gs = L.GSrc(dense.data_ptr(), 1, 1, 0, cg, 0)
    L.call("p2p_act_bwd", dtype, n, lh, lh, cg, C.byref(gate_b.view()), C.byref(gs), None, 0.3, C.byref(two.view(coff=8)), U.stream())
    L.call("p2p_conv_fewin_actbwd", L.OP_P, 1, dtype, n, lh, lh, lo_pad, cg, E.up32(cg), C.byref(lo_b.view()),
           C.byref(one.view(coff=8)), U.ptr(wn), C.byref(gate_b.view()), 0.3, U.stream())
    torch.cuda.synchronize()
    assert torch.equal(one.t, two.t)
    assert float(one.t.float().abs().max()) > 0 and np.count_nonzero(U.halo_to_np(one)[..., :8]) == 0
    with pytest.raises(RuntimeError):      # a gate view that does not hold whole 16-byte channel runs is refused
        bad = L.Tensor(gate_b.view().ptr, gate_b.view().img_stride, gate_b.view().row_stride, 60)
        L.call("p2p_conv_fewin_actbwd", L.OP_P, 1, dtype, n, lh, lh, lo_pad, cg, E.up32(cg), C.byref(lo_b.view()),
               C.byref(one.view(coff=8)), U.ptr(wn), C.byref(bad), 0.3, U.stream())
