"""p2p_igemm in f32 on inputs with a wide dynamic range (per-pixel scales 10^U(-6, 2)): element-wise error against float64,
relative to sum |terms| (the bound any summation order obeys)."""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from palette_and_histo_gan_amd import _lib as L, engine as E
from tests import gpu_util as U
from tests.test_kernels_gpu import oracle_ops
rng = np.random.default_rng(5)
n, lh, cg, cd = 2, 8, 128, 128
for sk in (1, 4):
    hi = rng.normal(size=(n, 2 * lh, 2 * lh, cg)) * 10.0 ** rng.uniform(-6, 2, size=(n, 2 * lh, 2 * lh, 1))
    lo = rng.normal(size=(n, lh, lh, cd)) * 10.0 ** rng.uniform(-6, 2, size=(n, lh, lh, 1))
    w = rng.normal(scale=0.05, size=(4, 4, cg, cd))
    hi, lo, w = hi.astype(np.float32), lo.astype(np.float32), w.astype(np.float32)
    g_ref, p_ref, _ = oracle_ops(hi, lo, w, 2)
    g_abs, p_abs, _ = oracle_ops(np.abs(hi), np.abs(lo), np.abs(w), 2)
    hi_b, lo_b = U.halo_from(hi, L.F32), U.halo_from(lo, L.F32)
    wn = torch.empty(16 * cg * cd, dtype=torch.float32, device=U.DEV); wt = torch.empty_like(wn)
    w_d = U.dev(w.reshape(-1))
    L.call("p2p_weight_prep", L.F32, U.ptr(w_d), cg, cd, U.ptr(wn), U.ptr(wt), U.stream())
    for op, ref, bound, shape in ((L.OP_G, g_ref, g_abs, (n, lh, lh, cd)), (L.OP_P, p_ref, p_abs, (n, 2 * lh, 2 * lh, cg))):
        k = sk if op == L.OP_G else min(sk, 4)
        out = E.DenseBuf(*shape, torch.float32, U.DEV)
        slabs = torch.zeros((k * int(np.prod(shape)),), dtype=torch.float32, device=U.DEV)
        hv, lv = (hi_b.view(), out.view()) if op == L.OP_G else (out.view(), lo_b.view())
        L.call("p2p_igemm", op, L.F32, n, lh, lh, cg, cd, C.byref(hv), C.byref(lv), U.ptr(wt if op == L.OP_G else wn), k,
               U.ptr(slabs) if k > 1 else None, None, U.stream())
        got = U.dense_to_np(out) if k == 1 else slabs.view(k, *shape).sum(0).cpu().numpy()
        err = np.abs(got.astype(np.float64) - ref) / (bound + 1e-300)
        print("op", "GP"[op], "sk", k, "max err / sum|terms| =", err.max(), " 99.9th pct", np.quantile(err, 0.999))
