import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import reference_graph as rg
from palette_and_histo_gan_amd import _lib as L, engine as E
from tests.test_train_step_gpu import setup_case, to_np
F64 = torch.float64
B, S = 2, 64
rng, Gp, Dp, src, tgt, masks = setup_case(B, S, 21)
ref = rg.train_step_rgba(Gp, Dp, torch.tensor(src, dtype=F64), torch.tensor(tgt, dtype=F64), [torch.tensor(m, dtype=F64) for m in masks], lambda_l1=100.0)
for dtype in (L.F32, L.BF16):
    for mf in (False, True):
        eng = E.Pix2PixEngine(4, 4, "tanh", S, dtype, use_mfma=mf)
        eng.set_params(to_np(Gp), to_np(Dp))
        out = eng.train_step_rgba(src, tgt, 100.0, masks=masks, apply_update=False).cpu().numpy()
        print("=== dtype", dtype, "mfma", mf, out)
        gg = eng.G.export(eng.G.grads)
        for k, r in ref["g_grads"].items():
            r = r.numpy(); g = gg[k]
            sc = np.abs(r).max()
            err = np.abs(g - r).max() / sc if sc > 0 else np.abs(g).max()
            nbad = int((np.abs(g - r) > 0.05 * sc).sum()) if sc > 0 else 0
            print(f"  {k:14s} scale {sc:9.3e} maxrel {err:9.3e} nbad {nbad}/{r.size} nan {int(np.isnan(g).sum())}")
