#!/usr/bin/env python3
"""Per-tensor gradient error of the f32 parity case of tests/test_train_step_gpu.py (run_case) against the f64 oracle: which layer
a kernel switch (environment variables, one process per setting) moves.  Diagnostic only."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import test_train_step_gpu as T      # noqa: E402
from palette_and_histo_gan_amd import _lib as L, engine as E      # noqa: E402
from oracle import reference_graph as rg      # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rng, Gp, Dp, src, tgt, masks = T.setup_case(B, 64, 21)
tm = [torch.tensor(m, dtype=torch.float64) for m in masks]
ref = rg.train_step_rgba(Gp, Dp, torch.tensor(src, dtype=torch.float64), torch.tensor(tgt, dtype=torch.float64), tm, lambda_l1=100.0)
eng = E.Pix2PixEngine(4, 4, "tanh", 64, L.F32, use_mfma=True)
eng.set_params(T.to_np(Gp), T.to_np(Dp))
out = eng.train_step_rgba(src, tgt, 100.0, masks=masks, apply_update=False).cpu().numpy()
got = eng.G.export(eng.G.grads)
for k, r in ref["g_grads"].items():
    r = r.numpy().astype(np.float64)
    g = got[k].astype(np.float64)
    e = np.abs(g - r).max() / max(np.abs(r).max(), 1e-30)
    if e > 2e-6:
        print(f"{k:16s} max-norm err {e:.3e}")
print("losses", out)
