#!/usr/bin/env python3
"""Runs the same no-update train step several times in one process and reports, per gradient tensor, whether the results are
bit-identical between runs (a race shows up as run-to-run differences)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from palette_and_histo_gan_amd import _lib as L, engine as E, dataset_utils as DU
B, dt, reps = int(sys.argv[1]), (L.F32 if sys.argv[2] == "f32" else L.BF16), int(sys.argv[3])
eng = E.Pix2PixEngine(4, 4, "tanh", 64, dt, device="cuda:0", seed=47)
src, tgt = DU.synthetic_rgba_batch(np.random.default_rng([47, 0]), B, 64, palette_size=None)
rng = np.random.default_rng(3)
masks = [rng.integers(0, 2, size=(B * r * r, f)).astype(np.uint8) for r, f in ((2, 512), (4, 512), (8, 256))]
ref = None
names = list(eng.G.offsets.keys()); offs = [eng.G.offsets[k] for k in names]
order = np.argsort(offs)
bad = {}
for r in range(reps):
    eng.train_step_rgba(src, tgt, 100.0, masks=masks, apply_update=False)
    torch.cuda.synchronize()
    g = eng.G.grads.cpu().numpy().copy()
    if ref is None:
        ref = g
        continue
    if not np.array_equal(g, ref):
        for i, o in enumerate(order):
            lo = offs[o]; hi = offs[order[i + 1]] if i + 1 < len(order) else len(g)
            if not np.array_equal(g[lo:hi], ref[lo:hi]):
                d = np.abs(g[lo:hi] - ref[lo:hi]).max() / (np.abs(ref[lo:hi]).max() + 1e-30)
                bad[names[o]] = max(bad.get(names[o], 0), d)
print("B", B, sys.argv[2], "runs", reps, "tensors that differ between runs:", {k: f"{v:.2e}" for k, v in bad.items()} or "none")
