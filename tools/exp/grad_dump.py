#!/usr/bin/env python3
"""Dumps the generator / discriminator gradients of one train step (no update) to an .npz: A/B of kernel switches across processes."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from palette_and_histo_gan_amd import _lib as L, engine as E, dataset_utils as DU
out, B, dt = sys.argv[1], int(sys.argv[2]), (L.F32 if sys.argv[3] == "f32" else L.BF16)
eng = E.Pix2PixEngine(4, 4, "tanh", 64, dt, device="cuda:0", seed=47)
src, tgt = DU.synthetic_rgba_batch(np.random.default_rng([47, 0]), B, 64, palette_size=None)
rng = np.random.default_rng(3)
masks = [rng.integers(0, 2, size=(B * r * r, f)).astype(np.uint8) for r, f in ((2, 512), (4, 512), (8, 256))]
losses = eng.train_step_rgba(src, tgt, 100.0, masks=masks, apply_update=False)
torch.cuda.synchronize()
g = {k: eng.G.g_tensor(k).cpu().numpy() if hasattr(eng.G, "g_tensor") else None for k in []}
np.savez(out, G=eng.G.grads.cpu().numpy(), D=eng.D.grads.cpu().numpy(), losses=losses.cpu().numpy(),
         names=np.array(list(eng.G.offsets.keys())), offs=np.array([eng.G.offsets[k] for k in eng.G.offsets]))
