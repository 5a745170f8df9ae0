#!/usr/bin/env python3
"""c2 step time with the weight-gradient side stream on / off (same process, alternating), ms per step."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from palette_and_histo_gan_amd import _lib as L, engine as E, dataset_utils as DU
B = 256
src, tgt = DU.synthetic_rgba_batch(np.random.default_rng([47, 0]), B, 64, palette_size=None)
src_d, tgt_d = torch.as_tensor(src).cuda(), torch.as_tensor(tgt).cuda()
engs = {ov: E.Pix2PixEngine(4, 4, "tanh", 64, L.BF16, device="cuda:0", seed=47, overlap_wgrad=ov) for ov in (True, False)}
def run(eng, n):
    for _ in range(5): eng.train_step_rgba(src_d, tgt_d, 100.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): eng.train_step_rgba(src_d, tgt_d, 100.0)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    for ov in (True, False):
        print("overlap", ov, "ms/step %.4f" % run(engs[ov], 40), flush=True)
