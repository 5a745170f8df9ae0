#!/usr/bin/env python3
"""How long does the host take to ISSUE one train step (ctypes launches, no synchronisation) compared with the device
time of the step?  Prints both; the step is launch-bound if they are close."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from palette_and_histo_gan_amd import _lib as L, engine as E  # noqa: E402

B, S = 256, 64
eng = E.Pix2PixEngine(4, 4, "tanh", S, L.BF16, device="cuda:0", seed=47)
rng = np.random.default_rng(0)
src = torch.as_tensor(rng.uniform(-1, 1, size=(B, S, S, 4)).astype(np.float32)).cuda()
tgt = torch.as_tensor(rng.uniform(-1, 1, size=(B, S, S, 4)).astype(np.float32)).cuda()
for _ in range(5):
    eng.train_step_rgba(src, tgt, 100.0)
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    eng.train_step_rgba(src, tgt, 100.0)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue {1e3 * (t1 - t0) / n:.3f} ms/step, wall {1e3 * (t2 - t0) / n:.3f} ms/step")
