#!/usr/bin/env python3
"""Per-tensor gradient error of the bf16 throughput mode against the oracle run with the same bf16 storage points
(tests/test_train_step_gpu.py run_case), printed per layer: where the bf16 backward path loses accuracy."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import reference_graph as rg                       # noqa: E402
from palette_and_histo_gan_amd import _lib as L               # noqa: E402
from palette_and_histo_gan_amd import engine as E             # noqa: E402
from tests.test_train_step_gpu import setup_case, to_np       # noqa: E402

F64 = torch.float64
B, S = 2, 64
rng, Gp, Dp, src, tgt, masks = setup_case(B, S, 21)
tm = [torch.tensor(m, dtype=F64) for m in masks]
with rg.storage_dtype(torch.bfloat16):
    ref = rg.train_step_rgba(Gp, Dp, torch.tensor(src, dtype=F64), torch.tensor(tgt, dtype=F64), tm, lambda_l1=100.0)
ref64 = rg.train_step_rgba(Gp, Dp, torch.tensor(src, dtype=F64), torch.tensor(tgt, dtype=F64), tm, lambda_l1=100.0)
eng = E.Pix2PixEngine(4, 4, "tanh", S, L.BF16)
eng.set_params(to_np(Gp), to_np(Dp))
out = eng.train_step_rgba(src, tgt, 100.0, masks=masks, apply_update=False).cpu().numpy()
print("losses", out)
print("want  ", [ref["g_loss"][i] for i in range(3)], [ref["d_loss"][i] for i in range(3)])
got = eng.G.export(eng.G.grads)
print(f"{'tensor':16s} {'L2 vs bf16-oracle':>18s} {'L2 vs f64 oracle':>18s} {'oracle bf16 vs f64':>18s}")
for k, r in ref["g_grads"].items():
    r = r.numpy().astype(np.float64)
    r64 = ref64["g_grads"][k].numpy().astype(np.float64)
    g = got[k].astype(np.float64)
    n = np.linalg.norm(r) + 1e-300
    print(f"{k:16s} {np.linalg.norm(g - r) / n:18.4f} {np.linalg.norm(g - r64) / (np.linalg.norm(r64) + 1e-300):18.4f} {np.linalg.norm(r - r64) / (np.linalg.norm(r64) + 1e-300):18.4f}")
