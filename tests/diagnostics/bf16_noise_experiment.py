#!/usr/bin/env python3
"""Where does the bf16 throughput mode's gradient error come from?  CPU only, oracle only: the f64 oracle is re-run with
bf16 rounding applied at chosen subsets of its storage points (weights / activations by map size) and every variant's
generator gradients are compared with the unrounded f64 run (relative L2 per tensor).  Result:
rounding ONLY the weights already moves the deep layers' gradients by 15-20 %, i.e. the error is a property of bf16
storage on this network (InstanceNorm over tiny, highly correlated maps amplifies rounding), not of the kernels.
    python tests/diagnostics/bf16_noise_experiment.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import reference_graph as rg
torch.set_num_threads(8)
F64 = torch.float64
B, S = 2, 64
rng = np.random.default_rng(21)
Gp = rg.perturb_affine(rg.init_params(rg.generator_param_shapes(4, 4), rng, F64), rng)
Dp = rg.perturb_affine(rg.init_params(rg.discriminator_param_shapes(4), rng, F64), rng)
src, tgt = rg.synthetic_rgba_batch(rng, B, S)
masks = [rng.integers(0, 2, size=s).astype(np.uint8) for s in rg.dropout_mask_shapes(B, S)]
tm = [torch.tensor(m, dtype=F64) for m in masks]
args = (Gp, Dp, torch.tensor(src, dtype=F64), torch.tensor(tgt, dtype=F64), tm)
ref = rg.train_step_rgba(*args, lambda_l1=100.0)

orig_q = rg._q
def run(filt, label):
    def q(x):
        if rg._QUANT is None or not filt(x): return x
        return x + (x.detach().to(rg._QUANT).to(x.dtype) - x.detach())
    rg._q = q
    with rg.storage_dtype(torch.bfloat16):
        out = rg.train_step_rgba(*args, lambda_l1=100.0)
    rg._q = orig_q
    errs = {}
    for k, r in ref["g_grads"].items():
        r = r.numpy(); g = out["g_grads"][k].numpy()
        errs[k] = np.linalg.norm(g - r) / (np.linalg.norm(r) + 1e-300)
    sel = ["down1.kernel", "down3.kernel", "down5.kernel", "up1.kernel", "up3.kernel", "up5.kernel", "up6.kernel", "last.kernel"]
    print(f"{label:50s}", " ".join(f"{errs[k]:.3f}" for k in sel), " loss rel", abs(out["g_loss"][0] - ref["g_loss"][0]) / ref["g_loss"][0])

is_w = lambda x: x.dim() == 4 and x.shape[0] == 4 and x.shape[1] == 4
run(lambda x: True, "all rounding points (current)")
run(lambda x: is_w(x), "weights only")
run(lambda x: not is_w(x), "activations/raw only")
run(lambda x: not is_w(x) and x.shape[1] > 8, "acts on maps > 8x8 only")
run(lambda x: not is_w(x) and x.shape[1] > 2, "acts on maps > 2x2 only")
run(lambda x: is_w(x) or x.shape[1] > 8, "weights + maps > 8x8")
run(lambda x: is_w(x) or x.shape[1] > 16, "weights + maps > 16x16")
run(lambda x: is_w(x) or x.shape[1] > 32, "weights + maps > 32x32")
