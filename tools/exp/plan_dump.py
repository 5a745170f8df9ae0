#!/usr/bin/env python3
"""Dumps every buffer of the step plan (activations with their halos, raw conv outputs, statistics, gradients) after one f32 train
step of the parity case (tests/test_train_step_gpu.py run_case): A/B of kernel switches across processes, `plan_dump.py cmp a b`
lists the buffers that differ.  Diagnostic only."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        x, y = a[k].astype(np.float64), b[k].astype(np.float64)
        if x.shape != y.shape:
            print(k, "shape", x.shape, y.shape); continue
        d = np.abs(x - y)
        if d.max() > 0:
            idx = np.unravel_index(np.argmax(d), d.shape)
            print(f"{k:24s} shape {x.shape} max diff {d.max():.3e} at {idx} (|x| max {np.abs(x).max():.3e}), differing {int((d > 0).sum())}")
    sys.exit(0)

if sys.argv[1] == "check_up6":
    # d(raw up6) recomputed in f64 from the dumped inputs of p2p_norm_act_bwd (closed form)
    a = np.load(sys.argv[2])
    g = a["P.gc.6"].astype(np.float64).reshape(2, 4096, 32)
    x = a["P.ru.6"].astype(np.float64).reshape(2, 4096, 32)
    st = a["P.su.6"].astype(np.float64)
    ga, be = a["up6.gamma"].astype(np.float64), a["up6.beta"].astype(np.float64)
    xh = (x - st[:, None, :, 0]) * st[:, None, :, 1]
    act = xh * ga + be
    d = g * (act > 0)
    m1, m2 = d.mean(1, keepdims=True), (d * xh).mean(1, keepdims=True)
    want = ga * st[:, None, :, 1] * (d - m1 - xh * m2)
    got = a["P.du.6"][:2 * 68 * 68 * 32].reshape(2, 68, 68, 32)[:, 2:66, 2:66, :].reshape(2, 4096, 32).astype(np.float64)
    print("du.6 vs closed form from its own inputs: max diff", np.abs(got - want).max(), "max |want|", np.abs(want).max())
    sys.exit(0)

from tests import test_train_step_gpu as T      # noqa: E402
from palette_and_histo_gan_amd import _lib as L, engine as E      # noqa: E402
B = 2
rng, Gp, Dp, src, tgt, masks = T.setup_case(B, 64, 21)
eng = E.Pix2PixEngine(4, 4, "tanh", 64, L.F32, use_mfma=True)
eng.set_params(T.to_np(Gp), T.to_np(Dp))
eng.train_step_rgba(src, tgt, 100.0, masks=masks, apply_update=False)
torch.cuda.synchronize()
P = eng.plan(B)
out = {}
def walk(prefix, v):
    if isinstance(v, torch.Tensor):
        out[prefix] = v.detach().float().cpu().numpy()
    elif hasattr(v, "t") and isinstance(getattr(v, "t"), torch.Tensor):
        walk(prefix, v._flat if hasattr(v, "_flat") else v.t)
    elif isinstance(v, dict):
        for k, x in v.items():
            walk(f"{prefix}.{k}", x)
    elif isinstance(v, (list, tuple)):
        for i, x in enumerate(v):
            walk(f"{prefix}.{i}", x)
walk("P", P)
out["G.grads"] = eng.G.grads.cpu().numpy()
exp = eng.G.export(eng.G.params)
out["up6.gamma"], out["up6.beta"] = exp["up6.gamma"], exp["up6.beta"]
np.savez(sys.argv[1], **out)
print(len(out), "buffers")
