#!/usr/bin/env python3
"""Device time of each of the first steps after a synchronize (the driver's window is 20 steps behind 5 warm-up steps): one event per
step boundary on the launch stream.  Diagnostic: where the 0.7 % between a 20-step and a 200-step window comes from."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from palette_and_histo_gan_amd import _lib as L, engine as E, dataset_utils as DU

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng = E.Pix2PixEngine(4, 4, "tanh", 64, L.BF16, device="cuda:0", seed=47)
src, tgt = DU.synthetic_rgba_batch(np.random.default_rng([47, 0]), B, 64, palette_size=None)
s_d, t_d = torch.as_tensor(src).cuda(), torch.as_tensor(tgt).cuda()
if len(sys.argv) > 2 and sys.argv[2] == "burn":        # 0.3 s of dense bf16 matmuls first: is the ramp the chip's (clocks) or the step's?
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(60):
        a @ a
    t1.record(); torch.cuda.synchronize()
    print("burn %.0f ms" % t0.elapsed_time(t1))
for rep in range(3):
    for _ in range(5):
        eng.train_step_rgba(s_d, t_d, 100.0)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    evs[0].record()
    for i in range(40):
        eng.train_step_rgba(s_d, t_d, 100.0)
        evs[i + 1].record()
    torch.cuda.synchronize()
    ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(40)]
    print("rep", rep, "steps 1-20 mean %.4f  steps 21-40 mean %.4f  first five:" % (np.mean(ms[:20]), np.mean(ms[20:])), " ".join("%.3f" % m for m in ms[:5]),
          " last five:", " ".join("%.3f" % m for m in ms[-5:]))
