#!/usr/bin/env python3
"""Experiment: run the c2 step with the MAIN stream at high HIP priority and the weight-gradient / histogram side streams at the
default (lower) priority.  usage: python tools/exp/prio_bench.py [c2] [main_prio] [steps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as BN
from palette_and_histo_gan_amd import _lib as L, engine as E

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
prio = int(sys.argv[2]) if len(sys.argv) > 2 else -1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
model, B, S, lam_l1, lam_hist, palette = BN.CONFIGS[cfg]
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
eng = E.Pix2PixEngine(4, 4, "tanh", S, L.BF16, device="cuda:0", seed=47)
src, tgt = BN.synthetic_batch(0, B, S, palette)
src_d, tgt_d = torch.as_tensor(src).cuda(), torch.as_tensor(tgt).cuda()
main = torch.cuda.Stream(priority=prio) if prio != 0 else torch.cuda.current_stream()
torch.cuda.synchronize()
with torch.cuda.stream(main):
    for _ in range(10):
        eng.train_step_rgba(src_d, tgt_d, lam_l1, lam_hist)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.train_step_rgba(src_d, tgt_d, lam_l1, lam_hist)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
print(f"{cfg} main priority {prio}: {B * steps / el:.0f} img/s, {el / steps * 1e3:.4f} ms/step")
