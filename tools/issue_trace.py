#!/usr/bin/env python3
"""Where does the HOST's time go while it issues train steps?  Per-step issue times (no synchronisation inside the loop), the
drain time afterwards and the wall time per step, for the replayed step (one p2p_replay call) and the eager one (Python/ctypes
per launch).  A step whose issue time jumps to the device's step time is a step on which the runtime made the host wait (queue
or kernel-argument pool full); a wall time above the device time with short issue times is a device that was not fed.

  python tools/issue_trace.py [config] [steps]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as BN  # noqa: E402
from palette_and_histo_gan_amd import _lib as L, engine as E, dataset_utils as DU  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
model, B, S, lam_l1, lam_hist, palette = BN.CONFIGS[cfg]
indexed = model == "indexed"
if indexed:
    eng = E.Pix2PixEngine(1, 256, "softmax", S, L.BF16, device="cuda:0", seed=47)
    src, tgt, _ = DU.synthetic_indexed_batch(np.random.default_rng([47, 0]), B, S, palette)
else:
    eng = E.Pix2PixEngine(4, 4, "tanh", S, L.BF16, device="cuda:0", seed=47)
    src, tgt = BN.synthetic_batch(0, B, S, palette)
src, tgt = torch.as_tensor(src).cuda(), torch.as_tensor(tgt).cuda()


def step():
    if indexed:
        return eng.train_step_indexed(src, tgt, lam_l1)
    return eng.train_step_rgba(src, tgt, lam_l1, lam_hist)


for mode in ("replay", "eager", "replay"):
    eng.replay_enabled = mode == "replay"
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ts = [time.perf_counter()]
    for _ in range(n):
        step()
        ts.append(time.perf_counter())
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    d = np.diff(np.array(ts)) * 1e3
    print(f"{cfg} {mode:6s}: issue mean {d.mean():.3f} ms/step (median {np.median(d):.3f}, max {d.max():.3f}), drain {1e3 * (t_end - ts[-1]):.2f} ms, "
          f"wall {1e3 * (t_end - ts[0]) / n:.3f} ms/step")
    print("   per step:", " ".join(f"{x:.2f}" for x in d))
