#!/usr/bin/env python3
"""Diagnostic (wrong numerics, timing only): how long is the c2 step when a whole class of launches is left out?
  python tools/exp/skip_diag.py  ->  one line per variant.  Tells what a perfect optimisation of that class could give."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as BN
from palette_and_histo_gan_amd import _lib as L, engine as E

model, B, S, lam_l1, lam_hist, palette = BN.CONFIGS["c2"]
src, tgt = BN.synthetic_batch(0, B, S, palette)
src_d, tgt_d = torch.as_tensor(src).cuda(), torch.as_tensor(tgt).cuda()


def run(tag, patch):
    eng = E.Pix2PixEngine(4, 4, "tanh", S, L.BF16, device="cuda:0", seed=47)
    patch(eng)
    for _ in range(10):
        eng.train_step_rgba(src_d, tgt_d, lam_l1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(60):
        eng.train_step_rgba(src_d, tgt_d, lam_l1)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 60
    print(f"{tag:40s} {el * 1e3:.4f} ms/step", flush=True)


def skip_names(names):
    def patch(eng):
        orig = L.call
        def call(name, *a):
            if name in names:
                return
            orig(name, *a)
        L.call = call
        E._ORIG_CALL = call          # keep the replay guard quiet (B = 256 is above the replay limit anyway)
    return patch


WG = {"p2p_wgrad_small", "p2p_wgemm", "p2p_wgemm_edge", "p2p_view_colsum", "p2p_colsum_batched"}
run("everything", lambda e: None)
run("no weight gradients (side stream)", skip_names(WG))
run("no Adam, no weight-copy refresh", skip_names({"p2p_adam_flat_dev", "p2p_weight_prep_batched", "p2p_adam_prep_batched"}))
run("no InstanceNorm backward", skip_names({"p2p_norm_act_bwd"}))
run("no pack_pair", skip_names({"p2p_pack_pair"}))
run("no wgrad, no Adam/prep", skip_names(WG | {"p2p_adam_flat_dev", "p2p_weight_prep_batched"}))
