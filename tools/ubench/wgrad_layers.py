#!/usr/bin/env python3
"""Times p2p_wgrad_small on the wide weight-gradient shapes of the c2 step (B = 256, S = 64, bf16), one launch shape at a
time: hipEvents around `reps` back-to-back launches.  P2P_WS_ABL=1/2 times the staging / the contraction alone.
Not part of the product."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from palette_and_histo_gan_amd import _lib as L          # noqa: E402
from palette_and_histo_gan_amd import engine as E        # noqa: E402

B0 = int(os.environ.get("UB_BATCH", "256"))
# (name, stride, lh, cg, cd, batch multiplier, hi pixel channels, lo pixel channels)
SHAPES = [("up6", 2, 32, 32, 128, 1, 32, 128), ("up5", 2, 16, 64, 256, 1, 64, 256), ("down2", 2, 16, 64, 128, 1, 64, 128),
          ("last", 1, 64, 36, 4, 1, 40, 8), ("D.last", 1, 32, 64, 1, 2, 64, 8), ("D.down", 2, 32, 8, 64, 2, 8, 64),
          ("down1", 2, 32, 4, 64, 1, 8, 64)]
if os.environ.get("UB_ONLY"):
    SHAPES = [s for s in SHAPES if s[0] in os.environ["UB_ONLY"].split(",")]


def main():
    reps = int(os.environ.get("UB_REPS", "20"))
    dev = "cuda:0"
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device=dev).manual_seed(1)
    for name, stride, lh, cg, cd, bm, hic, loc in SHAPES:
        B = B0 * bm
        hi = E.HaloBuf(B, stride * lh, stride * lh, hic, L.BF16, dev)
        lo = E.HaloBuf(B, lh, lh, loc, L.BF16, dev)
        for hb, creal in ((hi, cg), (lo, cd)):
            hb.t[:, 2:-2, 2:-2, :creal] = torch.randn((B, hb.h, hb.w, creal), device=dev, generator=g).to(torch.bfloat16)
        hv, lv = hi.view(), lo.view()
        nb = L.lib().p2p_wgrad_small_blocks(L.BF16, stride, B, lh, lh, cg, cd, hv.ld, lv.ld)
        ws = torch.empty(nb * 16 * cg * cd, dtype=torch.float32, device=dev)
        dw = torch.empty(16 * cg * cd, dtype=torch.float32, device=dev)

        def launch():
            L.call("p2p_wgrad_small", L.BF16, stride, B, lh, lh, cg, cd, C.byref(hv), C.byref(lv), C.c_void_p(dw.data_ptr()),
                   C.c_void_p(ws.data_ptr()), st)
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            launch()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / reps * 1e3
        fl = 2.0 * B * lh * lh * 16 * cg * cd
        mb = (B * (stride * lh) ** 2 * hic + B * lh * lh * loc) * 2 / 1e6        # algorithmic bytes: both operands once
        print(f"{name:6s} s={stride} lo={lh:2d} cg={cg:4d} cd={cd:4d} slabs={nb:4d} {us:7.1f} us {fl / us * 1e-6:7.1f} TFLOP/s "
              f"{mb / us:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
