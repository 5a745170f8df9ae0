#!/usr/bin/env python3
"""Times p2p_wgemm on the deep weight-gradient shapes of the c2 step (B = 256, S = 64, bf16), one launch shape at a time, with the
pixel split the engine would choose.  UB_COLD=1: a 1 GiB fill between launches (the step's condition), each launch timed on its own.
Not part of the product."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from palette_and_histo_gan_amd import _lib as L          # noqa: E402
from palette_and_histo_gan_amd import engine as E        # noqa: E402

B = int(os.environ.get("UB_BATCH", "256"))
COLD = os.environ.get("UB_COLD", "0") != "0"
SHAPES = [("up3", 4, 256, 1024), ("up4", 8, 128, 512), ("down3", 8, 128, 256), ("down4", 4, 256, 512), ("up2", 2, 512, 1024),
          ("down5", 2, 512, 512), ("up1", 1, 512, 512)]


def main():
    reps = int(os.environ.get("UB_REPS", "20"))
    dev = "cuda:0"
    eng = E.Pix2PixEngine(4, 4, "tanh", 64, L.BF16, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device=dev).manual_seed(1)
    flush = torch.empty(256 << 20, dtype=torch.float32, device=dev) if COLD else None
    tot = 0.0
    for name, lh, cg, cd in SHAPES:
        hi = E.HaloBuf(B, 2 * lh, 2 * lh, cg, L.BF16, dev)
        lo = E.HaloBuf(B, lh, lh, cd, L.BF16, dev)
        for hb in (hi, lo):
            hb.t[:, 2:-2, 2:-2, :] = torch.randn((B, hb.h, hb.w, hb.c), device=dev, generator=g).to(torch.bfloat16)
        ms = eng._msplit(B, lh, cg, cd)
        ws = torch.empty(max(ms, 1) * 16 * cg * cd, dtype=torch.float32, device=dev)
        dw = torch.empty(16 * cg * cd, dtype=torch.float32, device=dev)
        hv, lv = hi.view(), lo.view()

        def launch():
            L.call("p2p_wgemm", L.BF16, B, lh, lh, cg, cd, C.byref(hv), C.byref(lv), C.c_void_p(dw.data_ptr()), ms,
                   C.c_void_p(ws.data_ptr()) if ms > 1 else None, st)
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        if COLD:
            evs = []
            for r in range(reps):
                flush.fill_(float(r))
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                launch()
                b.record()
                evs.append((a, b))
            torch.cuda.synchronize()
            us = sorted(x.elapsed_time(y) for x, y in evs)[len(evs) // 2] * 1e3
        else:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                launch()
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) / reps * 1e3
        fl = 2.0 * B * lh * lh * 16 * cg * cd * (0.25 if lh == 1 else 1.0)
        tot += us
        print(f"{name:6s} lo={lh:2d} cg={cg:4d} cd={cd:4d} msplit={ms} {us:7.1f} us {fl / us * 1e-6:7.1f} TFLOP/s", flush=True)
    print(f"total {tot:.1f} us")


if __name__ == "__main__":
    main()
