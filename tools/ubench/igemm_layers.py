#!/usr/bin/env python3
"""Times p2p_igemm on the stride-2 block shapes of the c2 step (B = 256, S = 64, bf16), one launch shape at a time:
hipEvents around `reps` back-to-back launches on the current stream.  P2P_LIB selects a diagnostic build of the
library (tools/ubench/build_abl.sh), P2P_BRIG=0 the im2col kernel for every shape.  Not part of the product."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from palette_and_histo_gan_amd import _lib as L          # noqa: E402
from palette_and_histo_gan_amd import engine as E        # noqa: E402

B = int(os.environ.get("UB_BATCH", "256"))
COLD = os.environ.get("UB_COLD", "0") != "0"
SHAPES = [  # (name, op, lh, cg, cd)
    ("up5.fwd", L.OP_P, 16, 64, 256), ("up4.fwd", L.OP_P, 8, 128, 512), ("down2.dgrad", L.OP_P, 16, 64, 128),
    ("down3.dgrad", L.OP_P, 8, 128, 256), ("up5.dgrad", L.OP_G, 16, 64, 256), ("up4.dgrad", L.OP_G, 8, 128, 512),
    ("down2.fwd", L.OP_G, 16, 64, 128), ("down3.fwd", L.OP_G, 8, 128, 256),
    ("up3.fwd", L.OP_P, 4, 256, 1024), ("up3.dgrad", L.OP_G, 4, 256, 1024), ("down4.fwd", L.OP_G, 4, 256, 512),
    ("down4.dgrad", L.OP_P, 4, 256, 512), ("up2.fwd", L.OP_P, 2, 512, 1024), ("up2.dgrad", L.OP_G, 2, 512, 1024),
    ("down5.fwd", L.OP_G, 2, 512, 512), ("down5.dgrad", L.OP_P, 2, 512, 512),
]


def main():
    only = sys.argv[1].split(",") if len(sys.argv) > 1 else None
    reps = int(os.environ.get("UB_REPS", "20"))
    dev = "cuda:0"
    eng = E.Pix2PixEngine(4, 4, "tanh", 64, L.BF16, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device=dev).manual_seed(1)
    tot = 0.0
    flush = torch.empty(256 << 20, dtype=torch.float32, device=dev) if COLD else None
    for name, op, lh, cg, cd in SHAPES:
        if only and not any(name.startswith(o) for o in only):
            continue
        hi = E.HaloBuf(B, 2 * lh, 2 * lh, cg, L.BF16, dev)
        lo = E.HaloBuf(B, lh, lh, cd, L.BF16, dev)
        for hb in (hi, lo):
            hb.t[:, 2:-2, 2:-2, :] = torch.randn((B, hb.h, hb.w, hb.c), device=dev, generator=g).to(torch.bfloat16)
        w = (0.05 * torch.randn(16 * cg * cd, device=dev, generator=g)).to(torch.bfloat16)
        out = E.DenseBuf(B, lh if op == L.OP_G else 2 * lh, lh if op == L.OP_G else 2 * lh, cd if op == L.OP_G else cg, torch.bfloat16, dev)
        sk = eng._splitk(op, B, lh, cg, cd)
        slabs = torch.empty(sk * out.t.numel() if sk > 1 else 4, dtype=torch.float32, device=dev)
        hv, lv = (hi.view(), out.view()) if op == L.OP_G else (out.view(), lo.view())

        def launch():
            L.call("p2p_igemm", op, L.BF16, B, lh, lh, cg, cd, C.byref(hv), C.byref(lv), C.c_void_p(w.data_ptr()), sk,
                   C.c_void_p(slabs.data_ptr()) if sk > 1 else None, None, st)
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        if COLD:
            # the step's condition: operands last touched hundreds of MB of traffic ago.  A 1 GiB fill between launches evicts
            # L2 and the Infinity Cache; every launch is timed on its own (the event pair adds ~3-5 us to each)
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
        fl = 2.0 * B * lh * lh * 16 * cg * cd
        tot += us
        brig = L.lib().p2p_brig_ok(op, L.BF16, B, lh, lh, cg, cd) if sk == 1 else 0
        print(f"{name:12s} op={'GP'[op]} lo={lh:2d} cg={cg:4d} cd={cd:4d} sk={sk} {'brig  ' if brig else 'im2col'} {us:7.1f} us {fl / us * 1e-6:7.1f} TFLOP/s", flush=True)
    print(f"total {tot:.1f} us")


if __name__ == "__main__":
    main()
