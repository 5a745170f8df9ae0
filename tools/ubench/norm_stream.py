#!/usr/bin/env python3
"""InstanceNorm forward / backward launches of the c2 step (B = 256, bf16) timed alone, back to back (operands in the Infinity
Cache) and with 1 GiB written between launches (cold, as inside the step), next to a plain device copy of the same bytes --
what this chip gives a simple read + write stream.  Timing only; not part of the product."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from palette_and_histo_gan_amd import _lib as L          # noqa: E402
from palette_and_histo_gan_amd import engine as E        # noqa: E402

DEV = "cuda:0"


def timed(fn, flush, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    warm = a.elapsed_time(b) / reps * 1e3
    cold = 0.0
    for _ in range(8):
        flush.fill_(1.0)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        cold += a.elapsed_time(b) * 1e3 / 8
    return warm, cold


def main():
    B = 256
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    flush = torch.empty(1 << 28, dtype=torch.float32, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(1)
    for (h, c, tail) in ((64, 32, 8), (32, 64, 0), (16, 128, 0)):
        raw = E.DenseBuf(B, h, h, c, torch.bfloat16, DEV)
        raw.t.copy_(torch.randn(raw.t.shape, device=DEV, generator=g).to(torch.bfloat16))
        out = E.HaloBuf(B, h, h, c + (tail or c), L.BF16, DEV)          # a slice of a wider concat buffer
        src = E.HaloBuf(B, h, h, 8, L.BF16, DEV)
        gamma, beta = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
        stats = torch.empty((B, c, 2), dtype=torch.float32, device=DEV)
        nws = torch.empty(B * 16 * c * 2, dtype=torch.float32, device=DEV)
        gsrc_buf = E.DenseBuf(B, h, h, c, torch.bfloat16, DEV)
        gsrc_buf.t.copy_(torch.randn(raw.t.shape, device=DEV, generator=g).to(torch.bfloat16))
        draw = E.HaloBuf(B, h, h, c, L.BF16, DEV)
        part = torch.zeros((2, B, c), dtype=torch.float32, device=DEV)
        t_bytes = B * h * h * c * 2
        args = (L.BF16, B, h, h, c, raw.ptr(), 1, 1, 0, C.c_void_p(gamma.data_ptr()), C.c_void_p(beta.data_ptr()), 1e-3, L.ACT_RELU, 0.3,
                None, C.byref(out.view(coff=0)), None, C.c_void_p(stats.data_ptr()), C.c_void_p(nws.data_ptr()), nws.numel() * 4)

        def fwd(nsplit):
            if tail:
                L.call("p2p_norm_act_fwd_tail", *args, nsplit, C.byref(src.view()), 8, st)
            else:
                L.call("p2p_norm_act_fwd", *args, nsplit, st)

        def bwd(nsplit):
            L.call("p2p_norm_act_bwd", L.BF16, B, h, h, c, raw.ptr(), C.c_void_p(stats.data_ptr()), C.c_void_p(gamma.data_ptr()),
                   C.c_void_p(beta.data_ptr()), L.ACT_RELU, 0.3, None, C.byref(gsrc_buf.gsrc()), None, C.byref(draw.view()),
                   C.c_void_p(part[1].data_ptr()), C.c_void_p(part[0].data_ptr()), C.c_void_p(nws.data_ptr()), nws.numel() * 4, nsplit, st)
        dst = torch.empty_like(raw.t)
        w, cd = timed(lambda: dst.copy_(raw.t), flush)
        print(f"{h:3d}x{h:<3d} c={c:3d}  device copy (r + w {2 * t_bytes / 1e6:.0f} MB)      warm {w:6.1f} us {2 * t_bytes / w * 1e-6:5.2f} TB/s   cold {cd:6.1f} us {2 * t_bytes / cd * 1e-6:5.2f} TB/s", flush=True)
        for ns in (1, 4, 8):
            w, cd = timed(lambda: fwd(ns), flush)
            by = 3 * t_bytes + (2 * B * h * h * 16 if tail else 0)
            print(f"{h:3d}x{h:<3d} c={c:3d}  norm fwd nsplit={ns} (3 passes {by / 1e6:.0f} MB)   warm {w:6.1f} us {by / w * 1e-6:5.2f} TB/s   cold {cd:6.1f} us {by / cd * 1e-6:5.2f} TB/s", flush=True)
        for ns in (1, 4, 8, 16):
            w, cd = timed(lambda: bwd(ns), flush)
            by = 5 * t_bytes
            print(f"{h:3d}x{h:<3d} c={c:3d}  norm bwd nsplit={ns} (5 passes {by / 1e6:.0f} MB)   warm {w:6.1f} us {by / w * 1e-6:5.2f} TB/s   cold {cd:6.1f} us {by / cd * 1e-6:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
