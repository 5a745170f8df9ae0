#!/usr/bin/env python3
"""Times p2p_pack_pair on the c2 plan (B = 256, 64x64) with and without its two PARTIAL-pixel stores (v_c6: 16 of every 80
bytes, v_dfake: 8 of every 16 bytes), back to back (operands in the Infinity Cache) and with 1 GiB written between launches (cold,
as in the train step).  r03 result: cold 71 us with both, 30 us without v_c6, 24 us without both -- partial 32-byte sectors
cost a read-modify-write in HBM.  Timing only; not part of the product."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from palette_and_histo_gan_amd import _lib as L          # noqa: E402
from palette_and_histo_gan_amd import engine as E        # noqa: E402


def main():
    B, S, dev = 256, 64, "cuda:0"
    eng = E.Pix2PixEngine(4, 4, "tanh", S, L.BF16, device=dev)
    P = eng.plan(B)
    src = torch.rand(B, S, S, 4, device=dev)
    tgt = torch.rand(B, S, S, 4, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    views = (P["src"].view(), P["c"][6].view(coff=E.UP_FILTERS[5]), P["dcat"].view(coff=0), P["dcat"].view(coff=0, n0=B))
    flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    for abl in (0, 2, 8, 10):
        def launch():
            L.call("p2p_pack_pair", L.BF16, B, S, S, C.c_void_p(src.data_ptr()), C.c_void_p(tgt.data_ptr()),
                   C.byref(views[0]), None if abl & 2 else C.byref(views[1]), C.byref(views[2]),
                   None if abl & 8 else C.byref(views[3]), st)
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            launch()
        b.record()
        torch.cuda.synchronize()
        warm = a.elapsed_time(b) / 50 * 1e3
        # cold: 1 GiB written between the launches, so neither L2 nor the 256 MB Infinity Cache holds the operands (as in the step)
        cold = 0.0
        for _ in range(10):
            flush.fill_(1.0)
            a.record()
            launch()
            b.record()
            torch.cuda.synchronize()
            cold += a.elapsed_time(b) * 1e3 / 10
        print(f"abl={abl:2d}  warm {warm:7.1f} us   cold {cold:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
