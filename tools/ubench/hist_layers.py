"""Launch-by-launch timing of the histogram kernels at the c3 (256 x 64 x 64) and c5 (256 x 128 x 128) shapes.

    python tools/ubench/hist_layers.py [N S] ...

Prints ms per launch of p2p_rgbuv_hist_fwd3 (dense and colour-point form) and p2p_rgbuv_hist_hellinger_bwd3 and a checksum of each
result (to compare builds: the sums move only by rounding when a kernel is re-scheduled)."""
import ctypes as C
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from palette_and_histo_gan_amd import _lib as L  # noqa: E402

DEV = torch.device("cuda:0")
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def sprites(rng, N, S, colours=24):
    """sprite-like RGBA in [-1, 1]: a palette of opaque colours per image, 83.5 % of the pixels transparent black"""
    out = np.zeros((N, S, S, 4), np.uint8)
    for b in range(N):
        pal = np.concatenate([rng.integers(0, 256, size=(colours, 3)), np.full((colours, 1), 255)], axis=1).astype(np.uint8)
        opaque = rng.random((S, S)) >= 0.835
        out[b] = np.where(opaque[..., None], pal[rng.integers(0, colours, size=(S, S))], 0)
    return out.astype(np.float32) / 127.5 - 1.0


def one(N, S):
    rng = np.random.default_rng(5)
    tgt = sprites(rng, N, S)
    fake = np.clip(tgt + rng.normal(scale=0.05, size=tgt.shape), -1, 1).astype(np.float32)
    cap = 1024
    tt, ft = torch.tensor(tgt, device=DEV).contiguous(), torch.tensor(fake, device=DEV).contiguous()
    vt, vf = L.Tensor(tt.data_ptr(), S * S, S, 4), L.Tensor(ft.data_ptr(), S * S, S, 4)
    n = N * 3 * 64 * 64
    h_r, h_f, gh = (torch.empty(n, dtype=torch.float32, device=DEV) for _ in range(3))
    ws = torch.empty(L.lib().p2p_rgbuv_hist_fwd3_workspace_bytes(N) // 4, dtype=torch.float32, device=DEV)
    pts = torch.empty((N, cap, 4), dtype=torch.float32, device=DEV)
    npts = torch.empty((N,), dtype=torch.int32, device=DEV)
    tot = torch.empty((2, N), dtype=torch.float32, device=DEV)
    sq = torch.zeros(4, dtype=torch.float32, device=DEV)
    sqp = torch.zeros(N, dtype=torch.float32, device=DEV)
    dimg = torch.empty(N * S * S * 4, dtype=torch.float32, device=DEV)
    L.call("p2p_rgbuv_points", L.F32, N, S, S, C.byref(vt), cap, p(pts), p(npts), st())
    f_pts = lambda: L.call("p2p_rgbuv_hist_fwd3", L.F32, N, S, S, C.byref(vt), p(pts), p(npts), cap, p(h_r), p(ws), st())
    f_dense = lambda: L.call("p2p_rgbuv_hist_fwd3", L.F32, N, S, S, C.byref(vf), None, None, cap, p(h_f), p(ws), st())
    f_pts(); f_dense()
    L.call("p2p_hellinger_fwd", p(h_r), p(h_f), N, p(tot[0]), p(tot[1]), p(sqp), p(sq), st())
    f_bwd = lambda: L.call("p2p_rgbuv_hist_hellinger_bwd3", L.F32, N, S, S, C.byref(vf), p(h_r), p(h_f), p(tot[0]), p(tot[1]), p(sq),
                           1.0 / (2.0 * math.sqrt(2.0) * N), p(gh), p(dimg), st())
    t_pts, t_dense, t_bwd = timed(f_pts), timed(f_dense), timed(f_bwd)
    torch.cuda.synchronize()
    print(f"N={N} S={S}: fwd3(points, npts~{int(npts.float().mean())}) {t_pts:.4f} ms   fwd3(dense) {t_dense:.4f} ms   bwd3 {t_bwd:.4f} ms   "
          f"sums: h_r {h_r.double().sum().item():.9e} h_f {h_f.double().sum().item():.9e} |dimg| {dimg.double().abs().sum().item():.9e}",
          flush=True)


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    shapes = list(zip(a[0::2], a[1::2])) or [(256, 64), (256, 128)]
    for N, S in shapes:
        one(N, S)
