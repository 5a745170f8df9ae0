"""Is p2p_rgbuv_hist_hellinger_bwd3 / fwd3 reproducible launch to launch?  Runs each 10 times on the same input and reports where the
results differ (pixel index inside the image, batch of 64, lane)."""
import ctypes as C
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "ubench"))
from palette_and_histo_gan_amd import _lib as L  # noqa: E402
from hist_layers import sprites, p, st, DEV  # noqa: E402

N, S = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(5)
tgt = sprites(rng, N, S)
fake = np.clip(tgt + rng.normal(scale=0.05, size=tgt.shape), -1, 1).astype(np.float32)
tt, ft = torch.tensor(tgt, device=DEV).contiguous(), torch.tensor(fake, device=DEV).contiguous()
vt, vf = L.Tensor(tt.data_ptr(), S * S, S, 4), L.Tensor(ft.data_ptr(), S * S, S, 4)
n = N * 3 * 64 * 64
h_r, h_f, gh = (torch.empty(n, dtype=torch.float32, device=DEV) for _ in range(3))
ws = torch.empty(L.lib().p2p_rgbuv_hist_fwd3_workspace_bytes(N) // 4, dtype=torch.float32, device=DEV)
tot = torch.empty((2, N), dtype=torch.float32, device=DEV)
sq = torch.zeros(4, dtype=torch.float32, device=DEV)
sqp = torch.zeros(N, dtype=torch.float32, device=DEV)
L.call("p2p_rgbuv_hist_fwd3", L.F32, N, S, S, C.byref(vt), None, None, 1024, p(h_r), p(ws), st())
outs_f, outs_b = [], []
for i in range(10):
    h_f.fill_(float("nan"))
    L.call("p2p_rgbuv_hist_fwd3", L.F32, N, S, S, C.byref(vf), None, None, 1024, p(h_f), p(ws), st())
    outs_f.append(h_f.clone())
L.call("p2p_hellinger_fwd", p(h_r), p(h_f), N, p(tot[0]), p(tot[1]), p(sqp), p(sq), st())
for i in range(int(os.environ.get("REPS", "40"))):
    dimg = torch.full((N * S * S * 4,), float("nan"), dtype=torch.float32, device=DEV)
    L.call("p2p_rgbuv_hist_hellinger_bwd3", L.F32, N, S, S, C.byref(vf), p(h_r), p(h_f), p(tot[0]), p(tot[1]), p(sq),
           1.0 / (2.0 * math.sqrt(2.0) * N), p(gh), p(dimg), st())
    outs_b.append(dimg.view(N, S * S, 4).clone())
torch.cuda.synchronize()
for name, outs in (("fwd3", outs_f), ("bwd3", outs_b)):
    ref = outs[0]
    print(name, "finite:", bool(torch.isfinite(ref).all()))
    bad = 0
    for i in range(1, len(outs)):
        d = (outs[i] != ref)
        k = int(d.sum())
        bad += 1 if k else 0
        if k and bad <= 3:
            idx = d.nonzero()[:8].cpu().numpy()
            rel = float(((outs[i] - ref).abs().max() / ref.abs().max()).item())
            print(f"  run {i}: {k} elements differ, max rel {rel:.2e}, first: {idx.tolist()}")
            if name == "bwd3":
                for (a, b_, c_) in idx[:4]:
                    print("     pixel", a, b_, "x =", ft.view(N, S * S, 4)[a, b_].cpu().numpy() * 0.5 + 0.5, "ref", ref[a, b_].cpu().numpy(), "run", outs[i][a, b_].cpu().numpy())
                np.save(f"gpurun_out/hist_repro_ref.npy", ref.cpu().numpy()); np.save(f"gpurun_out/hist_repro_run.npy", outs[i].cpu().numpy())
    print(f"  {name}: {bad} of {len(outs) - 1} repeats differ from the first run")
