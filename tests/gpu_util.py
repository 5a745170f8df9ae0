"""Helpers shared by the -m gpu parity tests: move oracle tensors into the engine's HBM layout and back."""
import ctypes as C

import numpy as np
import torch

from palette_and_histo_gan_amd import _lib as L
from palette_and_histo_gan_amd import engine as E

DEV = "cuda:0"


def tdt(dtype):
    return torch.float32 if dtype == L.F32 else torch.bfloat16


def halo_from(x, dtype):
    """numpy (N,H,W,C) -> HaloBuf holding it in the interior."""
    n, h, w, c = x.shape
    hb = E.HaloBuf(n, h, w, c, dtype, DEV)
    hb.t[:, E.HALO:E.HALO + h, E.HALO:E.HALO + w, :] = torch.as_tensor(x).to(DEV).to(tdt(dtype))
    return hb


def halo_to_np(hb):
    return hb.t[:, E.HALO:E.HALO + hb.h, E.HALO:E.HALO + hb.w, :].float().cpu().numpy()


def dense_to_np(db):
    return db.t.float().cpu().numpy().reshape(db.n, db.h, db.w, db.c)


def dev(x, dt=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(x)).to(DEV).to(dt).contiguous()


def ptr(t):
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def q(x, dtype):
    """round a numpy array through the activation dtype (so the oracle sees what the kernel sees)."""
    if dtype == L.F32:
        return np.asarray(x, np.float32)
    return torch.as_tensor(np.asarray(x, np.float32)).to(torch.bfloat16).float().numpy()


def rel_err(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))
