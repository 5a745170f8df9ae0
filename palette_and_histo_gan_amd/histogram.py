"""Module-level functions with the reference's names (histogram.py:35-97) for callers outside the fused train step.

`calculate_rgbuv_histogram` runs the HIP forward kernel (p2p_rgbuv_hist_fwd); the scalar distances are a handful of
elementwise ops on the (B,64,64,3) result.  Inside train_step none of this is used: the loss and its gradient are
computed by the fused kernels without materialising the normalised histogram.
"""
import ctypes as C
import math

import torch

from . import _lib as L


def calculate_rgbuv_histogram(image_batch, size=64, method="inverse-quadratic", sigma=0.02, device="cuda:0"):
    """histogram.py:35-81 (size 64, inverse-quadratic kernel, sigma 0.02 are the only values the reference uses).
    (B, S, S, 4) values in [-1, 1] -> normalised (B, 64, 64, 3) f32 device tensor.  Two launches through the C ABI
    (p2p_rgbuv_hist_fwd + p2p_hist_normalize) on the current stream; no engine, no parameters are created for it."""
    if size != 64 or method != "inverse-quadratic" or abs(sigma - 0.02) > 1e-12:
        raise NotImplementedError("only the reference's call (size=64, inverse-quadratic, sigma=0.02) is built")
    L.lib()          # fail loudly if the HIP library is missing: there is no CPU path
    dev = torch.device(device)
    img = torch.as_tensor(image_batch).to(device=dev, dtype=torch.float32).contiguous()
    if img.dim() != 4 or img.shape[3] != 4:
        raise ValueError(f"expected a (B, H, W, 4) batch, got {tuple(img.shape)}")
    B, H, W = int(img.shape[0]), int(img.shape[1]), int(img.shape[2])
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        raw = torch.empty(B * 3 * 64 * 64, dtype=torch.float32, device=dev)
        out = torch.empty((B, 64, 64, 3), dtype=torch.float32, device=dev)
        L.call("p2p_rgbuv_hist_fwd", L.F32, B, H, W, C.byref(L.Tensor(img.data_ptr(), H * W, W, 4)), C.c_void_p(raw.data_ptr()), stream)
        L.call("p2p_hist_normalize", C.c_void_p(raw.data_ptr()), B, C.c_void_p(out.data_ptr()), stream)
    return out


def hellinger_loss(y_true, y_pred):
    """histogram.py:84-89"""
    b = y_true.shape[0]
    return (1.0 / math.sqrt(2.0)) * torch.sqrt(((torch.sqrt(y_pred) - torch.sqrt(y_true)) ** 2).sum()) / b


def l1_loss(y_true, y_pred):
    """histogram.py:92-93"""
    return (y_true - y_pred).abs().mean()


def l2_loss(y_true, y_pred):
    """histogram.py:96-97"""
    return ((y_true - y_pred) ** 2).mean()
