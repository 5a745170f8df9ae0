"""Module-level functions with the reference's names (histogram.py:35-97) for callers outside the fused train step.

`calculate_rgbuv_histogram` runs the HIP forward kernel (p2p_rgbuv_hist_fwd); the scalar distances are a handful of
elementwise ops on the (B,64,64,3) result.  Inside train_step none of this is used: the loss and its gradient are
computed by the fused kernels without materialising the normalised histogram.
"""
import math

import torch

from . import _lib as L

_ENGINES = {}


def _engine(img_size, device):
    from .engine import Pix2PixEngine
    key = (img_size, str(device))
    if key not in _ENGINES:
        _ENGINES[key] = Pix2PixEngine(4, 4, "tanh", img_size, L.F32, device=device)
    return _ENGINES[key]


def calculate_rgbuv_histogram(image_batch, size=64, method="inverse-quadratic", sigma=0.02, device="cuda:0"):
    """histogram.py:35-81 (size 64, inverse-quadratic kernel, sigma 0.02 are the only values the reference uses)."""
    if size != 64 or method != "inverse-quadratic" or abs(sigma - 0.02) > 1e-12:
        raise NotImplementedError("only the reference's call (size=64, inverse-quadratic, sigma=0.02) is built")
    return _engine(int(image_batch.shape[1]), device).rgbuv_histogram(image_batch)


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
