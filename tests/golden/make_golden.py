"""Generates tests/golden/*.npz from the float64 CPU oracle (run from the repo root: python tests/golden/make_golden.py).

The reference cannot run here (TensorFlow is not installable), so these vectors pin the ORACLE (they
catch an accidental change of its semantics) and give the GPU tests fixed expected outputs; they do not pin the oracle
to TensorFlow -- "parity unpinned" (oracle/__init__.py).  Inputs are regenerated from seeds (numpy default_rng), the
expected outputs are stored: per train step the loss scalars and, per gradient tensor, sum, abs-sum and 64 entries at
fixed pseudo-random positions."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import reference_graph as rg    # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
F64 = torch.float64


def sample_positions(name, size, k=64):
    rng = np.random.default_rng(abs(hash_name(name)) % (2 ** 32))
    return rng.integers(0, size, size=min(k, size))


def hash_name(name):
    h = 2166136261
    for ch in name.encode():
        h = ((h ^ ch) * 16777619) % (2 ** 32)
    return h


def summarise(grads, prefix, out):
    for k, g in grads.items():
        a = g.numpy().reshape(-1)
        out[f"{prefix}.{k}.sum"] = a.sum()
        out[f"{prefix}.{k}.abssum"] = np.abs(a).sum()
        out[f"{prefix}.{k}.samples"] = a[sample_positions(k, a.size)]


def rgba_case(seed, lambda_l1, lambda_hist):
    B, S = 2, 64
    rng = np.random.default_rng(seed)
    Gp = rg.perturb_affine(rg.init_params(rg.generator_param_shapes(4, 4), rng, F64), rng)
    Dp = rg.perturb_affine(rg.init_params(rg.discriminator_param_shapes(4), rng, F64), rng)
    src, tgt = rg.synthetic_rgba_batch(rng, B, S)
    masks = [rng.integers(0, 2, size=s).astype(np.uint8) for s in rg.dropout_mask_shapes(B, S)]
    return Gp, Dp, src, tgt, masks


def main():
    out = {}
    for tag, seed, l1, lh in (("baseline", 101, 100.0, None), ("histogram", 102, 30.0, 1.0)):
        Gp, Dp, src, tgt, masks = rgba_case(seed, l1, lh)
        ref = rg.train_step_rgba(Gp, Dp, torch.tensor(src, dtype=F64), torch.tensor(tgt, dtype=F64),
                                 [torch.tensor(m, dtype=F64) for m in masks], lambda_l1=l1, lambda_hist=lh)
        out[f"{tag}.g_loss"] = np.array(ref["g_loss"])
        out[f"{tag}.d_loss"] = np.array(ref["d_loss"])
        summarise(ref["g_grads"], f"{tag}.G", out)
        summarise(ref["d_grads"], f"{tag}.D", out)
        # the SAME oracle graph evaluated in float32: what f32 arithmetic itself does to this case's gradient samples (the yardstick
        # of the engine's f32 mode; the histogram case holds near-black fake pixels whose gradient is 1 / (x + 1e-6)-steep)
        f32 = torch.float32
        r32 = rg.train_step_rgba({k: v.to(f32) for k, v in Gp.items()}, {k: v.to(f32) for k, v in Dp.items()},
                                 torch.tensor(src, dtype=f32), torch.tensor(tgt, dtype=f32), [torch.tensor(m, dtype=f32) for m in masks],
                                 lambda_l1=l1, lambda_hist=lh)
        for k, g in r32["g_grads"].items():
            a = g.numpy().reshape(-1).astype(np.float64)
            want = out[f"{tag}.G.{k}.samples"]
            scale = max(np.abs(want).max(), out[f"{tag}.G.{k}.abssum"] / a.size + 1e-30)
            out[f"{tag}.G.{k}.f32dev"] = np.abs(a[sample_positions(k, a.size)] - want).max() / scale
            out[f"{tag}.G.{k}.f32dev_abssum"] = abs(np.abs(a).sum() - out[f"{tag}.G.{k}.abssum"]) / (out[f"{tag}.G.{k}.abssum"] + 1e-300)
    # indexed model
    rng = np.random.default_rng(103)
    Gp = rg.perturb_affine(rg.init_params(rg.generator_param_shapes(1, 256), rng, F64), rng)
    Dp = rg.perturb_affine(rg.init_params(rg.discriminator_param_shapes(1), rng, F64), rng)
    Gp["down1.kernel"] *= 0.05
    Dp["down.kernel"] *= 0.05
    src, tgt, _ = rg.synthetic_indexed_batch(rng, 2, 64)
    masks = [rng.integers(0, 2, size=s).astype(np.uint8) for s in rg.dropout_mask_shapes(2, 64)]
    ref = rg.train_step_indexed(Gp, Dp, torch.tensor(src), torch.tensor(tgt), [torch.tensor(m, dtype=F64) for m in masks], 0.01)
    out["indexed.g_loss"] = np.array(ref["g_loss"])
    out["indexed.d_loss"] = np.array(ref["d_loss"])
    summarise(ref["g_grads"], "indexed.G", out)
    summarise(ref["d_grads"], "indexed.D", out)
    # histogram of a fixed small image + its Hellinger gradient
    rng = np.random.default_rng(104)
    s8, t8 = rg.synthetic_rgba_batch(rng, 2, 8, palette_size=6)
    fake = np.clip(s8 + rng.normal(scale=0.05, size=s8.shape), -1, 1)
    ft = torch.tensor(fake, dtype=F64, requires_grad=True)
    hr = rg.rgbuv_histogram(torch.tensor(t8, dtype=F64))
    hf = rg.rgbuv_histogram(ft)
    loss = rg.hellinger_loss(hr, hf)
    loss.backward()
    out["hist8.fake"] = fake
    out["hist8.real"] = t8
    out["hist8.hist_real"] = hr.numpy().astype(np.float32)
    out["hist8.loss"] = np.array(float(loss))
    out["hist8.dfake"] = ft.grad.numpy()
    # argmax with engineered exact ties (bit-exact expectation)
    rng = np.random.default_rng(105)
    p = rng.random((64, 256)).astype(np.float32)
    p /= p.sum(1, keepdims=True)
    for r in range(0, 64, 3):
        i, j = sorted(rng.choice(256, 2, replace=False))
        p[r, i] = p[r, j] = p[r].max() * 2
    p[7, :] = 1.0 / 256
    out["argmax.probs"] = p
    out["argmax.index"] = np.argmax(p, -1).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **out)
    print("wrote", os.path.join(HERE, "oracle_vectors.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
