"""Not-gpu: the data-parallel algebra of SURVEY.md 8e over gloo with world_size 2 (one process per rank), with the
CPU oracle as the per-rank compute: global-count loss denominators + SUM all-reduce == the single-process global batch,
including the Hellinger loss's batch-wide square root."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import np_restatement as npr
from oracle import reference_graph as rg
from palette_and_histo_gan_amd import parallel as PAR

F64 = torch.float64
B, S = 2, 64


def _case():
    rng = np.random.default_rng(41)
    Gp = rg.perturb_affine(rg.init_params(rg.generator_param_shapes(4, 4), rng, F64), rng)
    Dp = rg.perturb_affine(rg.init_params(rg.discriminator_param_shapes(4), rng, F64), rng)
    src, tgt = rg.synthetic_rgba_batch(rng, B, S)
    masks = [rng.integers(0, 2, size=s).astype(np.float64) for s in rg.dropout_mask_shapes(B, S)]
    return Gp, Dp, src, tgt, masks


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
    torch.set_num_threads(2)
    comm = PAR.init_data_parallel("cpu", backend="gloo")
    Gp, Dp, src, tgt, masks = _case()
    lo, hi = PAR.shard_bounds(B, world, rank)
    sm = [torch.tensor(m[lo:hi], dtype=F64) for m in masks]
    out = rg.train_step_rgba(Gp, Dp, torch.tensor(src[lo:hi], dtype=F64), torch.tensor(tgt[lo:hi], dtype=F64), sm, 100.0)
    # every loss is an element mean: local mean * (B_local / B_global) == the rank's share of the global mean
    w = (hi - lo) / B
    g_flat = torch.cat([g.reshape(-1) for g in out["g_grads"].values()]) * w
    d_flat = torch.cat([g.reshape(-1) for g in out["d_grads"].values()]) * w
    losses = torch.tensor([out["g_loss"][1] * w, out["g_loss"][2] * w, out["d_loss"][1] * w, out["d_loss"][2] * w], dtype=F64)
    comm.allreduce_grads(g_flat, d_flat, losses)
    # Hellinger coupling: local sum of squares -> one scalar all-reduce -> global loss and gradient
    fake = out["fake"]
    hr = rg.rgbuv_histogram(torch.tensor(tgt[lo:hi], dtype=F64))
    hf = rg.rgbuv_histogram(fake)
    sq = ((hf.sqrt() - hr.sqrt()) ** 2).sum().reshape(1)
    comm.allreduce_scalar_sum(sq)
    hell = float(torch.sqrt(sq[0]) / np.sqrt(2.0) / B)
    dfake = npr.hist_hellinger_backward(fake.numpy(), hr.numpy(), global_sq_sum=float(sq[0]), global_batch=B)
    t = comm.max_scalar(float(rank))
    comm.barrier()
    if rank == 0:
        q.put(dict(g=g_flat.numpy(), d=d_flat.numpy(), losses=losses.numpy(), hell=hell, maxrank=t))
    q.put(("dfake", rank, lo, hi, dfake))
    comm.barrier()
    comm.destroy()


@pytest.mark.timeout(600)
def test_two_rank_sharded_step_equals_global_batch_step():
    world, port = 2, 29631
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, dfakes = None, {}
    for _ in range(world + 1):
        item = q.get(timeout=500)
        if isinstance(item, dict):
            got = item
        else:
            dfakes[item[1]] = item[2:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    Gp, Dp, src, tgt, masks = _case()
    ref = rg.train_step_rgba(Gp, Dp, torch.tensor(src, dtype=F64), torch.tensor(tgt, dtype=F64),
                             [torch.tensor(m, dtype=F64) for m in masks], 100.0)
    g_ref = torch.cat([g.reshape(-1) for g in ref["g_grads"].values()]).numpy()
    d_ref = torch.cat([g.reshape(-1) for g in ref["d_grads"].values()]).numpy()
    assert np.abs(got["g"] - g_ref).max() <= 1e-9 * np.abs(g_ref).max()
    assert np.abs(got["d"] - d_ref).max() <= 1e-9 * np.abs(d_ref).max()
    np.testing.assert_allclose(got["losses"], [ref["g_loss"][1], ref["g_loss"][2], ref["d_loss"][1], ref["d_loss"][2]], rtol=1e-10)
    assert got["maxrank"] == 1.0
    # Hellinger: the sharded loss with the all-reduced sum equals the global-batch loss; so does its gradient
    fake = ref["fake"].clone().requires_grad_(True)
    hr = rg.rgbuv_histogram(torch.tensor(tgt, dtype=F64))
    hl = rg.hellinger_loss(hr, rg.rgbuv_histogram(fake))
    hl.backward()
    assert abs(got["hell"] - float(hl)) < 1e-10 * float(hl)
    for rank, (lo, hi, df) in dfakes.items():
        np.testing.assert_allclose(df, fake.grad.numpy()[lo:hi], rtol=1e-7, atol=1e-14)
