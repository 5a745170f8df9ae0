"""-m gpu: data parallelism through the reference-API model classes (build-added capability): N ranks that are
handed the same global batch == one rank, for the indexed model, a ragged global batch, a rank with an EMPTY shard, and the
device dropout RNG (keyed by the global sample index, no injected masks).  Transport: gloo between two processes that
share the test box's single GPU; plus one step over a world-1 `nccl` (= RCCL) process group so that the asynchronous
bucket all-reduces and their stream ordering run under test."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from palette_and_histo_gan_amd import dataset_utils as D
from palette_and_histo_gan_amd import parallel as PAR
from palette_and_histo_gan_amd import pix2pix_model as M

pytestmark = pytest.mark.gpu

CASES = [  # (tag, model, global batch)
    ("indexed", "indexed", 4), ("ragged", "baseline", 5), ("empty", "histogram", 1)]


def _batch(model, Bg):
    if model == "indexed":
        return next(iter(D.synthetic_indexed_ds(Bg, batch_size=Bg, seed=9)))
    return next(iter(D.synthetic_rgba_ds(Bg, batch_size=Bg, palette_size=24, seed=9)))


def _make(model, ds, dp):
    kw = dict(dtype="f32", data_parallel=dp)
    if model == "indexed":
        return M.Pix2PixIndexedModel(ds, ds, "front2right", "dp-test", lambda_segmentation=0.01, **kw)
    if model == "histogram":
        return M.Pix2PixHistogramModel(ds, ds, "front2right", "dp-test", 30.0, 1.0, **kw)
    return M.Pix2PixModel(ds, ds, "front2right", "dp-test", 100.0, **kw)


def _one_step(model, Bg, dp):
    batch = _batch(model, Bg)
    m = _make(model, None, dp)
    g_loss, d_loss = m.train_step(batch, 0, 1)
    torch.cuda.synchronize()
    e = m.engine
    return (np.array([float(x) for x in g_loss + d_loss]), e.G.grads.cpu().numpy().copy(), e.D.grads.cpu().numpy().copy(),
            e.G.params.cpu().numpy().copy())


def _worker(rank, world, port, q, tmp):
    import faulthandler
    import sys
    faulthandler.dump_traceback_later(int(os.environ.get("P2P_TEST_DUMP_AFTER", "240")), exit=True, file=sys.stderr)       # a deadlocked rank reports where it stands
    os.chdir(tmp)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    try:
        comm = PAR.DataParallel("cuda:0", backend="gloo")
        for tag, model, Bg in CASES:
            res = _one_step(model, Bg, comm)
            print(f"[rank {rank}] case {tag} done", flush=True)
            if rank == 0:
                q.put((tag,) + res)
            comm.barrier()
        comm.destroy()
    except BaseException:       # a failing rank must not leave the parent (and its sibling) waiting
        import traceback
        q.put(("error", rank, traceback.format_exc()))
        q.close()
        q.join_thread()
        os._exit(1)
    q.close()
    q.join_thread()
    os._exit(0)          # skip interpreter teardown of a process that shares the GPU with its sibling rank


@pytest.mark.timeout(1200)
def test_two_rank_models_equal_one_rank(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    world, port = 2, 29651
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in CASES:
        item = q.get(timeout=600)
        if item[0] == "error":
            for p in procs:
                p.kill()
            pytest.fail(f"rank {item[1]} failed:\n{item[2]}")
        got[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0, f"rank process exited with {p.exitcode}"
    for tag, model, Bg in CASES:
        l1, g1, d1, p1 = _one_step(model, Bg, None)
        l2, g2, d2, p2 = got[tag]
        np.testing.assert_allclose(l2, l1, rtol=5e-6, atol=1e-9, err_msg=tag)
        assert np.abs(g2 - g1).max() <= 2e-5 * np.abs(g1).max(), tag
        assert np.abs(d2 - d1).max() <= 2e-5 * np.abs(d1).max(), tag
        assert np.abs(p2 - p1).max() < 3e-5, tag       # same Adam step (rounding-level gradients may move by a fraction of lr)


@pytest.mark.timeout(600)
def test_world1_rccl_step_matches_plain_step(tmp_path, monkeypatch):
    """One process, one rank, backend nccl (RCCL): the bucketed asynchronous all-reduces issued from the weight-gradient
    stream, wait_all(), the early Adam on the reduced buckets and the tail collective all execute; the result must be the
    plain single-GPU step bit for bit (a SUM over one rank)."""
    import torch.distributed as dist
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29653")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert not dist.is_initialized()
    comm = PAR.DataParallel("cuda:0", backend="nccl")
    try:
        batch = _batch("baseline", 4)
        ref = _one_step("baseline", 4, None)
        m = _make("baseline", None, comm)
        for step in range(2):
            g_loss, d_loss = m.train_step(batch, step, 1)
        torch.cuda.synchronize()
        m1 = _make("baseline", None, None)
        for step in range(2):
            g1, d1 = m1.train_step(batch, step, 1)
        torch.cuda.synchronize()
        assert all(float(a) == float(b) for a, b in zip(g_loss + d_loss, g1 + d1))
        assert torch.equal(m.engine.G.params, m1.engine.G.params) and torch.equal(m.engine.D.params, m1.engine.D.params)
        assert np.isfinite(ref[0]).all()
    finally:
        comm.destroy()


def test_c_abi_collective_world1():
    """include/p2pgan.h p2p_comm_*: RCCL reached through the C ABI alone (a host without PyTorch binds these); on the one-GPU
    test box a world of one rank: SUM all-reduce = identity, stream-ordered."""
    import ctypes as C
    from palette_and_histo_gan_amd import _lib as L
    ident = (C.c_char * 128)()
    L.call("p2p_comm_unique_id", C.cast(ident, C.c_void_p))
    comm = C.c_void_p()
    L.call("p2p_comm_init", C.cast(ident, C.c_void_p), 0, 1, C.byref(comm))
    try:
        x = torch.arange(1 << 20, dtype=torch.float32, device="cuda:0")
        want = x.clone()
        st = torch.cuda.current_stream()
        L.call("p2p_comm_allreduce_sum", comm, C.c_void_p(x.data_ptr()), x.numel(), C.c_void_p(st.cuda_stream))
        torch.cuda.synchronize()
        assert torch.equal(x, want)
    finally:
        L.call("p2p_comm_destroy", comm)
