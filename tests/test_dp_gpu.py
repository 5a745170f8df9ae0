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


def _fit_worker(rank, world, port, q, tmp):
    """S2SModel.fit() under data parallelism: the sprite dataset produces this rank's rows only, rank 0 alone writes logs,
    previews and checkpoints, every rank ends with the same weights"""
    import faulthandler
    import sys
    faulthandler.dump_traceback_later(int(os.environ.get("P2P_TEST_DUMP_AFTER", "240")), exit=True, file=sys.stderr)
    os.chdir(tmp)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    try:
        comm = PAR.DataParallel("cuda:0", backend="gloo") if world > 1 else None
        train, test = D.load_rgba_ds(2, 3, augment=True, batch_size=4, train_sizes=[10], test_sizes=[3], device="cuda:0", seed=5)
        m = M.Pix2PixModel(train, test, "front2right", f"dp-fit-w{world}", 100.0, dtype="f32", data_parallel=comm, seed=5)
        sizes = []
        orig = m.engine.train_step_rgba

        def spy(src, *a, **kw):
            sizes.append((int(src.shape[0]), kw.get("global_batch"), kw.get("batch_offset")))
            return orig(src, *a, **kw)
        m.engine.train_step_rgba = spy
        m.fit(5, 2, callbacks=["evaluate_l1"])          # 10 sprites, batch 4: global batches 4, 4, 2 (ragged), 4, 4
        torch.cuda.synchronize()
        # (round 5) the sharded steps of the repeated batch shape are REPLAYED: recorded C-ABI segments with the collectives in between
        replayed = [sum(1 for sg in segs if callable(sg)) for segs, _ in m.engine._replays.values()]
        q.put(("fit", rank, world, m.engine.G.params.cpu().numpy().copy(), m.engine.D.params.cpu().numpy().copy(), sizes,
               m.summary_writer is not None, list(m.checkpoint_manager.saved), replayed))
        if comm is not None:
            comm.barrier()
            comm.destroy()
    except BaseException:
        import traceback
        q.put(("error", rank, traceback.format_exc()))
        q.close()
        q.join_thread()
        os._exit(1)
    q.close()
    q.join_thread()
    os._exit(0)


@pytest.mark.timeout(1200)
def test_two_rank_fit_equals_one_rank_fit(tmp_path):
    """sharded batch production + rank-aware fit().  Datasets written once, then a 1-rank fit and a
    2-rank fit (gloo transport, both ranks on the test box's GPU) of the same 5 steps over a ragged epoch."""
    from tests import sprite_fixtures as F
    F.write_dataset(str(tmp_path), 10, 3, directions=(2, 3))
    ctx = mp.get_context("spawn")
    results = {}
    for world, port in ((1, 29661), (2, 29662)):
        q = ctx.Queue()
        procs = [ctx.Process(target=_fit_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
        for p in procs:
            p.start()
        for _ in range(world):
            item = q.get(timeout=600)
            if item[0] == "error":
                for p in procs:
                    p.kill()
                pytest.fail(f"rank {item[1]} (world {world}) failed:\n{item[2]}")
            results[(world, item[1])] = item[3:]
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    g1, d1, sizes1, log1, ck1, rep1 = results[(1, 0)]
    ga, da, sizes_a, log_a, ck_a, rep_a = results[(2, 0)]
    gb, db, sizes_b, log_b, ck_b, rep_b = results[(2, 1)]
    # the steps of the recurring shape were replayed on every rank: one recording, without host-side operations on one GPU, with the
    # bucket all-reduces, the waits and the tail collective between its segments under data parallelism
    assert rep1 == [0] and len(rep_a) == len(rep_b) == 1 and rep_a[0] == rep_b[0] >= 3, (rep1, rep_a, rep_b)
    # each rank was handed ITS rows of every global batch, with the global size and its offset
    assert [s[0] for s in sizes1] == [4, 4, 2, 4, 4]
    assert sizes_a == [(2, 4, 0), (2, 4, 0), (1, 2, 0), (2, 4, 0), (2, 4, 0)]
    assert sizes_b == [(2, 4, 2), (2, 4, 2), (1, 2, 1), (2, 4, 2), (2, 4, 2)]
    # rank 0 alone logs and checkpoints; the checkpoint exists once
    assert log_a and not log_b and len(ck_a) == 1 and ck_b == []
    ck_dir = os.path.join(str(tmp_path), os.path.dirname(ck_a[0]))
    assert sorted(os.listdir(ck_dir)) == [os.path.basename(ck_a[0])]
    # both ranks hold the same weights, and they are the 1-rank run's (f32 summation order, five Adam steps)
    assert np.array_equal(ga, gb) and np.array_equal(da, db)
    # f32 mode is batch-invariant on the data path (engine.batch_invariant: same kernel variant, K split and statistics algorithm
    # whatever the batch), so the two runs see bit-identical activations and data gradients and differ only in the order of the
    # weight-gradient sums over the batch: a rounding-level gradient entry whose sign changes moves by up to 2 lr = 4e-4 in one
    # Adam step (measured 4.3e-4 on a handful of entries, r03)
    assert np.abs(ga - g1).max() < 6e-4 and np.abs(da - d1).max() < 6e-4, (np.abs(ga - g1).max(), np.abs(da - d1).max())
    def worst():      # per-tensor report for a failing run
        from palette_and_histo_gan_amd import engine as E_
        offs, out = 0, []
        for k, shp in E_.generator_param_shapes(4, 4).items():
            n = int(np.prod(shp))
            out.append((float(np.abs(ga[offs:offs + n] - g1[offs:offs + n]).mean()), k))
            offs += (n + 3) // 4 * 4
        return sorted(out, reverse=True)[:6]
    assert np.abs(ga - g1).mean() < 2e-6, (np.abs(ga - g1).mean(), worst())


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
        # five steps each: with the communicator (steps 3-5 REPLAY the recorded step, the collectives re-issued between the segments
        # of the call list -- round 5), with the communicator and replay off (every launch from Python), and without a communicator
        runs = []
        for dp, replay in ((comm, True), (comm, False), (None, True)):
            m = _make("baseline", None, dp)
            m.engine.replay_enabled = replay
            for step in range(5):
                g_loss, d_loss = m.train_step(batch, step, 1)
            torch.cuda.synchronize()
            runs.append((m, [float(x) for x in g_loss + d_loss]))
        (m, la), (m2, lb), (m1, lc) = runs
        assert la == lb == lc
        for other in (m2, m1):
            assert torch.equal(m.engine.G.params, other.engine.G.params) and torch.equal(m.engine.D.params, other.engine.D.params)
            assert torch.equal(m.engine.G.m, other.engine.G.m) and torch.equal(m.engine.G.v, other.engine.G.v)
        assert len(m.engine._replays) == 1 and len(m2.engine._replays) == 0 and len(m1.engine._replays) == 1
        segs = next(iter(m.engine._replays.values()))[0]
        assert sum(1 for sg in segs if callable(sg)) >= len(m.engine.G.buckets) + 1 and len(next(iter(m1.engine._replays.values()))[0]) == 1
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
