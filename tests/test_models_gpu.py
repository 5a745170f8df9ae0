"""-m gpu: the reference-API model classes driven like experiments.ipynb drives them, the golden vectors through the
engine, and the 2-rank data-parallel step (gloo transport, both ranks on the one GPU of the test box)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import reference_graph as rg
from palette_and_histo_gan_amd import _lib as L
from palette_and_histo_gan_amd import dataset_utils as D
from palette_and_histo_gan_amd import engine as E
from palette_and_histo_gan_amd import parallel as PAR
from palette_and_histo_gan_amd import pix2pix_model as M

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F64 = torch.float64


def test_fit_loop_runs_like_the_notebook(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    train = D.synthetic_rgba_ds(10, batch_size=4)          # 4 + 4 + 2: the ragged tail batch of the reference
    test = D.synthetic_rgba_ds(6, batch_size=4, seed=3)
    model = M.Pix2PixModel(train, test, "front2right", "pix2pix-test", lambda_l1=100.0)
    assert model.generator.count_params() == 29_307_844 and model.discriminator.count_params() == 9_217
    w0 = model.generator.trainable_variables[0].clone()
    model.fit(7, 3, callbacks=["show_discriminator_output", "evaluate_fid", "evaluate_l1"])
    assert model.engine.G.t == 7 and not torch.equal(w0, model.generator.trainable_variables[0])
    assert len(model.checkpoint_manager.saved) == 1 and os.path.exists(model.checkpoint_manager.saved[0])
    rows = [r for r in open(model.summary_writer.path)]
    assert any("generator/l1_loss" in r for r in rows) and any("l1-evaluation/test" in r for r in rows)
    # the same scalars as a TensorBoard event file (what the reference's tf.summary calls leave behind)
    from palette_and_histo_gan_amd import tb_events
    ev = list(tb_events.read_events(model.summary_writer.events.path))
    assert sum(1 for _, tag, _ in ev if tag == "generator/l1_loss") == 7 and all(np.isfinite(v) for _, _, v in ev if isinstance(v, float))
    # F4 (side2side_model.py:58-61,86-93): the custom-scalar layout at step 0, and at step 0 and every update_steps the preview sheet
    # as <log folder>/step_NNNNNN.png AND as an image summary whose tag is that path, at step (step + 1) // update_steps
    from palette_and_histo_gan_amd import png
    folder = os.path.dirname(model.summary_writer.path)
    assert sorted(f for f in os.listdir(folder) if f.endswith(".png")) == ["step_000001.png", "step_000003.png", "step_000006.png"]
    assert ev[0][0] == 0 and ev[0][1] == "custom_scalars__config__" and b"l1\\-evaluation" in ev[0][2][0]
    images = [(st, tag, v) for st, tag, v in ev if isinstance(v, list) and tag.endswith(".png")]
    assert [st for st, _, _ in images] == [0, 1, 2] and images[1][1].endswith(os.sep.join([model.now_string, "step_000003.png"]))
    for st, tag, (w, h, data) in images:
        sheet = png.decode_png(data)
        assert (int(w), int(h)) == (sheet.shape[1], sheet.shape[0]) == (3 * 64 + 4, 6 * 64 + 10)      # 6 examples x (Input | Target | Generated)
        assert np.array_equal(sheet, png.read_png(tag))
    # the reference's attribute surface (pix2pix_model.py:12-36)
    assert model.loss_object is not None and model.checkpoint.generator is model.generator
    assert model.generator_optimizer.learning_rate == 0.0002 and model.generator_optimizer.beta_1 == 0.5
    assert model.generator_optimizer.iterations == 7
    adv = model.loss_object(torch.ones(3), torch.zeros(3))
    assert abs(float(adv) - np.log(2.0)) < 1e-6
    g_loss, d_loss = model.train_step(next(iter(train)), 7, 3)
    assert len(g_loss) == 3 and len(d_loss) == 3 and all(torch.isfinite(x) for x in g_loss + d_loss)
    fake = model.generate(next(iter(test)))
    assert tuple(fake.shape) == (4, 64, 64, 4) and float(fake.abs().max()) <= 1.0
    logits = model.discriminator([next(iter(test))[1], next(iter(test))[0]], training=True)
    assert tuple(logits.shape) == (4, 32, 32, 1)
    # the notebook's evaluation calls (side2side_model.py:202-239): sheets Input | Target | Generated as PNG files
    from palette_and_histo_gan_amd import png
    folder = model.generate_images_from_dataset("test", num_images=3)
    files = sorted(os.listdir(folder))
    assert files == ["0.png", "1.png", "2.png"]
    sheet = png.read_png(os.path.join(folder, "1.png"))
    assert sheet.shape == (64, 3 * 64 + 4, 4)
    first = next(iter(test.unbatch().take(2).batch(1)))                       # sample 0 of the test set
    want = np.round((np.asarray(first[0][0], np.float32) * 0.5 + 0.5) * 255).astype(np.uint8)
    assert (png.read_png(os.path.join(folder, "0.png"))[:, :64] == want).all()
    patches = model.show_discriminated_images("test", 2)
    assert len(patches) == 2 and patches[0]["real"].shape == (32, 32) and 0.0 < patches[0]["fake_mean"] < 1.0


def test_scalar_log_does_not_synchronise_and_hooks_fail_loudly(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    train = D.synthetic_rgba_ds(8, batch_size=4)
    model = M.Pix2PixModel(train, train, "front2right", "pix2pix-hooks-test", lambda_l1=100.0)
    model.fit(1, 1)
    model.train_step(next(iter(train)), 1, 1)
    pend = model.summary_writer._rows
    assert len(pend) == 6 and all(isinstance(r[1], torch.Tensor) and r[1].is_cuda for r in pend)      # still on the device
    model.summary_writer.flush()
    assert not model.summary_writer._rows

    class Custom(M.Pix2PixIndexedModel):      # the palette-index step is fused around its softmax head: ITS hooks cannot be replaced
        def generator_loss(self, fake_predicted, fake_image, real_image):
            return super().generator_loss(fake_predicted, fake_image, real_image)

    ids = D.synthetic_indexed_ds(4, batch_size=4)
    c = Custom(ids, ids, "front2right", "pix2pix-hooks-test")
    with pytest.raises(NotImplementedError, match="fused"):
        c.train_step(next(iter(ids)), 0, 1)

    class OneChannel(M.Pix2PixIndexedModel):        # the builders take the reference's arguments (networks.py:39,53)
        def create_generator(self):
            return M.UnetGenerator(1, 256, "softmax")

    o = OneChannel(D.synthetic_indexed_ds(4, batch_size=4), None, "front2right", "pix2pix-hooks-test")
    assert o.generator.count_params() == 29_437_888 and o.generator.input_channels == 1
    # Keras-order weight list round trip through the handle
    w = o.discriminator.get_weights()
    o.discriminator.set_weights(list(w.values()))
    with pytest.raises(ValueError):
        o.discriminator.set_weights(list(w.values())[:-1])


@pytest.mark.parametrize("kind", ["baseline", "histogram", "indexed"])
def test_checkpoint_restore_resumes_bit_exactly(tmp_path, monkeypatch, kind):
    """save -> two more steps -> restore -> the same two steps: weights, Adam moments / step counts and the device
    dropout counter all come back, so the replayed steps reproduce the first run bit for bit (f32 mode; every reduction of
    the step has a fixed order: no float atomics in the loss, histogram or softmax kernels)."""
    monkeypatch.chdir(tmp_path)
    if kind == "indexed":
        train = D.synthetic_indexed_ds(8, batch_size=4)
        model = M.Pix2PixIndexedModel(train, train, "front2right", "pix2pix-ckpt-test", lambda_segmentation=0.01, dtype="f32")
    elif kind == "histogram":
        train = D.synthetic_rgba_ds(8, batch_size=4, palette_size=24)
        model = M.Pix2PixHistogramModel(train, train, "front2right", "pix2pix-ckpt-test", 30.0, 1.0, dtype="f32")
    else:
        train = D.synthetic_rgba_ds(8, batch_size=4)
        model = M.Pix2PixModel(train, train, "front2right", "pix2pix-ckpt-test", lambda_l1=100.0, dtype="f32")
    batches = list(iter(train))
    model.fit(2, 1)
    path = model.checkpoint_manager.save()
    run_a = [model.train_step(batches[i % 2], 2 + i, 1) for i in range(2)]
    w_a = model.engine.G.params.clone()
    assert model.checkpoint_manager.restore(path) == path and model.engine.G.t == 2
    run_b = [model.train_step(batches[i % 2], 2 + i, 1) for i in range(2)]
    for (ga, da), (gb, db) in zip(run_a, run_b):
        assert all(float(x) == float(y) for x, y in zip(ga + da, gb + db))
    assert torch.equal(w_a, model.engine.G.params)


def test_histogram_and_indexed_models_train(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    train, test = D.synthetic_rgba_ds(8, batch_size=4, palette_size=24), D.synthetic_rgba_ds(4, batch_size=4, seed=5)
    hm = M.Pix2PixHistogramModel(train, test, "front2right", "pix2pix-hist-test", lambda_l1=30.0, lambda_histogram=1.0)
    hm.fit(3, 2)
    g_loss, d_loss = hm.train_step(next(iter(train)), 3, 2)
    assert len(g_loss) == 4 and float(g_loss[3]) > 0 and all(torch.isfinite(x) for x in g_loss)
    itrain, itest = D.synthetic_indexed_ds(8, batch_size=4), D.synthetic_indexed_ds(4, batch_size=4, seed=6)
    im = M.Pix2PixIndexedModel(itrain, itest, "front2right", "pix2pix-idx-test", lambda_segmentation=0.01)
    assert im.generator.count_params() == 29_437_888 and im.discriminator.count_params() == 3_073
    im.fit(3, 2)
    g_loss, d_loss = im.train_step(next(iter(itrain)), 3, 2)
    assert len(g_loss) == 4 and all(torch.isfinite(x) for x in g_loss + d_loss)
    idx, probs = im.generate_with_probs(next(iter(itest)))
    assert idx.dtype == torch.int32 and tuple(idx.shape) == (4, 64, 64, 1) and tuple(probs.shape) == (4, 64, 64, 256)
    assert torch.equal(idx[..., 0].long(), torch.argmax(probs, -1))          # palette-index argmax, bit-exact
    np.testing.assert_allclose(probs.sum(-1).cpu().numpy(), 1.0, rtol=1e-5)


def test_host_batches_upload_on_a_copy_stream_and_train_identically(tmp_path, monkeypatch):
    """A batch source that yields numpy arrays (a tf.data pipeline of the reference through Dataset.from_batches) is uploaded by
    dataset_utils.upload_async on a stream of its own, beside the previous step's kernels; six steps (the replayed ones included) must
    leave exactly the weights of the model that is fed device tensors."""
    monkeypatch.chdir(tmp_path)
    batches = list(D.synthetic_rgba_ds(24, batch_size=4, palette_size=24, seed=31))
    assert isinstance(batches[0][0], np.ndarray)
    runs = []
    for host in (True, False):
        m = M.Pix2PixModel(None, None, "front2right", "upload-test", lambda_l1=100.0, seed=7)
        for t, b in enumerate(batches):
            bb = b if host else tuple(torch.as_tensor(x).cuda() for x in b)
            g_loss, d_loss = m.train_step(bb, t, 1)
        torch.cuda.synchronize()
        runs.append((m, [float(x) for x in g_loss + d_loss]))
    (a, la), (b, lb) = runs
    assert la == lb
    assert torch.equal(a.engine.G.params, b.engine.G.params) and torch.equal(a.engine.D.params, b.engine.D.params)
    assert len(a.engine._replays) == 1 and len(D._COPY_STREAMS) == 1


def _params(seed):
    rng = np.random.default_rng(seed)
    Gp = rg.perturb_affine(rg.init_params(rg.generator_param_shapes(4, 4), rng, F64), rng)
    Dp = rg.perturb_affine(rg.init_params(rg.discriminator_param_shapes(4), rng, F64), rng)
    return rng, Gp, Dp


def test_overridden_loss_hooks_run_through_autograd_across_the_hooks():
    """SURVEY.md B1 / VERDICT r04 missing-3: generator_loss / discriminator_loss are hooks subclasses rely on (the reference's own
    Pix2PixHistogramModel overrides one, pix2pix_model.py:242-250).  A subclass that overrides them gets engine.train_step_rgba_hooked:
    kernels forward, the hooks on torch tensors with autograd, their gradients into the backward kernels.
    (a) hooks that restate the reference's formulas must reproduce the FUSED step: losses 1e-6, every gradient tensor 1e-5 of its
        max-norm (f32 mode, injected dropout masks);
    (b) hooks of a different loss family (least-squares GAN + a weighted L2 image term) against the float64 oracle graph
        differentiated by torch autograd: losses 1e-5, gradients 1e-4 of max-norm."""
    B, S = 2, 64
    rng, Gp, Dp = _params(71)
    src, tgt = rg.synthetic_rgba_batch(rng, B, S, palette_size=24)
    masks = [rng.integers(0, 2, size=s).astype(np.uint8) for s in rg.dropout_mask_shapes(B, S)]
    to_np = lambda p: {k: v.numpy() for k, v in p.items()}      # noqa: E731

    def engine():
        eng = E.Pix2PixEngine(4, 4, "tanh", S, L.F32)
        eng.set_params(to_np(Gp), to_np(Dp))
        return eng

    bce = torch.nn.functional.binary_cross_entropy_with_logits

    def gen_ref(fp, fake, real):
        adv = bce(fp, torch.ones_like(fp))
        l1 = (real - fake).abs().mean()
        return adv + 100.0 * l1, adv, l1

    def disc_ref(rp, fp):
        r, f = bce(rp, torch.ones_like(rp)), bce(fp, torch.zeros_like(fp))
        return r + f, r, f

    fused, hooked = engine(), engine()
    out_f = fused.train_step_rgba(src, tgt, 100.0, masks=masks, apply_update=False).cpu().numpy()
    out_h = hooked.train_step_rgba_hooked(src, tgt, gen_ref, disc_ref, masks=masks, apply_update=False).cpu().numpy()
    for i in (0, 1, 2, 4, 5, 6):
        assert abs(out_h[i] - out_f[i]) <= 1e-6 * abs(out_f[i]), (i, out_h[i], out_f[i])
    for a, b in ((hooked.G, fused.G), (hooked.D, fused.D)):
        ga, gb = a.export(a.grads), b.export(b.grads)
        for k in ga:
            assert np.abs(ga[k] - gb[k]).max() <= 1e-5 * np.abs(gb[k]).max() + 1e-12, k

    # (b) least-squares GAN hooks against the oracle graph under autograd
    def gen_ls(fp, fake, real):
        adv = ((fp - 1.0) ** 2).mean()
        l2 = ((real - fake) ** 2).mean()
        return adv + 25.0 * l2, adv, l2

    def disc_ls(rp, fp):
        r, f = ((rp - 1.0) ** 2).mean(), (fp ** 2).mean()
        return 0.5 * (r + f), r, f

    eng = engine()
    out = eng.train_step_rgba_hooked(src, tgt, gen_ls, disc_ls, masks=masks, apply_update=False).cpu().numpy()
    Gl = {k: v.clone().requires_grad_(True) for k, v in Gp.items()}
    Dl = {k: v.clone().requires_grad_(True) for k, v in Dp.items()}
    s64, t64 = torch.tensor(src, dtype=F64), torch.tensor(tgt, dtype=F64)
    fake = rg.unet_generator(Gl, s64, [torch.tensor(m, dtype=F64) for m in masks], "tanh")
    rp, fp = rg.patch_discriminator(Dl, t64, s64), rg.patch_discriminator(Dl, fake, s64)
    g = gen_ls(fp, fake, t64)
    d = disc_ls(rp, fp)
    g_grads = torch.autograd.grad(g[0], list(Gl.values()), retain_graph=True, allow_unused=True)
    d_grads = torch.autograd.grad(d[0], list(Dl.values()), allow_unused=True)
    want = [float(g[0].detach()), float(g[1].detach()), float(g[2].detach()), 0.0, float(d[0].detach()), float(d[1].detach()), float(d[2].detach())]
    for i in (0, 1, 2, 4, 5, 6):
        assert abs(out[i] - want[i]) <= 1e-5 * abs(want[i]), (i, out[i], want[i])
    for store, names, grads in ((eng.G, list(Gl), g_grads), (eng.D, list(Dl), d_grads)):
        got = store.export(store.grads)
        for k, gr in zip(names, grads):
            ref = np.zeros_like(got[k]) if gr is None else gr.numpy()
            assert np.abs(got[k] - ref).max() <= 1e-4 * np.abs(ref).max() + 1e-12, (k, np.abs(got[k] - ref).max(), np.abs(ref).max())


def test_a_subclass_with_its_own_losses_trains_through_fit(tmp_path, monkeypatch):
    """the class-level route: a subclass of Pix2PixModel with least-squares losses goes through fit() (bf16, device dropout RNG), its
    hooks are called every step with differentiable tensors, the loss goes down and the reference's scalars are logged"""
    monkeypatch.chdir(tmp_path)
    calls = []

    class LsGan(M.Pix2PixModel):
        def generator_loss(self, fake_predicted, fake_image, real_image):
            calls.append((fake_predicted.requires_grad, fake_image.requires_grad, tuple(fake_image.shape)))
            adv = ((fake_predicted - 1.0) ** 2).mean()
            l1 = (real_image - fake_image).abs().mean()
            return adv + self.lambda_l1 * l1, adv, l1

        def discriminator_loss(self, real_predicted, fake_predicted):
            r, f = ((real_predicted - 1.0) ** 2).mean(), (fake_predicted ** 2).mean()
            return 0.5 * (r + f), r, f

    train = D.synthetic_rgba_ds(8, batch_size=4, palette_size=24)
    m = LsGan(train, train, "front2right", "pix2pix-lsgan-test", lambda_l1=100.0)
    w0 = m.generator.trainable_variables[0].clone()
    first = [float(x) for x in m.train_step(next(iter(train)), 0, 1)[0]]
    m.fit(30, 10)
    last = [float(x) for x in m.train_step(next(iter(train)), 31, 10)[0]]
    assert len(calls) == 32 and all(c == (True, True, (4, 64, 64, 4)) for c in calls)
    assert not torch.equal(w0, m.generator.trainable_variables[0]) and m.engine.G.t == 32
    assert last[2] < 0.6 * first[2], (first, last)          # the L1 term fell
    rows = [r for r in open(m.summary_writer.path)]
    assert any("generator/l1_loss" in r for r in rows) and any("discriminator/total_loss" in r for r in rows)


def test_engine_reproduces_golden_vectors():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "oracle_vectors.npz"))
    for tag, seed, l1, lh in (("baseline", 101, 100.0, None), ("histogram", 102, 30.0, 1.0)):
        Gp, Dp, src, tgt, masks = mg.rgba_case(seed, l1, lh)
        eng = E.Pix2PixEngine(4, 4, "tanh", 64, L.F32)
        eng.set_params({k: v.numpy() for k, v in Gp.items()}, {k: v.numpy() for k, v in Dp.items()})
        out = eng.train_step_rgba(src, tgt, l1, lambda_hist=lh, masks=masks, apply_update=False).cpu().numpy()
        g, d = gold[f"{tag}.g_loss"], gold[f"{tag}.d_loss"]
        want = [g[0], g[1], g[2], g[3] if lh else 0.0, d[0], d[1], d[2]]
        for i in range(7):
            assert abs(out[i] - want[i]) <= 1e-4 * abs(want[i]) + 1e-12, (tag, i, out[i], want[i])
        grads = eng.G.export(eng.G.grads)
        report, failed = {}, []
        for k, a in grads.items():
            a = a.reshape(-1).astype(np.float64)
            samples = gold[f"{tag}.G.{k}.samples"]
            got = a[mg.sample_positions(k, a.size)]
            scale = gold[f"{tag}.G.{k}.abssum"] / a.size + 1e-30
            # PER-TENSOR bounds: 2 % of the tensor's scale / 1e-3 of its absolute sum, or 1.5 x what the ORACLE evaluated
            # in float32 does to THIS tensor where that is more (histogram case: near-black fake pixels whose gradient is
            # 1 / (x + 1e-6)-steep; the f32 oracle is 2.6 % off on up3.kernel's samples and 1.3e-3 on up4.gamma's sum).  A regression
            # on a well-conditioned tensor can no longer hide under another tensor's ill-conditioning.
            dev = np.abs(got - samples).max() / max(np.abs(samples).max(), scale)
            dev_sum = abs(np.abs(a).sum() - gold[f"{tag}.G.{k}.abssum"]) / (gold[f"{tag}.G.{k}.abssum"] + 1e-30)
            f32dev, f32dev_sum = float(gold[f"{tag}.G.{k}.f32dev"]), float(gold[f"{tag}.G.{k}.f32dev_abssum"])
            tol = max(2e-2, 1.5 * f32dev)
            # absolute sums: the yardstick is the LAYER's worst tensor (kernel, gamma and beta gradients of a layer are sums over the
            # same gradient field d(raw), so one flipped pixel moves all three; measured r05: down2.beta 1.35e-3 with the f32 oracle
            # at 6.1e-4 on beta and 1.06e-3 on gamma -- profiles/r05_golden_engine_dev.json holds every tensor's figures)
            layer_sums = [float(gold[f"{tag}.G.{k2}.f32dev_abssum"]) for k2 in grads if k2.split(".")[0] == k.split(".")[0]]
            layer_dev_sum = max([x for x in layer_sums if np.isfinite(x)] + [0.0])
            tol_sum = max(1e-3, 1.5 * layer_dev_sum)
            report[k] = {"dev": float(dev), "f32_oracle_dev": f32dev, "dev_abssum": float(dev_sum), "f32_oracle_dev_abssum": f32dev_sum}
            if not (dev < tol and (dev_sum <= tol_sum or gold[f"{tag}.G.{k}.abssum"] < 1e-12)):
                failed.append((k, dev, tol, dev_sum, tol_sum))
        out_dir = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, f"golden_engine_dev_{tag}.json"), "w") as f:       # the engine's measured per-tensor deviation
            json.dump(report, f, indent=1)
        assert not failed, (tag, failed)
    # argmax fixture, bit-exact
    p = torch.as_tensor(gold["argmax.probs"]).to("cuda:0")
    out = torch.empty(p.shape[0], dtype=torch.int32, device="cuda:0")
    import ctypes as C
    L.call("p2p_argmax_lastdim", C.c_void_p(p.data_ptr()), p.shape[0], p.shape[1], C.c_void_p(out.data_ptr()),
           C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert np.array_equal(out.cpu().numpy(), gold["argmax.index"])


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    comm = PAR.DataParallel("cuda:0", backend="gloo")
    rng = np.random.default_rng(51)
    Gp = rg.perturb_affine(rg.init_params(rg.generator_param_shapes(4, 4), rng, F64), rng)
    Dp = rg.perturb_affine(rg.init_params(rg.discriminator_param_shapes(4), rng, F64), rng)
    B = 4
    src, tgt = rg.synthetic_rgba_batch(rng, B, 64, palette_size=24)
    masks = [rng.integers(0, 2, size=s).astype(np.uint8) for s in rg.dropout_mask_shapes(B, 64)]
    lo, hi = PAR.shard_bounds(B, world, rank)
    eng = E.Pix2PixEngine(4, 4, "tanh", 64, L.F32, device="cuda:0")
    eng.set_params({k: v.numpy() for k, v in Gp.items()}, {k: v.numpy() for k, v in Dp.items()})
    out = eng.train_step_rgba(src[lo:hi], tgt[lo:hi], 30.0, lambda_hist=1.0, masks=[m[lo:hi] for m in masks],
                              global_batch=B, dp=comm)
    torch.cuda.synchronize()
    if rank == 0:
        q.put((out.cpu().numpy(), eng.G.grads.cpu().numpy(), eng.D.grads.cpu().numpy(), eng.G.params.cpu().numpy()))
    comm.barrier()
    comm.destroy()
    q.close()
    q.join_thread()
    os._exit(0)          # skip interpreter teardown of a process that shares the GPU with its sibling rank


@pytest.mark.timeout(900)
def test_two_rank_engine_step_equals_single_rank_global_batch():
    """SURVEY.md 8e equivalence test on the real kernels: 2 ranks x 2 images (histogram model: includes the Hellinger
    scalar exchange) == 1 rank x 4 images; f32, tolerance 1e-5 of each tensor's max-norm on gradients."""
    world, port = 2, 29641
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out2, g2, d2, p2 = q.get(timeout=800)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0, f"rank process exited with {p.exitcode}"
    rng = np.random.default_rng(51)
    Gp = rg.perturb_affine(rg.init_params(rg.generator_param_shapes(4, 4), rng, F64), rng)
    Dp = rg.perturb_affine(rg.init_params(rg.discriminator_param_shapes(4), rng, F64), rng)
    src, tgt = rg.synthetic_rgba_batch(rng, 4, 64, palette_size=24)
    masks = [rng.integers(0, 2, size=s).astype(np.uint8) for s in rg.dropout_mask_shapes(4, 64)]
    eng = E.Pix2PixEngine(4, 4, "tanh", 64, L.F32)
    eng.set_params({k: v.numpy() for k, v in Gp.items()}, {k: v.numpy() for k, v in Dp.items()})
    out1 = eng.train_step_rgba(src, tgt, 30.0, lambda_hist=1.0, masks=masks).cpu().numpy()
    np.testing.assert_allclose(out2, out1, rtol=2e-6)
    g1, d1 = eng.G.grads.cpu().numpy(), eng.D.grads.cpu().numpy()
    assert np.abs(g2 - g1).max() <= 1e-5 * np.abs(g1).max()
    assert np.abs(d2 - d1).max() <= 1e-5 * np.abs(d1).max()
    # same Adam update as the single-rank run (entries with rounding-level gradients may move by a few % of lr = 2e-4)
    assert np.abs(p2 - eng.G.params.cpu().numpy()).max() < 2e-5
