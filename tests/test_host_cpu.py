"""Not-gpu tests: the C-ABI library exports every symbol include/p2pgan.h declares, host logic (datasets, sharding,
work accounting, layouts), and the oracle against the committed golden vectors."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import reference_graph as rg
from palette_and_histo_gan_amd import _lib as L
from palette_and_histo_gan_amd import dataset_utils as D
from palette_and_histo_gan_amd import engine as E
from palette_and_histo_gan_amd import flops as FL
from palette_and_histo_gan_amd import parallel as PAR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "oracle_vectors.npz")


def header_symbols():
    text = open(os.path.join(ROOT, "include", "p2pgan.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(p2p_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_symbol_of_the_header():
    if not os.path.exists(L.LIB_PATH):
        from palette_and_histo_gan_amd import build
        build.build_library(verbose=False)
    lib = ctypes.CDLL(L.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/p2pgan.h but not exported"
    # and the ctypes binding covers the same set
    assert sorted(L.exported_symbols()) == syms
    L.lib()
    assert L.lib().p2p_version() >= 1
    assert L.lib().p2p_wgemm_workspace_bytes(2, 4, 4, 64, 128, 4) == 4 * 16 * 64 * 128 * 4


def test_parameter_layout_matches_reference_counts():
    g = E.generator_param_shapes(4, 4)
    d = E.discriminator_param_shapes(4)
    assert sum(int(np.prod(s)) for s in g.values()) == 29_307_844          # experiments.ipynb:198
    assert sum(int(np.prod(s)) for s in d.values()) == 9_217               # experiments.ipynb:199
    assert sum(int(np.prod(s)) for s in E.generator_param_shapes(1, 256).values()) == 29_437_888
    assert sum(int(np.prod(s)) for s in E.discriminator_param_shapes(1).values()) == 3_073
    assert list(g) == list(rg.generator_param_shapes(4, 4)) and dict(g) == dict(rg.generator_param_shapes(4, 4))
    assert E.pad8(36) == 40 and E.pad8(4) == 8 and E.up32(36) == 64 and E.up32(1) == 32


def test_algorithmic_work_matches_survey():
    m = FL.layer_macs(64, 4, 4)
    g_fwd = sum(v for k, v in m.items() if not k.startswith("D."))
    assert g_fwd == 441_450_496
    assert FL.train_step_flops_per_image(64) == 2 * 1_369_440_256
    assert FL.train_step_flops_per_image(128) == 2 * 5_477_761_024
    assert FL.train_step_flops_per_image(64, 1, 256, indexed=True) == 2 * 2_961_178_624


def test_dataset_follows_the_tf_data_slice_used_by_the_loop():
    ds = D.synthetic_rgba_ds(10, batch_size=4, img_size=64)
    batches = list(ds)
    assert [len(b[0]) for b in batches] == [4, 4, 2]                         # batch() without drop_remainder
    src, tgt = batches[0]
    assert src.dtype == np.float32 and src.shape == (4, 64, 64, 4) and src.min() == -1.0 and src.max() <= 1.0
    assert set(np.unique(src[..., 3])) <= {-1.0, 1.0}                        # alpha in {0,255}
    steps = list(ds.repeat().take(7).enumerate())
    assert [s for s, _ in steps] == list(range(7)) and len(steps[3][1][0]) == 4
    ex = list(ds.unbatch().take(3).batch(1).as_numpy_iterator())
    assert len(ex) == 3 and ex[0][0].shape == (1, 64, 64, 4)
    ids = D.synthetic_indexed_ds(5, batch_size=4)
    s, t, pal = next(iter(ids))
    assert s.dtype == np.int32 and s.shape == (4, 64, 64, 1) and pal.shape == (4, 256, 4)
    assert tuple(pal[0, 255]) == (255, 0, 220, 255) and s.max() < 24


def test_shard_bounds_cover_ragged_batches():
    for gb, world in ((256, 8), (250, 8), (2, 2), (5, 4), (3, 8)):
        spans = [PAR.shard_bounds(gb, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == gb
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def _check(tag, ref, gold):
    np.testing.assert_allclose(np.array(ref["g_loss"]), gold[f"{tag}.g_loss"], rtol=1e-12)
    np.testing.assert_allclose(np.array(ref["d_loss"]), gold[f"{tag}.d_loss"], rtol=1e-12)
    for pre, grads in (("G", ref["g_grads"]), ("D", ref["d_grads"])):
        for k, g in grads.items():
            a = g.numpy().reshape(-1)
            np.testing.assert_allclose(a.sum(), gold[f"{tag}.{pre}.{k}.sum"], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(np.abs(a).sum(), gold[f"{tag}.{pre}.{k}.abssum"], rtol=1e-9, atol=1e-12)


def test_oracle_reproduces_golden_vectors():
    """Pins the oracle's semantics to the committed vectors (tests/golden/make_golden.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    gold = np.load(GOLD)
    F64 = torch.float64
    Gp, Dp, src, tgt, masks = mg.rgba_case(101, 100.0, None)
    ref = rg.train_step_rgba(Gp, Dp, torch.tensor(src, dtype=F64), torch.tensor(tgt, dtype=F64),
                             [torch.tensor(m, dtype=F64) for m in masks], lambda_l1=100.0)
    _check("baseline", ref, gold)
    # histogram of the stored small image and its Hellinger gradient
    ft = torch.tensor(gold["hist8.fake"], dtype=F64, requires_grad=True)
    hr = rg.rgbuv_histogram(torch.tensor(gold["hist8.real"], dtype=F64))
    np.testing.assert_allclose(hr.numpy(), gold["hist8.hist_real"], rtol=1e-5, atol=1e-12)
    loss = rg.hellinger_loss(hr, rg.rgbuv_histogram(ft))
    loss.backward()
    np.testing.assert_allclose(float(loss), float(gold["hist8.loss"]), rtol=1e-12)
    np.testing.assert_allclose(ft.grad.numpy(), gold["hist8.dfake"], rtol=1e-9, atol=1e-15)
    assert np.array_equal(torch.argmax(torch.tensor(gold["argmax.probs"]), -1).numpy().astype(np.int32), gold["argmax.index"])


def test_product_path_has_no_cpu_fallback():
    """Constructing an engine without a GPU must fail (no silent CPU path), and a missing library must raise."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Exception):
        E.Pix2PixEngine(4, 4, "tanh", 64, L.BF16, device="cuda:0")
    for mod in ("engine.py", "pix2pix_model.py", "networks.py", "side2side_model.py", "parallel.py", "_lib.py"):
        text = open(os.path.join(ROOT, "palette_and_histo_gan_amd", mod)).read()
        assert "import oracle" not in text and "from oracle" not in text, mod


def test_replay_table_covers_the_int_entry_points_and_replays_host_calls():
    """p2p_replay (include/p2pgan.h "step replay"): every int-returning entry point has a thunk whose argument count equals the
    binding's; a recorded list is packed into p2p_replay_call records and issued by ONE call.  Exercised here with the one entry
    point that needs no GPU (p2p_png_unfilter): by-value ints, pointers, a pointer slot that is re-read at every replay (the
    mechanism behind the batch pointers), and the error path naming the failing call."""
    lib = L.lib()
    for name, argtypes in L.SIGNATURES.items():
        fn = lib.p2p_replay_fn_index(name.encode())
        assert fn >= 0, name
        assert lib.p2p_replay_fn_nargs(fn) == len(argtypes) <= E.ReplayCall.MAX_ARGS, name
    assert lib.p2p_replay_fn_index(b"p2p_last_error") == -1 and lib.p2p_replay_fn_index(b"p2p_replay") == -1
    assert ctypes.sizeof(E.ReplayCall) == 16 + 8 * 24
    h, rb, bpp = 3, 8, 4
    rng = np.random.default_rng(0)
    rows = rng.integers(0, 256, size=(h, rb), dtype=np.uint8)
    filt = np.concatenate([np.array([[1], [2], [0]], np.uint8), rows], axis=1).copy()      # Sub, Up, None
    want = np.zeros((h, rb), np.uint8)
    assert lib.p2p_png_unfilter(filt.ctypes.data, h, rb, bpp, want.ctypes.data) == 0
    out_a, out_b = np.zeros((h, rb), np.uint8), np.zeros((h, rb), np.uint8)
    slot = ctypes.c_void_p(out_a.ctypes.data)
    rec = [("p2p_png_unfilter", (ctypes.c_void_p(filt.ctypes.data), h, rb, bpp, slot)),
           ("p2p_png_unfilter", (filt.ctypes.data, h, rb, ctypes.c_int(bpp), slot))]
    arr, n = E.pack_replay(rec)
    assert n == 2 and arr[0].ind64 == 0b10001 and arr[0].ind32 == 0 and arr[1].ind64 == 0b10000 and arr[1].ind32 == 0b01000
    assert lib.p2p_replay(arr, n) == 0 and np.array_equal(out_a, want) and not out_b.any()
    slot.value = out_b.ctypes.data                  # the next replay writes where the slot points NOW
    assert lib.p2p_replay(arr, n) == 0 and np.array_equal(out_b, want)
    bad, _ = E.pack_replay([("p2p_png_unfilter", (filt.ctypes.data, h, rb, bpp, slot)), ("p2p_png_unfilter", (None, h, rb, bpp, slot))])
    assert lib.p2p_replay(bad, 2) != 0
    msg = lib.p2p_last_error().decode()
    assert "call 1 of 2" in msg and "p2p_png_unfilter" in msg, msg
    with pytest.raises(L.P2PError):
        E.pack_replay([("p2p_png_unfilter", (1, 2, 3))])
    with pytest.raises(TypeError):
        E.pack_replay([("p2p_png_unfilter", ("text", h, rb, bpp, slot))])
