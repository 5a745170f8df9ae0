import os, sys, importlib.util
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from palette_and_histo_gan_amd import _lib as L, engine as E
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
gold = np.load(os.path.join(ROOT, "tests", "golden", "oracle_vectors.npz"))
for tag, seed, l1, lh in (("baseline", 101, 100.0, None), ("histogram", 102, 30.0, 1.0)):
    Gp, Dp, src, tgt, masks = mg.rgba_case(seed, l1, lh)
    eng = E.Pix2PixEngine(4, 4, "tanh", 64, L.F32)
    eng.set_params({k: v.numpy() for k, v in Gp.items()}, {k: v.numpy() for k, v in Dp.items()})
    out = eng.train_step_rgba(src, tgt, l1, lambda_hist=lh, masks=masks, apply_update=False).cpu().numpy()
    grads = eng.G.export(eng.G.grads)
    rows = []
    for k, a in grads.items():
        a = a.reshape(-1).astype(np.float64)
        samples = gold[f"{tag}.G.{k}.samples"]
        got = a[mg.sample_positions(k, a.size)]
        scale = gold[f"{tag}.G.{k}.abssum"] / a.size + 1e-30
        rows.append((np.abs(got - samples).max() / max(np.abs(samples).max(), scale), k))
    print(tag, sorted(rows, reverse=True)[:5])
