import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import reference_graph as rg
from palette_and_histo_gan_amd import _lib as L, engine as E
from tests.test_train_step_gpu import setup_case, to_np
from tests import gpu_util as U
F64 = torch.float64
B, S = 2, 64
rng, Gp, Dp, src, tgt, masks = setup_case(B, S, 21)
engs = {}
for mf in (False, True):
    eng = E.Pix2PixEngine(4, 4, "tanh", S, L.F32, use_mfma=mf)
    eng.set_params(to_np(Gp), to_np(Dp))
    eng.train_step_rgba(src, tgt, 100.0, masks=masks, apply_update=False)
    engs[mf] = eng
Pa, Pb = engs[False].plans[B], engs[True].plans[B]
def cmp(name, a, b):
    a, b = a.float().cpu().numpy(), b.float().cpu().numpy()
    sc = np.abs(b).max() + 1e-30
    print(f"{name:12s} scale {sc:9.3e} maxrel {np.abs(a-b).max()/sc:9.3e} nz_a {np.count_nonzero(a)} nz_b {np.count_nonzero(b)}")
for i in range(6, 0, -1):
    cmp(f"ru{i}", Pa["ru"][i].t, Pb["ru"][i].t)
    cmp(f"gc{i}", Pa["gc"][i].t, Pb["gc"][i].t)
    cmp(f"du{i}", Pa["du"][i].t, Pb["du"][i].t)
for i in range(6, 0, -1):
    cmp(f"ga{i}", Pa["ga"][i].t, Pb["ga"][i].t)
    cmp(f"dd{i}", Pa["dd"][i].t, Pb["dd"][i].t)
cmp("c5", Pa["c"][5].t, Pb["c"][5].t)
ga, gb = engs[False].G.export(engs[False].G.grads), engs[True].G.export(engs[True].G.grads)
d = np.abs(ga["up6.kernel"] - gb["up6.kernel"])
print("up6.kernel diff by tap:", d.reshape(16, -1).max(1))
print("by g:", d.max(axis=(0,1,3))[:8], "by d:", d.max(axis=(0,1,2))[:8])
