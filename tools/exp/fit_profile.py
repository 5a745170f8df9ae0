import cProfile, pstats, sys, os, time, io
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from palette_and_histo_gan_amd import dataset_utils as D, pix2pix_model as M
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import quality_run as Q
rng = np.random.default_rng(0)
os.makedirs("/tmp/fitprof", exist_ok=True); os.chdir("/tmp/fitprof")
pairs = [Q.character_pair(rng) for _ in range(64)]
src = np.stack([p[0] for p in pairs]).astype(np.float32) / 127.5 - 1
tgt = np.stack([p[1] for p in pairs]).astype(np.float32) / 127.5 - 1
dev = lambda a: torch.as_tensor(a).cuda()
train = D.Dataset.from_batches((dev(src[i:i+4]), dev(tgt[i:i+4])) for i in range(0, 64, 4))
m = M.Pix2PixModel(train, train, "front2right", "fitprof", lambda_l1=100.0)
m.fit(50, 1000)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
m.fit(3000, 100000)
torch.cuda.synchronize()
pr.disable()
el = time.perf_counter() - t0
print(f"3000 steps in {el:.2f} s = {el/3000*1e3:.3f} ms/step")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:4500])
