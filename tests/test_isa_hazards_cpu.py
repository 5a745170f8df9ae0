"""not gpu: the compiled kernels hold no global store whose data registers a following MFMA writes back before the store is known to
have read them (tools/exp/store_mfma_hazard.py; measured in round 4: hipcc pads that pair for vector instructions only, and the
matrix pipe's write-back is not interlocked against the store's read).  Scans every source file that holds MFMAs (~1 min of hipcc -S)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "palette_and_histo_gan_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
KERNELS = ["igemm", "wgrad_small", "brig", "hist", "wgemm", "conv_strip", "conv_fewin", "conv_fewout", "head_softmax"]      # every file with MFMAs, longest compile first


def _asm(name, out_dir):
    out = os.path.join(out_dir, name + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                    "-I" + CSRC, os.path.join(CSRC, name + ".hip"), "-o", out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_mfma_writes_back_into_the_data_registers_of_a_pending_store(tmp_path):
    with ThreadPoolExecutor(max_workers=4) as pool:
        files = list(pool.map(lambda n: _asm(n, str(tmp_path)), KERNELS))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exp", "store_mfma_hazard.py")] + files, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert all(f"{os.path.basename(f)}: 0 store" in r.stdout for f in files), r.stdout
