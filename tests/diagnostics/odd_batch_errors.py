#!/usr/bin/env python3
"""f32 whole-step gradient error against the f64 oracle for several batch sizes and seeds (GPU): shows that the occasional
O(1e-3) max-norm outlier is a ReLU/LeakyReLU/dropout-gated element whose pre-activation is ~1e-6 from zero (f32 vs f64
summation order), not a batch-size dependent defect.    python tests/diagnostics/odd_batch_errors.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from palette_and_histo_gan_amd import _lib as L                    # noqa: E402
from tests.test_train_step_gpu import run_case                      # noqa: E402

for B in (1, 2, 3, 4, 5):
    for seed in (60, 61, 62):
        out, want, wg, wd = run_case(L.F32, True, seed=seed, B=B)
        print(f"B={B} seed={seed}: worst max-norm {wg[0][0]} {wg[0][1]:.2e}   worst L2 {wg[1][0]} {wg[1][1]:.2e}", flush=True)
