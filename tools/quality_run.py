#!/usr/bin/env python3
"""Training-quality run of the notebook recipe (experiments.ipynb:204-212,268-275,387: "baseline (no aug.)" model, lambda_l1 = 100,
batch 4, 250 train / 44 test pairs, 10 080 steps, evaluation every 252) in BOTH arithmetic modes, reporting report_l1()
(side2side_model.py:162-176) the way the notebook prints it (experiments.ipynb:372: 0.00789 / 0.06371 train / test on the
RPG-Maker sprites with TensorFlow).

The reference's sprite files do not travel with this repository (their licence is not stated), so the pairs are SYNTHETIC
characters with a learnable front -> right relation: a character is a palette plus a few body parts; its "right" view is the
same character drawn narrower, shifted, with the face parts repainted in the hair colour.  The published numbers are therefore
not the yardstick here -- f32 mode is: the question this run answers is whether the bf16 storage mode (the benchmarked dtype)
trains to the same quality as the f32 parity mode.   python tools/quality_run.py [--steps 10080] > gpurun_out/quality_run.json
"""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from palette_and_histo_gan_amd import dataset_utils as D      # noqa: E402
from palette_and_histo_gan_amd import pix2pix_model as M      # noqa: E402


def character_pair(rng, S=64):
    """(front, right) uint8 RGBA sprites of one synthetic character"""
    n_col = int(rng.integers(6, 13))
    pal = np.concatenate([rng.integers(0, 256, size=(n_col, 3)), np.full((n_col, 1), 255)], axis=1).astype(np.uint8)
    skin, hair, shirt, trousers, shoe, eye = (pal[i % n_col] for i in range(6))
    cx = S // 2 + int(rng.integers(-3, 4))
    head_r, head_y = int(rng.integers(7, 11)), int(rng.integers(14, 20))
    torso_w, torso_h = int(rng.integers(8, 13)), int(rng.integers(12, 17))
    leg_h = int(rng.integers(10, 15))
    yy, xx = np.mgrid[0:S, 0:S]

    def draw(squeeze, shift, side):
        img = np.zeros((S, S, 4), np.uint8)
        x = (xx - cx - shift) / squeeze
        torso_top = head_y + head_r - 1
        img[(np.abs(x) <= torso_w) & (yy >= torso_top) & (yy < torso_top + torso_h)] = shirt
        legs_top = torso_top + torso_h
        leg = (yy >= legs_top) & (yy < legs_top + leg_h) & (np.abs(np.abs(x) - torso_w / 2) <= torso_w / 3)
        img[leg] = trousers
        img[leg & (yy >= legs_top + leg_h - 3)] = shoe
        head = x * x + (yy - head_y) ** 2 <= head_r * head_r
        img[head] = skin
        img[head & (yy < head_y - head_r // 3)] = hair
        eyes = (np.abs(np.abs(x) - head_r / 2.5) <= 1) & (np.abs(yy - head_y) <= 1)
        if side:       # seen from the right: the face is covered by hair, one eye
            img[head & (x < 0)] = hair
            eyes &= x > 0
        img[eyes & head] = eye
        return img
    return draw(1.0, 0, False), draw(0.6, 2, True)


def make_sets(n_train=250, n_test=44, seed=47):
    rng = np.random.default_rng(seed)
    pairs = [character_pair(rng) for _ in range(n_train + n_test)]
    src = np.stack([p[0] for p in pairs])
    tgt = np.stack([p[1] for p in pairs])
    return (src[:n_train], tgt[:n_train]), (src[n_train:], tgt[n_train:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10080)          # ceil(250 / 4) * 160 (experiments.ipynb:268)
    ap.add_argument("--update-steps", type=int, default=252)
    ap.add_argument("--dtypes", default="f32,bf16")
    args = ap.parse_args()
    (tr_s, tr_t), (te_s, te_t) = make_sets()
    os.makedirs(os.path.join(ROOT, "gpurun_out", "quality"), exist_ok=True)
    os.chdir(os.path.join(ROOT, "gpurun_out", "quality"))       # fit() writes its logs and checkpoints under ./temp-side2side
    out = {"recipe": {"model": "baseline (no aug.)", "lambda_l1": 100.0, "batch": 4, "train": 250, "test": 44, "steps": args.steps,
                      "data": "synthetic front -> right character pairs (tools/quality_run.py)"},
           "reference_published": {"l1_train": 0.00789, "l1_test": 0.06371, "note": "experiments.ipynb:372, RPG-Maker sprites, TF"}}
    for name in args.dtypes.split(","):
        train = D.SpriteRGBADataset(tr_s, tr_t, augment=False, batch_size=4, seed=47)
        test = D.SpriteRGBADataset(te_s, te_t, augment=False, batch_size=4, seed=48)
        with contextlib.redirect_stdout(sys.stderr):
            model = M.Pix2PixModel(train, test, "front2right", f"quality-{name}", lambda_l1=100.0, dtype=name, seed=47)
        curve = []
        orig = model.report_l1

        def spy(num_images=44, step=None, _orig=orig, _curve=curve):
            tr, te = _orig(num_images, step)
            _curve.append([int(step) if step is not None else -1, float(tr), float(te)])
            return tr, te
        model.report_l1 = spy
        t0 = time.time()
        with contextlib.redirect_stdout(sys.stderr):             # fit() prints its progress like the reference; stdout carries the JSON only
            model.fit(args.steps, args.update_steps, callbacks=["evaluate_l1"])
        torch.cuda.synchronize()
        wall = time.time() - t0
        tr, te = orig(44)
        out[name] = {"l1_train": float(tr), "l1_test": float(te), "wall_s": round(wall, 1),
                     "images_per_s_incl_eval": round(args.steps * 4 / wall, 1), "l1_curve_step_train_test": curve}
        print(f"[{name}] L1 {float(tr):.5f} / {float(te):.5f} (train/test), {wall:.1f} s", file=sys.stderr, flush=True)
    if "f32" in out and "bf16" in out:
        out["bf16_vs_f32"] = {"l1_train_ratio": out["bf16"]["l1_train"] / out["f32"]["l1_train"],
                              "l1_test_ratio": out["bf16"]["l1_test"] / out["f32"]["l1_test"]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
