#!/usr/bin/env python3
"""experiments.ipynb of fegemo/palette-and-histo-gan as a script against this build: the notebook's cells with their imports
changed and nothing else (cell numbers in the comments).  Run from a folder that holds `datasets/rpg-maker-xp/...`:

    python examples/experiments.py --model 1 --epochs 1            # 0 baseline (no aug.), 1 baseline, 2 indexed, 3 histogram
"""
import argparse
import os
import sys
from math import ceil

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# cell 1
from palette_and_histo_gan_amd.tf_compat import tf                     # noqa: E402   (reference: import tensorflow as tf)

print("Tensorflow version: ", tf.__version__)
if tf.test.gpu_device_name():
    print("Default GPU: {}".format(tf.test.gpu_device_name()))
else:
    print("Not using a GPU - it will take long!!")

# cell 3
from palette_and_histo_gan_amd.configuration import *                   # noqa: E402,F401,F403

ap = argparse.ArgumentParser()
ap.add_argument("--model", type=int, default=0)
ap.add_argument("--epochs", type=int, default=160)
ap.add_argument("--train-size", type=int, default=TRAIN_SIZE)          # smaller folders for a smoke run
ap.add_argument("--test-size", type=int, default=TEST_SIZE)
args = ap.parse_args()
print("DATASET_SIZE", DATASET_SIZE)
print("TRAIN_SIZE", args.train_size)
print("TEST_SIZE", args.test_size)
tf.random.set_seed(SEED)

# cell 5
MODELS = ["baseline (no aug.)", "baseline", "indexed", "histogram"]
model = MODELS[args.model]
source_direction = DIRECTION_FRONT
target_direction = DIRECTION_RIGHT
architecture_name = f"{DIRECTIONS[source_direction]}-to-{DIRECTIONS[target_direction]}"

# cell 7
from palette_and_histo_gan_amd.dataset_utils import load_indexed_ds, load_rgba_ds      # noqa: E402

sizes = dict(train_sizes=[args.train_size], test_sizes=[args.test_size])
if model == "baseline (no aug.)":
    train_ds, test_ds = load_rgba_ds(source_direction, target_direction, augment=False, **sizes)
elif model in ("baseline", "histogram"):
    train_ds, test_ds = load_rgba_ds(source_direction, target_direction, **sizes)
else:
    train_ds, test_ds = load_indexed_ds(source_direction, target_direction, palette_ordering="grayness", **sizes)

# cell 9
from palette_and_histo_gan_amd.pix2pix_model import (Pix2PixAugmentedModel, Pix2PixHistogramModel, Pix2PixIndexedModel,      # noqa: E402
                                                     Pix2PixModel)

if model == "baseline (no aug.)":
    model = Pix2PixModel(train_ds=train_ds, test_ds=test_ds, model_name="baseline (no aug.)",
                         architecture_name=architecture_name, lambda_l1=100.)
elif model == "baseline":
    model = Pix2PixAugmentedModel(train_ds=train_ds, test_ds=test_ds, model_name="baseline",
                                  architecture_name=architecture_name, lambda_l1=100.)
elif model == "indexed":
    model = Pix2PixIndexedModel(train_ds=train_ds, test_ds=test_ds, model_name="indexed",
                                architecture_name=architecture_name, lambda_segmentation=0.01)
else:
    model = Pix2PixHistogramModel(train_ds=train_ds, test_ds=test_ds, model_name="histogram",
                                  architecture_name=architecture_name, lambda_l1=30., lambda_histogram=1.)

# cell 10
EPOCHS = args.epochs
STEPS = ceil(args.train_size / BATCH_SIZE) * EPOCHS
UPDATE_STEPS = max(1, STEPS // 40)
print(f"Starting training for {EPOCHS} epochs in {STEPS} steps, updating visualization every {UPDATE_STEPS} steps...")
callbacks = ["show_discriminator_output", "evaluate_l1"]                # "evaluate_fid" needs the InceptionV3 download
model.fit(STEPS, UPDATE_STEPS, callbacks=callbacks)

# cells 12-16
model.save_generator()
model.generate_images_from_dataset("test")
l1_train, l1_test = model.report_l1()
print(f"L1: {float(l1_train):.5f} / {float(l1_test):.5f} (train/test)")
