#!/usr/bin/env python3
"""The workflow of the reference's experiments.ipynb against this build, as a command-line script: pick one of the four
models, load the sprite datasets, train, evaluate, export.  Every call it makes exists under the same name and with the
same arguments in the reference (cells noted in the comments); only the imports differ.  Run it from a folder that holds
`datasets/rpg-maker-xp/{train,test}/<direction>/<n>.png`:

    python examples/experiments.py --model histogram --epochs 1
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from palette_and_histo_gan_amd import configuration as cfg                     # noqa: E402
from palette_and_histo_gan_amd import dataset_utils, pix2pix_model             # noqa: E402
from palette_and_histo_gan_amd.tf_compat import tf                             # noqa: E402   stand-in for `import tensorflow as tf`

# model name -> (dataset loader arguments, model class, its loss weights)           cells 5, 7 and 9 of the notebook
RECIPES = {
    "baseline (no aug.)": (dict(augment=False), pix2pix_model.Pix2PixModel, dict(lambda_l1=100.)),
    "baseline": (dict(augment=True), pix2pix_model.Pix2PixAugmentedModel, dict(lambda_l1=100.)),
    "indexed": (dict(palette_ordering="grayness"), pix2pix_model.Pix2PixIndexedModel, dict(lambda_segmentation=0.01)),
    "histogram": (dict(augment=True), pix2pix_model.Pix2PixHistogramModel, dict(lambda_l1=30., lambda_histogram=1.)),
}


def main():
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--model", default="baseline (no aug.)", choices=sorted(RECIPES))
    ap.add_argument("--epochs", type=int, default=160)                               # cell 10
    ap.add_argument("--source", default="front", choices=cfg.DIRECTIONS)
    ap.add_argument("--target", default="right", choices=cfg.DIRECTIONS)
    ap.add_argument("--train-size", type=int, default=cfg.TRAIN_SIZE, help="use fewer sprites (smoke runs)")
    ap.add_argument("--test-size", type=int, default=cfg.TEST_SIZE)
    args = ap.parse_args()

    print("library:", tf.__version__, "| device:", tf.test.gpu_device_name() or "none -- it will not run")        # cell 1
    tf.random.set_seed(cfg.SEED)                                                                                  # cell 3
    src, tgt = cfg.DIRECTIONS.index(args.source), cfg.DIRECTIONS.index(args.target)
    ds_args, model_class, weights = RECIPES[args.model]
    sizes = dict(train_sizes=[args.train_size], test_sizes=[args.test_size])
    if model_class is pix2pix_model.Pix2PixIndexedModel:
        train_ds, test_ds = dataset_utils.load_indexed_ds(src, tgt, **ds_args, **sizes)
    else:
        train_ds, test_ds = dataset_utils.load_rgba_ds(src, tgt, **ds_args, **sizes)
    model = model_class(train_ds=train_ds, test_ds=test_ds, model_name=args.model,
                        architecture_name=f"{args.source}-to-{args.target}", **weights)

    steps = cfg.ceil(args.train_size / cfg.BATCH_SIZE) * args.epochs
    update_steps = max(1, steps // 40)
    print(f"{args.model}: {args.epochs} epochs = {steps} steps, evaluation every {update_steps} steps")
    model.fit(steps, update_steps, callbacks=["show_discriminator_output", "evaluate_l1"])     # "evaluate_fid": InceptionV3 download

    model.save_generator()                                                                      # cells 12-16
    model.generate_images_from_dataset("test")
    l1_train, l1_test = model.report_l1()
    print(f"L1: {float(l1_train):.5f} / {float(l1_test):.5f} (train/test)")


if __name__ == "__main__":
    main()
