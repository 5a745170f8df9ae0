"""S2SModel: the training loop the hot path sits behind (reference side2side_model.py:28-126).

Kept: constructor signature, fit/do_fit call order (examples -> loop over train_ds.repeat().take(steps).enumerate()
-> periodic evaluation -> train_step(batch, step, update_steps) -> checkpoint cadence), the overridable hooks and the
attribute names subclasses use.  Replaced: TensorBoard writer -> a JSON-lines scalar log; tf.train.CheckpointManager
-> a flat-tensor checkpoint (out of scope beyond `.save()`); the matplotlib previews and the
FID callback (needs an InceptionV3 download, frechet_inception_distance.py:76) are reported as skipped.
"""
from abc import ABC, abstractmethod
import datetime
import json
import os
import time

import torch

from .configuration import TEMP_FOLDER


def seconds_to_human_readable(seconds):
    seconds = int(seconds)
    h, rem = divmod(seconds, 3600)
    m, s = divmod(rem, 60)
    return f"{h:02d}:{m:02d}:{s:02d}"


def show_eta(training_start_time, step_start_time, current_step, training_starting_step, total_steps, update_steps):
    """side2side_model.py:14-25"""
    now = time.time()
    elapsed = now - training_start_time
    steps_so_far = float(current_step - training_starting_step)
    eta = elapsed / (steps_so_far + 1.0) * (total_steps - steps_so_far)
    print(f"Time since start: {seconds_to_human_readable(elapsed)}")
    print(f"Estimated time to finish: {seconds_to_human_readable(eta)}")
    print(f"Last {update_steps} steps took: {now - step_start_time:.2f}s\n")


class ScalarLog:
    """Stand-in for tf.summary's file writer: scalars appended as JSON lines under the reference's log folder."""

    def __init__(self, folder):
        os.makedirs(folder, exist_ok=True)
        self.path = os.path.join(folder, "scalars.jsonl")
        self._rows = []

    def scalar(self, name, value, step):
        self._rows.append({"name": name, "value": float(value), "step": int(step)})

    def flush(self):
        if self._rows:
            with open(self.path, "a") as f:
                for r in self._rows:
                    f.write(json.dumps(r) + "\n")
            self._rows = []


class CheckpointManager:
    """`.save()` as do_fit calls it (side2side_model.py:121-122): generator, discriminator and both Adam states."""

    def __init__(self, engine, directory, max_to_keep=1):
        self.engine, self.directory, self.max_to_keep = engine, directory, max_to_keep
        self.saved = []

    def save(self):
        os.makedirs(self.directory, exist_ok=True)
        path = os.path.join(self.directory, f"ckpt-{len(self.saved) + 1}.pt")
        e = self.engine
        torch.save({"G": e.G.params.cpu(), "G.m": e.G.m.cpu(), "G.v": e.G.v.cpu(), "G.t": e.G.t,
                    "D": e.D.params.cpu(), "D.m": e.D.m.cpu(), "D.v": e.D.v.cpu(), "D.t": e.D.t,
                    "mask_counter": int(e.mask_counter_dev.item()), "step_count": e.step_count}, path)
        self.saved.append(path)
        while len(self.saved) > self.max_to_keep:
            old = self.saved.pop(0)
            if os.path.exists(old):
                os.remove(old)
        return path

    @property
    def latest_checkpoint(self):
        return self.saved[-1] if self.saved else None

    def restore(self, path=None):
        """Resume: weights, both Adam states (moments and step counts) and the dropout counter, so that the next step
        is the one that would have followed the save (tf.train.Checkpoint.restore in the reference's workflow)."""
        path = path or self.latest_checkpoint
        if path is None:
            raise FileNotFoundError("no checkpoint to restore")
        ck = torch.load(path, map_location="cpu")
        e = self.engine
        for store, k in ((e.G, "G"), (e.D, "D")):
            store.params.copy_(ck[k])
            store.m.copy_(ck[k + ".m"])
            store.v.copy_(ck[k + ".v"])
            store.t = int(ck[k + ".t"])
            store.t_dev.fill_(store.t)
        e.mask_counter_dev.fill_(int(ck.get("mask_counter", 0)))
        e.step_count = int(ck.get("step_count", 0))
        e.refresh_weight_copies()
        return path


class S2SModel(ABC):
    def __init__(self, train_ds, test_ds, model_name, architecture_name="s2smodel"):
        """side2side_model.py:29-52"""
        self.generator = None
        self.discriminator = None
        self.summary_writer = None
        self.now_string = None
        self.log_folders = None
        self.train_ds = train_ds
        self.test_ds = test_ds
        self.model_name = model_name
        self.architecture_name = architecture_name
        self.checkpoint_dir = os.sep.join([TEMP_FOLDER, "training-checkpoints", self.architecture_name, self.model_name])

    def fit(self, steps, update_steps, callbacks=[], starting_step=0):
        """side2side_model.py:54-65"""
        if starting_step == 0:
            self.log_folders = [TEMP_FOLDER, "logs", self.architecture_name, self.model_name]
            self.now_string = datetime.datetime.now().strftime("%Y%m%d-%H%M%S")
            self.summary_writer = ScalarLog(os.sep.join([*self.log_folders, self.now_string]))
        try:
            self.do_fit(steps, update_steps, callbacks, starting_step)
        finally:
            self.summary_writer.flush()

    def do_fit(self, steps, update_steps=1000, callbacks=[], starting_step=0):
        """side2side_model.py:67-122"""
        examples = self.select_examples_for_visualization()
        training_start_time = time.time()
        step_start_time = training_start_time
        for step, batch in self.train_ds.repeat().take(steps).enumerate():
            step += starting_step
            if (step + 1) % update_steps == 0 or step == 0:
                if step != 0:
                    show_eta(training_start_time, step_start_time, step, starting_step, steps, update_steps)
                step_start_time = time.time()
                self.preview_generated_images_during_training(examples, None, step + 1)
                if "show_discriminator_output" in callbacks:
                    print("Discriminator output patches: plotting is not part of this build (skipped)")
                if "evaluate_l1" in callbacks:
                    l1_train, l1_test = self.report_l1(step=(step + 1) // update_steps)
                    print(f" L1: {l1_train:.5f} / {l1_test:.5f} (train/test)")
                if "evaluate_fid" in callbacks:
                    print("FID needs the InceptionV3 ImageNet weights (network fetch): skipped")
                print(f"Step: {(step + 1) / 1000}k")
            self.train_step(batch, step, update_steps)
            if (step + 1) % (update_steps * 5) == 0 or (step - starting_step + 1) == steps:
                self.checkpoint_manager.save()

    @abstractmethod
    def train_step(self, batch, step, UPDATE_STEPS):
        pass

    @abstractmethod
    def select_examples_for_visualization(self, number_of_examples=6):
        pass

    @abstractmethod
    def preview_generated_images_during_training(self, examples, save_name, step):
        pass

    def report_l1(self, step=None, num_batches=8):
        """side2side_model.py:162-176: mean |target - generated| over a few batches of train and test."""
        out = []
        for ds in (self.train_ds, self.test_ds):
            tot, cnt = 0.0, 0
            for batch in ds.take(num_batches):
                l1 = self.evaluate_l1_batch(batch)
                tot += l1 * len(batch[0])
                cnt += len(batch[0])
            out.append(tot / max(cnt, 1))
        if self.summary_writer is not None and step is not None:
            self.summary_writer.scalar("l1_evaluation/train", out[0], step)
            self.summary_writer.scalar("l1_evaluation/test", out[1], step)
        return tuple(out)
