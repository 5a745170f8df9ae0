"""S2SModel: the training loop the hot path sits behind (reference side2side_model.py:28-126).

Kept: constructor signature, fit/do_fit call order (examples -> loop over train_ds.repeat().take(steps).enumerate()
-> periodic evaluation -> train_step(batch, step, update_steps) -> checkpoint cadence), the overridable hooks, the
attribute names subclasses use, report_l1 over TEST_SIZE images (side2side_model.py:162-176).  Replaced: the TensorFlow
summary writer -> ScalarLog (JSON lines + a real TensorBoard event file, scalars kept on the device until flush());
tf.train.Checkpoint / CheckpointManager -> flat-tensor files with the same object structure; the matplotlib previews
and the FID callback (needs an InceptionV3 download, frechet_inception_distance.py:76) are reported as skipped.
"""
from abc import ABC, abstractmethod
import datetime
import json
import os
import time

import numpy as np
import torch

from . import tb_events
from .configuration import TEMP_FOLDER, TEST_SIZE


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
    """Stand-in for tf.summary's file writer.  scalar() only RECORDS: a device scalar stays on the device (no host
    synchronisation inside train_step, so the host keeps running ahead of the GPU); flush() -- called by fit() at the end and
    automatically every `max_pending` rows -- moves all pending values to the host in ONE transfer and appends them to
    scalars.jsonl and to a TensorBoard event file in the reference's log folder."""

    def __init__(self, folder, max_pending=4096):
        os.makedirs(folder, exist_ok=True)
        self.path = os.path.join(folder, "scalars.jsonl")
        self.events = tb_events.EventFileWriter(folder)
        self.max_pending = max_pending
        self._rows = []          # (name, value or device tensor, step, wall time)

    def scalar(self, name, value, step):
        self._rows.append((name, value, int(step), time.time()))
        if len(self._rows) >= self.max_pending:
            self.flush()

    def image(self, name, sheet, step, png_bytes=None):
        """tf.summary.image(name, data, step, max_outputs=5) of ONE uint8 RGBA sheet (side2side_model.py:92-93); png_bytes: the
        sheet already encoded (fit() has just written it to `name`)"""
        from . import png
        self.flush()            # keep the file in step order
        data = png_bytes if png_bytes is not None else png.encode_png(sheet)
        self.events.add_values([tb_events.encode_image_value(name, [data], sheet.shape[1], sheet.shape[0])], step)

    def write_raw_pb(self, value, step):
        """tf.summary.experimental.write_raw_pb of one serialized Summary.Value (side2side_model.py:60-61)"""
        self.flush()
        self.events.add_values([value], step)

    def flush(self):
        if not self._rows:
            return
        rows, self._rows = self._rows, []
        dev = [i for i, r in enumerate(rows) if isinstance(r[1], torch.Tensor) and r[1].is_cuda]
        host = {}
        if dev:
            stacked = torch.stack([rows[i][1].detach().reshape(()).to(torch.float32) for i in dev]).cpu().numpy()
            host = {i: float(v) for i, v in zip(dev, stacked)}
        out = [(r[0], host[i] if i in host else float(r[1]), r[2], r[3]) for i, r in enumerate(rows)]
        with open(self.path, "a") as f:
            for name, value, step, _ in out:
                f.write(json.dumps({"name": name, "value": value, "step": step}) + "\n")
        self.events.add_scalars(out)


class Checkpoint:
    """tf.train.Checkpoint(generator_optimizer=, discriminator_optimizer=, generator=, discriminator=)
    (pix2pix_model.py:30-34): the objects whose state a checkpoint holds.  All of that state lives in the engine's flat
    buffers: weights, Adam moments and step counts of both networks, and the device dropout counter."""

    def __init__(self, generator_optimizer=None, discriminator_optimizer=None, generator=None, discriminator=None, engine=None):
        self.generator_optimizer, self.discriminator_optimizer = generator_optimizer, discriminator_optimizer
        self.generator, self.discriminator, self.engine = generator, discriminator, engine

    def state(self):
        e = self.engine
        return {"G": e.G.params.cpu(), "G.m": e.G.m.cpu(), "G.v": e.G.v.cpu(), "G.t": e.G.t,
                "D": e.D.params.cpu(), "D.m": e.D.m.cpu(), "D.v": e.D.v.cpu(), "D.t": e.D.t,
                "mask_counter": int(e.mask_counter_dev.item()), "step_count": e.step_count}

    def save(self, path):
        torch.save(self.state(), path)
        return path

    def restore(self, path):
        """weights, both Adam states (moments and step counts) and the dropout counter, so that the next step is the one that
        would have followed the save"""
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


class CheckpointManager:
    """tf.train.CheckpointManager(checkpoint, directory=, max_to_keep=1) (pix2pix_model.py:35-36); `.save()` as do_fit
    calls it (side2side_model.py:121-122)."""

    def __init__(self, checkpoint, directory, max_to_keep=1):
        self.checkpoint, self.directory, self.max_to_keep = checkpoint, directory, max_to_keep
        self.saved = []

    def save(self):
        os.makedirs(self.directory, exist_ok=True)
        path = self.checkpoint.save(os.path.join(self.directory, f"ckpt-{len(self.saved) + 1}.pt"))
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
        path = path or self.latest_checkpoint
        if path is None:
            raise FileNotFoundError("no checkpoint to restore")
        return self.checkpoint.restore(path)


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
        self.layout_summary = S2SModel.create_layout_summary()

    @staticmethod
    def create_layout_summary():
        """side2side_model.py:240-273: the custom-scalar layout (one multiline chart for the FID tags, one for the L1 evaluation
        tags), as the serialized Summary.Value that fit() writes at step 0"""
        return tb_events.encode_layout_value(tb_events.encode_layout([
            ("Fréchet Inception Distance", [("FID for train and test", [r"^fid\/"])]),
            ("L1 Evaluation", [("L1 for train and test", [r"^l1\-evaluation\/"])])]))

    @property
    def is_main_rank(self):
        """data parallelism (build-added): rank 0 owns every side effect of fit() -- the log folder, the TensorBoard event file,
        previews, evaluation callbacks, console output and checkpoints; the other ranks only train"""
        dp = getattr(self, "data_parallel", None)
        return dp is None or dp.rank == 0

    @staticmethod
    def _whole(dataset):
        """evaluation / preview view of a dataset: whole batches whatever the training shard (dataset_utils.set_shard)"""
        return dataset.unsharded() if hasattr(dataset, "unsharded") else dataset

    def _rank_barrier(self):
        dp = getattr(self, "data_parallel", None)
        if dp is not None:
            dp.barrier()

    def fit(self, steps, update_steps, callbacks=[], starting_step=0):
        """side2side_model.py:54-65"""
        if starting_step == 0 and self.is_main_rank:
            self.log_folders = [TEMP_FOLDER, "logs", self.architecture_name, self.model_name]
            self.now_string = datetime.datetime.now().strftime("%Y%m%d-%H%M%S")
            self.summary_writer = ScalarLog(os.sep.join([*self.log_folders, self.now_string]))
            self.summary_writer.write_raw_pb(self.layout_summary, step=0)          # side2side_model.py:59-61
        try:
            self.do_fit(steps, update_steps, callbacks, starting_step)
        finally:
            if self.summary_writer is not None:
                self.summary_writer.flush()
        # success path only: nobody leaves fit() before rank 0 has written its last checkpoint and log rows.  A rank whose do_fit
        # raised must NOT enter a barrier while its peers sit in gradient all-reduces (mismatched collectives hang until the
        # timeout instead of surfacing the exception): it re-raises and the launcher tears the job down
        self._rank_barrier()

    def do_fit(self, steps, update_steps=1000, callbacks=[], starting_step=0):
        """side2side_model.py:67-122"""
        main = self.is_main_rank
        examples = self.select_examples_for_visualization() if main else []
        training_start_time = time.time()
        step_start_time = training_start_time
        for step, batch in self.train_ds.repeat().take(steps).enumerate():
            step += starting_step
            if main and ((step + 1) % update_steps == 0 or step == 0):
                if step != 0:
                    show_eta(training_start_time, step_start_time, step, starting_step, steps, update_steps)
                step_start_time = time.time()
                # side2side_model.py:86-93: the sheet goes to <log folder>/step_NNNNNN.png and, under that name, into the event file
                save_image_name = os.sep.join([TEMP_FOLDER, "logs", self.architecture_name, self.model_name, str(self.now_string),
                                               "step_{:06d}.png".format(step + 1)])
                print(f"Previewing images generated at step {step + 1} (3 test + 3 train)...")
                sheet = self.preview_generated_images_during_training(examples, save_image_name, step + 1)
                if self.summary_writer is not None:
                    with open(save_image_name, "rb") as f:          # the file preview_generated_images_during_training has just written
                        self.summary_writer.image(save_image_name, sheet, step=(step + 1) // update_steps, png_bytes=f.read())
                if "show_discriminator_output" in callbacks:
                    self.show_discriminated_images("test", 2)          # side2side_model.py:96-97 (numbers instead of plots)
                if "evaluate_l1" in callbacks:
                    l1_train, l1_test = self.report_l1(step=(step + 1) // update_steps)
                    print(f" L1: {float(l1_train):.5f} / {float(l1_test):.5f} (train/test)")
                if "evaluate_fid" in callbacks:
                    print("FID needs the InceptionV3 ImageNet weights (network fetch): skipped")
                print(f"Step: {(step + 1) / 1000}k")
            self.train_step(batch, step, update_steps)
            if main and ((step + 1) % (update_steps * 5) == 0 or (step - starting_step + 1) == steps):
                self.checkpoint_manager.save()      # every rank holds the same weights and Adam state: one copy on disk

    @abstractmethod
    def train_step(self, batch, step, UPDATE_STEPS):
        pass

    @abstractmethod
    def select_examples_for_visualization(self, number_of_examples=6):
        pass

    def _to_rgba_u8(self, image, palette=None):
        """one (S,S,C) image of a batch as displayable uint8 RGBA: normalised RGBA -> (x * 0.5 + 0.5) * 255 (what the reference
        hands to imshow, pix2pix_model.py:141), palette indices -> palette colours (io_utils.py:96-103)"""
        from . import io_utils
        t = image.detach().cpu() if isinstance(image, torch.Tensor) else torch.as_tensor(np.asarray(image))
        if palette is not None:
            pal = palette.detach().cpu() if isinstance(palette, torch.Tensor) else torch.as_tensor(np.asarray(palette))
            return io_utils.indexed_to_rgba(t.to(torch.int64), pal).clamp(0, 255).to(torch.uint8).numpy()
        return ((t.to(torch.float32) * 0.5 + 0.5).clamp(0, 1) * 255.0).round().to(torch.uint8).numpy()

    def preview_generated_images_during_training(self, examples, save_name, step):
        """pix2pix_model.py:127-157 / :332-365: one row per example with the columns Input | Target | Generated.  The
        reference draws a matplotlib figure; here the sheet is a uint8 RGBA array (returned) and, with `save_name`, a PNG."""
        from . import png
        rows = []
        for ex in examples:
            fake = self.generate(ex)
            pal = ex[2][0] if len(ex) == 3 else None
            tiles = [self._to_rgba_u8(ex[0][0], pal), self._to_rgba_u8(ex[1][0], pal), self._to_rgba_u8(fake[0], pal)]
            gap = np.zeros((tiles[0].shape[0], 2, 4), np.uint8)
            rows.append(np.concatenate([tiles[0], gap, tiles[1], gap, tiles[2]], axis=1))
            rows.append(np.zeros((2, rows[-1].shape[1], 4), np.uint8))
        sheet = np.concatenate(rows[:-1], axis=0) if rows else np.zeros((1, 1, 4), np.uint8)
        if save_name is not None:
            folder = os.path.dirname(save_name)
            if folder:
                os.makedirs(folder, exist_ok=True)
            png.write_png(save_name, sheet)
        return sheet

    def generate_images_from_dataset(self, dataset_name="test", num_images=None, steps=None):
        """side2side_model.py:202-224: `<TEMP_FOLDER>/generated-images/<architecture>/<model>/<i>.png`, one sheet per sample"""
        import shutil
        from .configuration import TRAIN_SIZE
        is_test = dataset_name == "test"
        limit = TEST_SIZE if is_test else TRAIN_SIZE
        num_images = limit if num_images is None else min(num_images, limit)
        dataset = list(self._whole(self.test_ds if is_test else self.train_ds).unbatch().take(num_images).batch(1).as_numpy_iterator())
        base = os.sep.join([TEMP_FOLDER, "generated-images", self.architecture_name, self.model_name])
        shutil.rmtree(base, ignore_errors=True)
        os.makedirs(base, exist_ok=True)
        for i, images in enumerate(dataset):
            self.preview_generated_images_during_training([images], os.sep.join([base, f"{i}.png"]), steps)
        print(f"Generated {len(dataset)} images (using \"{dataset_name}\" dataset)")
        return base

    def debug_discriminator_patches(self, batch_of_one):
        """pix2pix_model.py:160-231: sigmoid of the discriminator's patch logits for the real and the generated image of one
        sample (the reference plots them); returns {"real": (h,w) array, "fake": ..., "real_mean": float, "fake_mean": float}"""
        source, real = batch_of_one[0], batch_of_one[1]
        fake = self.generator(source, training=True)
        out = {}
        for name, img in (("real", real), ("fake", fake)):
            logits = self.discriminator([img, source], training=True)
            prob = torch.sigmoid(logits[0, :, :, 0].to(torch.float32)).cpu().numpy()
            out[name], out[name + "_mean"] = prob, float(prob.mean())
        return out

    def show_discriminated_images(self, dataset_name="test", num_images=2):
        """side2side_model.py:228-239 (numbers instead of plots)"""
        dataset = self._whole(self.test_ds if dataset_name == "test" else self.train_ds)
        res = []
        for images in list(dataset.unbatch().take(num_images).batch(1).as_numpy_iterator()):
            res.append(self.debug_discriminator_patches(images))
            print(f"Discriminated target {res[-1]['real_mean']:.3f} / generated {res[-1]['fake_mean']:.3f}")
        return res

    # -- evaluation (side2side_model.py:140-176) ---------------------------------------------------------------------------
    def select_examples_for_evaluation(self, num_images, dataset):
        """pix2pix_model.py:112-122 / :433-452: (real_images, fake_images) of the first num_images samples, each generated as a
        batch of one like the reference does; indexed batches are looked up in their palette (io_utils.py:96-103)."""
        from . import io_utils
        real, fake = [], []
        dataset = self._whole(dataset)
        for batch in dataset.unbatch().take(num_images).batch(1):
            out = self.generate(batch)
            host = lambda t: (t.detach().cpu() if isinstance(t, torch.Tensor) else torch.as_tensor(np.asarray(t)))   # noqa: E731
            if len(batch) == 3:          # (source_idx, target_idx, palette)
                pal = host(batch[2][0])
                real.append(io_utils.indexed_to_rgba(host(batch[1][0]), pal).to(torch.float32))
                fake.append(io_utils.indexed_to_rgba(out[0].cpu(), pal).to(torch.float32))
            else:
                real.append(host(batch[1][0]).to(torch.float32))
                fake.append(out[0].to(torch.float32).cpu())
        return torch.stack(real), torch.stack(fake)

    def evaluate_l1(self, real_images, fake_images):
        """pix2pix_model.py:124-125"""
        return (torch.as_tensor(fake_images, dtype=torch.float32) - torch.as_tensor(real_images, dtype=torch.float32)).abs().mean()

    # -- model export (side2side_model.py:178-200: the reference writes SavedModels; here the documented Keras-layout file) --
    def _model_path(self, which):
        return os.sep.join(["models", "py", which, self.architecture_name, self.model_name])

    def save_generator(self):
        """side2side_model.py:178-184: `models/py/generator/<architecture>/<model>/weights.p2pw.npz` (keras_weights.py)"""
        from . import keras_weights
        os.makedirs(self._model_path("generator"), exist_ok=True)
        return keras_weights.export_model(self, os.path.join(self._model_path("generator"), "weights.p2pw.npz"),
                                          with_optimizer=False, which=("generator",))

    def save_discriminator(self):
        """side2side_model.py:190-196"""
        from . import keras_weights
        os.makedirs(self._model_path("discriminator"), exist_ok=True)
        return keras_weights.export_model(self, os.path.join(self._model_path("discriminator"), "weights.p2pw.npz"),
                                          with_optimizer=False, which=("discriminator",))

    def load_generator(self):
        """side2side_model.py:186-188: replaces the generator only (the discriminator and both optimizers stay as they are)"""
        from . import keras_weights
        keras_weights.import_model(self, os.path.join(self._model_path("generator"), "weights.p2pw.npz"),
                                   with_optimizer=False, which=("generator",))

    def load_discriminator(self):
        """side2side_model.py:198-200: replaces the discriminator only"""
        from . import keras_weights
        keras_weights.import_model(self, os.path.join(self._model_path("discriminator"), "weights.p2pw.npz"),
                                   with_optimizer=False, which=("discriminator",))

    def report_l1(self, num_images=TEST_SIZE, step=None):
        """side2side_model.py:162-176"""
        train_real_images, train_fake_images = self.select_examples_for_evaluation(num_images, self.train_ds)
        test_real_images, test_fake_images = self.select_examples_for_evaluation(num_images, self.test_ds)
        train_value = self.evaluate_l1(train_real_images, train_fake_images)
        test_value = self.evaluate_l1(test_real_images, test_fake_images)
        if self.summary_writer is not None and step is not None:
            self.summary_writer.scalar("l1-evaluation/train", train_value, step)
            self.summary_writer.scalar("l1-evaluation/test", test_value, step)
        return train_value, test_value
