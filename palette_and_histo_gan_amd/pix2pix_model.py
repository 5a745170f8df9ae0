"""The Pix2Pix model family with the reference's class surface (pix2pix_model.py:11-325) on the MI355X engine.

    Pix2PixModel(train_ds, test_ds, model_name, architecture_name, lambda_l1)
    Pix2PixAugmentedModel(same)                  -- augmentation lives in the dataset (pix2pix_model.py:232-234)
    Pix2PixHistogramModel(..., lambda_l1, lambda_histogram)
    Pix2PixIndexedModel(train_ds, test_ds, model_name, architecture_name, lambda_segmentation=0.5)

train_step(batch, step, update_steps) keeps the reference's contract (one optimisation step of both networks with
gradients taken at the same pre-update weights, scalars logged at step // update_steps) and additionally RETURNS
(g_loss_tuple, d_loss_tuple) as device scalars, and tolerates summary_writer=None.
Extra keyword arguments (dtype, img_size, device, data_parallel) are build-added; defaults reproduce the reference.
"""
import torch

from . import _lib as L
from . import histogram as _histogram  # noqa: F401  (module parity with the reference's `import histogram`)
from .configuration import IMG_SIZE, MAX_PALETTE_SIZE
from .engine import Pix2PixEngine
from .networks import PatchDiscriminator, UnetGenerator
from .side2side_model import CheckpointManager, S2SModel


class Pix2PixModel(S2SModel):
    def __init__(self, train_ds, test_ds, model_name, architecture_name, lambda_l1, dtype="bf16", img_size=IMG_SIZE,
                 device="cuda:0", data_parallel=None, seed=47):
        super().__init__(train_ds, test_ds, model_name, architecture_name)
        self.lambda_l1 = lambda_l1
        self._dtype = {"bf16": L.BF16, "f32": L.F32}[dtype] if isinstance(dtype, str) else dtype
        self._img_size, self._device, self._seed = img_size, device, seed
        self.data_parallel = data_parallel
        self.engine = self.create_engine()
        self.generator = self.create_generator()
        self.discriminator = self.create_discriminator()
        print(f"Generator: {self.generator.name} with {self.generator.count_params():,} parameters")
        print(f"Discriminator: {self.discriminator.name} with {self.discriminator.count_params():,} parameters")
        # tf.keras.optimizers.Adam(0.0002, beta_1=0.5) x2 live inside the engine (pix2pix_model.py:28-29)
        self.generator_optimizer = self.engine.G
        self.discriminator_optimizer = self.engine.D
        self.checkpoint_manager = CheckpointManager(self.engine, self.checkpoint_dir, max_to_keep=1)

    # -- construction hooks (pix2pix_model.py:38-42) ------------------------------------------------------------
    def create_engine(self):
        return Pix2PixEngine(4, 4, "tanh", self._img_size, self._dtype, device=self._device, seed=self._seed)

    def create_generator(self):
        return UnetGenerator(self.engine)

    def create_discriminator(self):
        return PatchDiscriminator(self.engine)

    # -- losses as standalone evaluations (pix2pix_model.py:44-56); train_step computes them fused ----------------
    def generator_loss(self, fake_predicted, fake_image, real_image):
        fp = torch.as_tensor(fake_predicted, dtype=torch.float32)
        adv = torch.nn.functional.binary_cross_entropy_with_logits(fp, torch.ones_like(fp))
        l1 = (torch.as_tensor(real_image, dtype=torch.float32).to(fp.device) - torch.as_tensor(fake_image, dtype=torch.float32).to(fp.device)).abs().mean()
        return adv + self.lambda_l1 * l1, adv, l1

    def discriminator_loss(self, real_predicted, fake_predicted):
        rp = torch.as_tensor(real_predicted, dtype=torch.float32)
        fp = torch.as_tensor(fake_predicted, dtype=torch.float32)
        real = torch.nn.functional.binary_cross_entropy_with_logits(rp, torch.ones_like(rp))
        fake = torch.nn.functional.binary_cross_entropy_with_logits(fp, torch.zeros_like(fp))
        return fake + real, real, fake

    def generate(self, batch):
        """pix2pix_model.py:58-60"""
        source_image, _ = batch
        return self.generator(source_image, training=True)

    # -- the hot path ----------------------------------------------------------------------------------------------
    def _dp(self):
        dp = self.data_parallel
        return (1, None) if dp is None else (dp.world, dp)

    def train_step(self, batch, step, update_steps):
        """pix2pix_model.py:62-89"""
        source_image, real_image = batch
        world, dp = self._dp()
        out = self.engine.train_step_rgba(source_image, real_image, self.lambda_l1,
                                          global_batch=len(source_image) * world, dp=dp)
        g_loss, d_loss = (out[0], out[1], out[2]), (out[4], out[5], out[6])
        self._log(g_loss, d_loss, step, update_steps)
        return g_loss, d_loss

    def _log(self, g_loss, d_loss, step, update_steps):
        if self.summary_writer is None:
            return
        s = int(step) // int(update_steps)
        self.log_generator_loss(g_loss, s)
        self.log_discriminator_loss(d_loss, s)

    def log_generator_loss(self, g_loss, step):
        """pix2pix_model.py:91-95"""
        total_loss, adversarial_loss, l1_loss = g_loss[:3]
        self.summary_writer.scalar("generator/total_loss", total_loss, step)
        self.summary_writer.scalar("generator/adversarial_loss", adversarial_loss, step)
        self.summary_writer.scalar("generator/l1_loss", l1_loss, step)

    def log_discriminator_loss(self, d_loss, step):
        """pix2pix_model.py:97-101"""
        total_loss, real_loss, fake_loss = d_loss
        self.summary_writer.scalar("discriminator/total_loss", total_loss, step)
        self.summary_writer.scalar("discriminator/real_loss", real_loss, step)
        self.summary_writer.scalar("discriminator/fake_loss", fake_loss, step)

    # -- evaluation helpers used by do_fit ---------------------------------------------------------------------------
    def select_examples_for_visualization(self, number_of_examples=6):
        """pix2pix_model.py:103-110"""
        num_train_examples = number_of_examples // 2
        num_test_examples = number_of_examples - num_train_examples
        train_examples = self.train_ds.unbatch().take(num_train_examples).batch(1)
        test_examples = self.test_ds.unbatch().take(num_test_examples).batch(1)
        return list(test_examples.as_numpy_iterator()) + list(train_examples.as_numpy_iterator())

    def preview_generated_images_during_training(self, examples, save_name, step):
        """The reference plots input/target/generated triples (pix2pix_model.py:112-125); here the images are generated
        (so the forward path runs exactly as in the reference's loop) and returned, plotting is out of scope."""
        return [self.generate(ex) for ex in examples]

    def evaluate_l1_batch(self, batch):
        source, target = batch[0], batch[1]
        fake = self.generate((source, target))
        return float((torch.as_tensor(target, dtype=torch.float32).to(fake.device) - fake).abs().mean())


class Pix2PixAugmentedModel(Pix2PixModel):
    """pix2pix_model.py:232-234"""

    def __init__(self, train_ds, test_ds, model_name, architecture_name, lambda_l1, **kw):
        super().__init__(train_ds, test_ds, model_name, architecture_name, lambda_l1, **kw)


class Pix2PixHistogramModel(Pix2PixAugmentedModel):
    """pix2pix_model.py:237-258"""

    def __init__(self, train_ds, test_ds, model_name, architecture_name, lambda_l1, lambda_histogram, **kw):
        super().__init__(train_ds, test_ds, model_name, architecture_name, lambda_l1, **kw)
        self.lambda_histogram = lambda_histogram

    def generator_loss(self, fake_predicted, fake_image, real_image):
        real_histogram = self.engine.rgbuv_histogram(real_image)
        fake_histogram = self.engine.rgbuv_histogram(fake_image)
        histogram_loss = _histogram.hellinger_loss(real_histogram, fake_histogram)
        total_loss, adversarial_loss, l1_loss = super().generator_loss(fake_predicted, fake_image, real_image)
        total_loss = total_loss + self.lambda_histogram * histogram_loss
        return total_loss, adversarial_loss, l1_loss, histogram_loss

    def train_step(self, batch, step, update_steps):
        source_image, real_image = batch
        world, dp = self._dp()
        out = self.engine.train_step_rgba(source_image, real_image, self.lambda_l1, lambda_hist=self.lambda_histogram,
                                          global_batch=len(source_image) * world, dp=dp)
        g_loss, d_loss = (out[0], out[1], out[2], out[3]), (out[4], out[5], out[6])
        self._log(g_loss, d_loss, step, update_steps)
        return g_loss, d_loss

    def log_generator_loss(self, g_loss, step):
        """pix2pix_model.py:255-258"""
        super().log_generator_loss(g_loss[:3], step)
        self.summary_writer.scalar("generator/histogram_loss", g_loss[3], step)


class Pix2PixIndexedModel(Pix2PixModel):
    """pix2pix_model.py:261-330"""

    def __init__(self, train_ds, test_ds, model_name, architecture_name, lambda_segmentation=0.5, **kw):
        super().__init__(train_ds, test_ds, model_name, architecture_name, 0.0, **kw)      # lambda_l1 = 0 (:263)
        self.lambda_segmentation = lambda_segmentation

    def create_engine(self):
        # UnetGenerator(1, MAX_PALETTE_SIZE, "softmax") / PatchDiscriminator(1)  (pix2pix_model.py:267-271)
        return Pix2PixEngine(1, MAX_PALETTE_SIZE, "softmax", self._img_size, self._dtype, device=self._device, seed=self._seed)

    def generate(self, batch):
        """pix2pix_model.py:283-287"""
        source_image = batch[0]
        return self.engine.generate_indexed(source_image)

    def generate_with_probs(self, batch):
        """pix2pix_model.py:289-293"""
        source_image = batch[0]
        return self.engine.generate_indexed(source_image, with_probs=True)

    def train_step(self, batch, step, update_steps):
        """pix2pix_model.py:295-325"""
        source_image, real_image, _ = batch
        world, dp = self._dp()
        out = self.engine.train_step_indexed(source_image, real_image, self.lambda_segmentation,
                                             global_batch=len(source_image) * world, dp=dp)
        g_loss, d_loss = (out[0], out[1], out[2], out[3]), (out[4], out[5], out[6])
        self._log(g_loss, d_loss, step, update_steps)
        return g_loss, d_loss

    def log_generator_loss(self, g_loss, step):
        """pix2pix_model.py:327-330"""
        super().log_generator_loss(g_loss[:3], step)
        self.summary_writer.scalar("generator/segmentation_loss", g_loss[3], step)

    def evaluate_l1_batch(self, batch):
        fake = self.generate(batch).to(torch.float32)
        target = torch.as_tensor(batch[1], dtype=torch.float32).to(fake.device)
        return float((target - fake).abs().mean())
