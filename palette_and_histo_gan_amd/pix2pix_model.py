"""The Pix2Pix model family with the reference's class surface (pix2pix_model.py:11-325) on the MI355X engine.

    Pix2PixModel(train_ds, test_ds, model_name, architecture_name, lambda_l1)
    Pix2PixAugmentedModel(same)                  -- augmentation lives in the dataset (pix2pix_model.py:232-234)
    Pix2PixHistogramModel(..., lambda_l1, lambda_histogram)
    Pix2PixIndexedModel(train_ds, test_ds, model_name, architecture_name, lambda_segmentation=0.5)

The constructor follows the reference's (pix2pix_model.py:12-36): create_generator() / create_discriminator() with the
reference's builder signatures, `loss_object`, two Adam(0.0002, beta_1=0.5) optimizers, `checkpoint`,
`checkpoint_manager`.  train_step(batch, step, update_steps) keeps the reference's contract (one optimisation step of
both networks with gradients taken at the same pre-update weights, scalars logged at step // update_steps) and
additionally RETURNS (g_loss_tuple, d_loss_tuple) as device scalars, and tolerates summary_writer=None.

The step is FUSED: generator_loss / discriminator_loss of the classes below are evaluated by the HIP kernels inside
engine.train_step_* together with their gradients (there is no autograd tape here).  The methods exist with the
reference's signatures for standalone evaluation, but a SUBCLASS that overrides one of them cannot change the fused step;
train_step detects that and raises instead of silently ignoring the override.
Extra keyword arguments (dtype, img_size, device, data_parallel, seed) are build-added; defaults reproduce the reference.

Data parallelism (build-added): with `data_parallel`, every rank agrees on the same GLOBAL batch (same
dataset seed, same order) and works on its contiguous shard (parallel.shard_bounds).  The sprite datasets of dataset_utils
are told their shard (set_shard) and MATERIALISE ONLY THAT SHARE of every batch (dataset_utils.ShardedBatch carries the global
batch size and the shard's offset); any other batch source is taken as the whole global batch and sliced here.  Loss
denominators use the global batch, the dropout stream is keyed by the global sample index, gradients are summed over ranks.
Ragged global batches and ranks with an empty shard are handled.
"""
import torch

from . import _lib as L
from . import histogram as _histogram  # noqa: F401  (module parity with the reference's `import histogram`)
from .configuration import IMG_SIZE, MAX_PALETTE_SIZE
from .engine import Pix2PixEngine
from .networks import PatchDiscriminator, UnetGenerator
from .dataset_utils import ShardedBatch
from .parallel import shard_bounds
from .side2side_model import Checkpoint, CheckpointManager, S2SModel


class BinaryCrossentropy:
    """tf.keras.losses.BinaryCrossentropy(from_logits=True) (pix2pix_model.py:19) for standalone evaluation: mean over all
    elements of max(x,0) - x*z + log1p(exp(-|x|))."""

    def __init__(self, from_logits=True):
        if not from_logits:
            raise NotImplementedError("the reference only uses from_logits=True")

    def __call__(self, y_true, y_pred):
        x = torch.as_tensor(y_pred, dtype=torch.float32)
        z = torch.as_tensor(y_true, dtype=torch.float32).to(x.device)
        return torch.nn.functional.binary_cross_entropy_with_logits(x, z)


class CategoricalCrossentropy:
    """tf.keras.losses.CategoricalCrossentropy(from_logits=False) (pix2pix_model.py:265) for standalone evaluation.
    Keras 2.9 evaluates it on the logits cached by the softmax activation (softmax_cross_entropy_with_logits) -- that is what
    train_step's fused kernel computes (log-sum-exp form, csrc/softmax.hip) and what `logits=` selects here.  Handed
    probabilities alone, Keras' documented fallback applies: p /= sum p; p = clip(p, 1e-7, 1 - 1e-7); mean of -sum t log p.
    The two agree to < 1e-6 unless a target probability is below 1e-7, where the fallback is capped at -log 1e-7 = 16.1 per
    pixel and the logits form is not."""

    def __call__(self, y_true, y_pred, logits=None):
        if logits is not None:
            z = torch.as_tensor(logits, dtype=torch.float32)
            t = torch.as_tensor(y_true, dtype=torch.float32).to(z.device)
            return -(t * torch.log_softmax(z, dim=-1)).sum(-1).mean()
        p = torch.as_tensor(y_pred, dtype=torch.float32)
        t = torch.as_tensor(y_true, dtype=torch.float32).to(p.device)
        p = p / p.sum(-1, keepdim=True)
        p = p.clamp(1e-7, 1.0 - 1e-7)
        return -(t * p.log()).sum(-1).mean()


class Adam:
    """tf.keras.optimizers.Adam(learning_rate, beta_1) as the reference constructs it (pix2pix_model.py:28-29): the
    hyper-parameters live here, the moments and the step count in the engine's flat buffers (engine.ParamStore)."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon
        self._store = None

    @property
    def iterations(self):
        return 0 if self._store is None else self._store.t


class Pix2PixModel(S2SModel):
    def __init__(self, train_ds, test_ds, model_name, architecture_name, lambda_l1, dtype="bf16", img_size=IMG_SIZE,
                 device="cuda:0", data_parallel=None, seed=None):
        super().__init__(train_ds, test_ds, model_name, architecture_name)
        if seed is None:
            from .tf_compat import global_seed      # configuration.SEED unless the notebook's tf.random.set_seed changed it
            seed = global_seed()
        self.lambda_l1 = lambda_l1
        self._dtype = {"bf16": L.BF16, "f32": L.F32}[dtype] if isinstance(dtype, str) else dtype
        self._img_size, self._device, self._seed = img_size, device, seed
        self.data_parallel = data_parallel
        if data_parallel is not None and hasattr(train_ds, "set_shard"):
            train_ds.set_shard(data_parallel.rank, data_parallel.world)      # produce this rank's rows only

        self.generator = self.create_generator()
        self.discriminator = self.create_discriminator()
        self.loss_object = BinaryCrossentropy(from_logits=True)
        self.generator_optimizer = Adam(0.0002, beta_1=0.5)
        self.discriminator_optimizer = Adam(0.0002, beta_1=0.5)
        self.engine = self.create_engine()

        print(f"Generator: {self.generator.name} with {self.generator.count_params():,} parameters")
        print(f"Discriminator: {self.discriminator.name} with {self.discriminator.count_params():,} parameters")

        self.checkpoint = Checkpoint(generator_optimizer=self.generator_optimizer,
                                     discriminator_optimizer=self.discriminator_optimizer,
                                     generator=self.generator, discriminator=self.discriminator, engine=self.engine)
        self.checkpoint_manager = CheckpointManager(self.checkpoint, directory=self.checkpoint_dir, max_to_keep=1)
        self._hooks_checked = False
        self._custom_hooks = False

    # -- construction hooks (pix2pix_model.py:38-42) ------------------------------------------------------------
    def create_generator(self):
        return UnetGenerator(4, 4, "tanh")

    def create_discriminator(self):
        return PatchDiscriminator(4)

    def create_engine(self):
        """build-added: one device engine for both networks, from the architecture the two builders recorded"""
        g, d = self.generator, self.discriminator
        if d.input_channels != g.input_channels:
            raise ValueError("the discriminator sees [target, source] images of the generator's input channel count")
        eng = Pix2PixEngine(g.input_channels, g.output_channels, g.last_activation, self._img_size, self._dtype,
                            device=self._device, seed=self._seed)
        go, do = self.generator_optimizer, self.discriminator_optimizer
        if (go.learning_rate, go.beta_1, go.beta_2, go.epsilon) != (do.learning_rate, do.beta_1, do.beta_2, do.epsilon):
            raise NotImplementedError("both optimizers share one set of Adam hyper-parameters (as in the reference)")
        eng.lr, eng.beta1, eng.beta2, eng.adam_eps = go.learning_rate, go.beta_1, go.beta_2, go.epsilon
        g.bind(eng, eng.G)
        d.bind(eng, eng.D)
        go._store, do._store = eng.G, eng.D
        return eng

    # -- losses as standalone evaluations (pix2pix_model.py:44-56); train_step computes them fused ----------------
    def generator_loss(self, fake_predicted, fake_image, real_image):
        fp = torch.as_tensor(fake_predicted, dtype=torch.float32)
        adversarial_loss = self.loss_object(torch.ones_like(fp), fp)
        real = torch.as_tensor(real_image, dtype=torch.float32).to(fp.device)
        fake = torch.as_tensor(fake_image, dtype=torch.float32).to(fp.device)
        l1_loss = (real - fake).abs().mean()
        total_loss = adversarial_loss + (self.lambda_l1 * l1_loss)
        return total_loss, adversarial_loss, l1_loss

    def discriminator_loss(self, real_predicted, fake_predicted):
        rp = torch.as_tensor(real_predicted, dtype=torch.float32)
        fp = torch.as_tensor(fake_predicted, dtype=torch.float32)
        real_loss = self.loss_object(torch.ones_like(rp), rp)
        fake_loss = self.loss_object(torch.zeros_like(fp), fp)
        total_loss = fake_loss + real_loss
        return total_loss, real_loss, fake_loss

    def generate(self, batch):
        """pix2pix_model.py:58-60"""
        source_image, _ = batch
        return self.generator(source_image, training=True)

    # -- the hot path ----------------------------------------------------------------------------------------------
    def _check_hooks(self):
        """Which train step serves this class?  The reference's own loss sets (Pix2PixModel / Pix2PixHistogramModel /
        Pix2PixIndexedModel) run fused: losses and gradients in one kernel sequence, no tape.  A subclass that OVERRIDES
        generator_loss / discriminator_loss (the hooks are part of the boundary) gets `engine.train_step_rgba_hooked`:
        the kernels run forward, the hooks are evaluated on torch tensors with autograd, their gradients enter the backward kernels.
        The palette-index model's step is fused around its softmax head and argmax: overriding ITS hooks is refused."""
        if self._hooks_checked:
            return
        known = (Pix2PixModel, Pix2PixHistogramModel, Pix2PixIndexedModel)
        custom = []
        for hook in ("generator_loss", "discriminator_loss"):
            owner = next((c for c in type(self).__mro__ if hook in c.__dict__), None)
            if owner not in known:
                custom.append(hook)
        if custom and isinstance(self, Pix2PixIndexedModel):
            raise NotImplementedError(
                f"{type(self).__name__}.{custom[0]} overrides the reference's loss, but the palette-index train_step is one fused "
                f"sequence of HIP kernels around the softmax head and its argmax (no gradient path from the discriminator, "
                f"pix2pix_model.py:295-325): add the loss to engine.train_step_indexed or subclass train_step.")
        if custom and self.data_parallel is not None:
            raise NotImplementedError("overridden loss hooks run on one GPU (the hooked step issues no collectives)")
        self._custom_hooks = bool(custom)
        self._hooks_checked = True

    def _shard(self, batch, tensors):
        """(local shard of every tensor, global batch, samples in front of the shard, DataParallel or None)"""
        dp = self.data_parallel
        if isinstance(batch, ShardedBatch):       # the dataset produced this rank's rows only
            return tensors, batch.global_batch, batch.offset, dp
        Bg = len(tensors[0])
        if dp is None:
            return self._upload(tensors), Bg, 0, None
        lo, hi = shard_bounds(Bg, dp.world, dp.rank)
        return self._upload([t[lo:hi] for t in tensors]), Bg, lo, dp

    def _upload(self, tensors):
        """host batches (numpy / CPU tensors) go up on a copy stream, beside the previous step's kernels (dataset_utils.upload_async)"""
        if len(tensors[0]) == 0 or self.engine.device.type != "cuda":
            return tensors
        from .dataset_utils import upload_async
        return upload_async(tensors, self.engine.device)

    def train_step(self, batch, step, update_steps):
        """pix2pix_model.py:62-89"""
        self._check_hooks()
        source_image, real_image = batch
        (src, real), Bg, lo, dp = self._shard(batch, [source_image, real_image])
        if self._custom_hooks:
            out = self.engine.train_step_rgba_hooked(src, real, self.generator_loss, self.discriminator_loss)
        elif len(src) == 0:
            out = self.engine.train_step_empty(self.lambda_l1, dp=dp)
        else:
            out = self.engine.train_step_rgba(src, real, self.lambda_l1, global_batch=Bg, dp=dp, batch_offset=lo)
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
        train_examples = self._whole(self.train_ds).unbatch().take(num_train_examples).batch(1)
        test_examples = self._whole(self.test_ds).unbatch().take(num_test_examples).batch(1)
        return list(test_examples.as_numpy_iterator()) + list(train_examples.as_numpy_iterator())

    def evaluate_l1_batch(self, batch):
        source, target = batch[0], batch[1]
        fake = self.generate((source, target))
        return (torch.as_tensor(target, dtype=torch.float32).to(fake.device) - fake).abs().mean()


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

    def discriminator_loss(self, real_predicted, fake_predicted):
        return super().discriminator_loss(real_predicted, fake_predicted)

    def train_step(self, batch, step, update_steps):
        self._check_hooks()
        source_image, real_image = batch
        (src, real), Bg, lo, dp = self._shard(batch, [source_image, real_image])
        if self._custom_hooks:
            out = self.engine.train_step_rgba_hooked(src, real, self.generator_loss, self.discriminator_loss)
        elif len(src) == 0:
            out = self.engine.train_step_empty(self.lambda_l1, lambda_hist=self.lambda_histogram, dp=dp)
        else:
            out = self.engine.train_step_rgba(src, real, self.lambda_l1, lambda_hist=self.lambda_histogram,
                                              global_batch=Bg, dp=dp, batch_offset=lo)
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
        super().__init__(train_ds, test_ds, model_name, architecture_name, 0., **kw)      # lambda_l1 = 0 (:263)
        self.lambda_segmentation = lambda_segmentation
        self.segmentation_loss_object = CategoricalCrossentropy()

    def create_generator(self):
        return UnetGenerator(1, MAX_PALETTE_SIZE, "softmax")

    def create_discriminator(self):
        return PatchDiscriminator(1)

    def generator_loss(self, fake_predicted, fake_image, real_image):
        """pix2pix_model.py:273-278 (fake_image = probabilities, real_image = one-hot)"""
        segmentation_loss = self.segmentation_loss_object(real_image, fake_image)
        total_loss, adversarial_loss, l1_loss = super().generator_loss(fake_predicted, fake_image, real_image)
        total_loss = total_loss + self.lambda_segmentation * segmentation_loss
        return total_loss, adversarial_loss, l1_loss, segmentation_loss

    def discriminator_loss(self, real_predicted, fake_predicted):
        return super().discriminator_loss(real_predicted, fake_predicted)

    def generate(self, batch):
        """pix2pix_model.py:283-287"""
        source_image = batch[0]
        return self.engine.generate_indexed(source_image)

    def debug_discriminator_patches(self, batch_of_one):
        """pix2pix_model.py:372-431: as the base class, on index images (the discriminator of this model sees indices)"""
        source, real = batch_of_one[0], batch_of_one[1]
        fake = self.generate(batch_of_one)
        out = {}
        for name, img in (("real", real), ("fake", fake)):
            logits = self.discriminator([img, source], training=True)
            prob = torch.sigmoid(logits[0, :, :, 0].to(torch.float32)).cpu().numpy()
            out[name], out[name + "_mean"] = prob, float(prob.mean())
        return out

    def generate_with_probs(self, batch):
        """pix2pix_model.py:289-293"""
        source_image = batch[0]
        return self.engine.generate_indexed(source_image, with_probs=True)

    def train_step(self, batch, step, update_steps):
        """pix2pix_model.py:295-325"""
        self._check_hooks()
        source_image, real_image, _ = batch
        (src, real), Bg, lo, dp = self._shard(batch, [source_image, real_image])
        if len(src) == 0:
            out = self.engine.train_step_empty(0.0, lambda_aux=self.lambda_segmentation, dp=dp)
        else:
            out = self.engine.train_step_indexed(src, real, self.lambda_segmentation, global_batch=Bg, dp=dp, batch_offset=lo)
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
        return (target - fake).abs().mean()
