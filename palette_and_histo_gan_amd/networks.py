"""Handles with the call surface of the reference's Keras models (networks.py:39-98) over the device engine.

The reference builds two independent tf.keras.Model objects; here the generator and the discriminator of one model
share a Pix2PixEngine (one HBM plan, one kernel library), so these builders return light handles bound to it.
"""
from . import engine as E


class _Handle:
    def __init__(self, engine, store, name):
        self._engine, self._store, self.name = engine, store, name

    @property
    def trainable_variables(self):
        """f32 device views of the flat parameter buffer, Keras variable order."""
        return [self._store.view(self._store.params, k) for k in self._store.shapes]

    trainable_weights = trainable_variables

    def count_params(self):
        return self._store.count()

    def get_weights(self):
        return self._store.export()

    def set_weights(self, values):
        self._store.load(values)
        self._engine.refresh_weight_copies()


class UnetGeneratorHandle(_Handle):
    """UnetGenerator(input_channels, output_channels, last_activation) (networks.py:53-98)."""

    def __call__(self, source_image, training=True):
        # the reference passes training=True everywhere (pix2pix_model.py:60,67): dropout is always on
        if self._engine.head == "softmax":
            return self._engine.generate_indexed(source_image, with_probs=True)[1]
        return self._engine.generate(source_image)


class PatchDiscriminatorHandle(_Handle):
    """PatchDiscriminator(input_channels) (networks.py:39-50); called as D([target, source], training=True)."""

    def __call__(self, inputs, training=True):
        target_image, source_image = inputs
        return self._engine.discriminate(target_image, source_image)


def UnetGenerator(engine):
    return UnetGeneratorHandle(engine, engine.G, "unet-gen")


def PatchDiscriminator(engine):
    return PatchDiscriminatorHandle(engine, engine.D, "patch-disc")
