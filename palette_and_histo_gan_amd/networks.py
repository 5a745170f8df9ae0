"""Builders with the reference's signatures (networks.py:39,53) over the device engine.

    UnetGenerator(input_channels, output_channels, last_activation)      networks.py:53-98
    PatchDiscriminator(input_channels)                                   networks.py:39-50

The reference builds two independent tf.keras.Model objects.  Here both networks of a model live in one
Pix2PixEngine (one HBM plan, one kernel library), so a builder returns a light handle that records the architecture
arguments; the model constructor (pix2pix_model.py) creates the engine from the two handles and binds them to it.
A bound handle has the call surface the reference uses: `generator(source, training=True)`,
`discriminator([target, source], training=True)`, `.name`, `.trainable_variables` / `.trainable_weights`,
`.count_params()`, `.get_weights()` / `.set_weights()` (Keras variable order, Keras layouts: HWIO for Conv2D,
(kh, kw, Cout, Cin) for Conv2DTranspose).
"""
from collections import OrderedDict

import numpy as np


class _Handle:
    def __init__(self, name):
        self.name = name
        self._engine = self._store = None

    def bind(self, engine, store):
        self._engine, self._store = engine, store
        return self

    def _bound(self):
        if self._engine is None:
            raise RuntimeError(f"{self.name}: not bound to an engine yet (a Pix2Pix*Model binds it in its constructor)")
        return self._engine

    @property
    def trainable_variables(self):
        """f32 device views of the flat parameter buffer, Keras variable order."""
        self._bound()
        return [self._store.view(self._store.params, k) for k in self._store.shapes]

    trainable_weights = trainable_variables

    def count_params(self):
        self._bound()
        return self._store.count()

    def get_weights(self):
        """OrderedDict name -> numpy array, Keras variable order and layouts."""
        self._bound()
        return self._store.export()

    def set_weights(self, values):
        """dict name -> array, or a list in Keras variable order (what keras.Model.get_weights() returns)."""
        self._bound()
        if not isinstance(values, dict):
            values = list(values)
            if len(values) != len(self._store.shapes):
                raise ValueError(f"{self.name}: expected {len(self._store.shapes)} arrays, got {len(values)}")
            values = OrderedDict(zip(self._store.shapes, values))
        for k, shape in self._store.shapes.items():
            if tuple(np.shape(values[k])) != tuple(shape):
                raise ValueError(f"{self.name}: {k} has shape {np.shape(values[k])}, expected {tuple(shape)}")
        self._store.load(values)
        self._engine.refresh_weight_copies()


class UnetGeneratorHandle(_Handle):
    def __init__(self, input_channels, output_channels, last_activation):
        super().__init__("unet-gen")
        if last_activation not in ("tanh", "softmax"):
            raise ValueError("last_activation must be 'tanh' or 'softmax' (the two heads the reference uses)")
        self.input_channels, self.output_channels, self.last_activation = input_channels, output_channels, last_activation

    def __call__(self, source_image, training=True):
        # the reference passes training=True everywhere (pix2pix_model.py:60,67): dropout is always on
        eng = self._bound()
        if eng.head == "softmax":
            return eng.generate_indexed(source_image, with_probs=True)[1]
        return eng.generate(source_image)


class PatchDiscriminatorHandle(_Handle):
    def __init__(self, input_channels):
        super().__init__("patch-disc")
        self.input_channels = input_channels

    def __call__(self, inputs, training=True):
        target_image, source_image = inputs
        return self._bound().discriminate(target_image, source_image)


def UnetGenerator(input_channels, output_channels, last_activation):
    """networks.py:53"""
    return UnetGeneratorHandle(input_channels, output_channels, last_activation)


def PatchDiscriminator(input_channels):
    """networks.py:39"""
    return PatchDiscriminatorHandle(input_channels)
