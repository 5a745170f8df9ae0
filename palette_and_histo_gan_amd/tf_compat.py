"""Stand-in for the three direct TensorFlow calls of experiments.ipynb (cells 1 and 3: `tf.__version__`,
`tf.test.gpu_device_name()`, `tf.random.set_seed(SEED)`), so the notebook's cells run against this build with only their
imports changed (examples/experiments.py runs the same workflow):

    from palette_and_histo_gan_amd.tf_compat import tf

`tf.random.set_seed(s)` sets the default seed of everything created afterwards that takes `seed=None`: the dataset shuffles
and augmentation draws (dataset_utils.load_rgba_ds / load_indexed_ds) and the models' weight initialisation and dropout
streams (pix2pix_model.Pix2Pix*Model).  The random streams are numpy's and the device's, not TensorFlow's.
"""
from .configuration import SEED

_state = {"seed": SEED}


def global_seed():
    return _state["seed"]


class _Random:
    @staticmethod
    def set_seed(seed):
        _state["seed"] = int(seed)


class _Test:
    @staticmethod
    def gpu_device_name():
        """'/device:GPU:0' when a HIP device is visible, else '' (the notebook only tests truthiness)"""
        import torch
        return "/device:GPU:0" if torch.cuda.is_available() else ""

    @staticmethod
    def is_gpu_available():
        return bool(_Test.gpu_device_name())


class _TF:
    random = _Random()
    test = _Test()

    @property
    def __version__(self):
        import torch
        from . import _lib as L
        return f"p2pgan-mi355x (libp2pgan_hip v{L.lib().p2p_version()}, torch {torch.__version__})"


tf = _TF()
