"""MI355X-native engine for the Pix2Pix side2side training step of fegemo/palette-and-histo-gan.

The package mirrors the reference's Python module names (configuration, networks, histogram,
side2side_model, pix2pix_model) on top of hand-written gfx950 HIP kernels reached through the C ABI in
include/p2pgan.h.  Importing the package does not need a GPU; constructing a model does.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
