"""wvhash -- MI355X-native wavelet-hashing retrieval hot path.

Mirrors the plugin namespaces of ArseneAmoya/image-retrieval-wavelet (``main.transforms``,
``main.engine``, ``main.models``) for the transform / ranking / attention-head path only; all
arithmetic runs in libwvhash.so (HIP, gfx950) through the C ABI in include/wvhash.h.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
