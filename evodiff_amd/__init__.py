"""MI355X-native sampler for EvoDiff's ByteNetLMTime denoiser.

Mirrors the reference's Python surface for this one path — ``evodiff.pretrained`` sequence loaders and
``evodiff.generate.generate_oaardm`` / ``generate_d3pm`` — and runs it in ``libevodiff_hip.so``
(hand-written gfx950 kernels behind the C ABI of ``include/evodiff_hip.h``).  There is no CPU path:
calling the model without the library or without a GPU raises.
"""
from . import constants, utils, model, pretrained, generate  # noqa: F401

__all__ = ["constants", "utils", "model", "pretrained", "generate"]
