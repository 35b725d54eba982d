"""The thin PyTorch-ROCm C++ extension over the C ABI: ``torch.ops.evodiff_amd.{forward, oadm_sample_, d3pm_sample_,
score_nll}`` (evodiff_amd/csrc/torch_ops.cpp).  The operators take torch tensors, run on PyTorch's current HIP stream
and raise ``RuntimeError`` on failure; ``handle`` is ``int(model.hip_handle(device).value)``.  The ctypes binding
(``evodiff_amd._lib``) is the torch-free way to the same entry points; both go through ``libevodiff_hip.so``."""
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libevodiff_torch_ops.so")
_loaded = False


def load():
    """Register the operators (once); raises if the extension has not been built (``__graft_entry__.build()``)."""
    global _loaded
    if not _loaded:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        torch.ops.load_library(LIB_PATH)
        _loaded = True
    return torch.ops.evodiff_amd


def handle_of(model, device):
    """The integer the operators take as ``handle``: the model's C-ABI handle on ``device``."""
    return int(model.hip_handle(device).value)
