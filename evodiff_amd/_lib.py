"""ctypes binding of libevodiff_hip.so (the C ABI declared in include/evodiff_hip.h)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# $EVODIFF_AMD_LIB: another build of the same library (A/B comparisons of kernel changes on one GPU box)
LIB_PATH = os.environ.get("EVODIFF_AMD_LIB") or os.path.join(_HERE, "csrc", "libevodiff_hip.so")

PREC_FP32, PREC_BF16 = 0, 1
K_EMBED, K_LNACT, K_GEMM1X1, K_CONV, K_FINAL, K_CONV_OTHER, K_COUNT = 0, 1, 2, 3, 4, 5, 6
KIND_NAMES = ("embed", "ln_act", "gemm1x1", "conv", "final", "conv_other")
OPTIONS = {"prune": 0, "min_tuned_tiles": 1, "split_target": 2, "incremental": 3, "fuse_reduce_ln": 4,
           "split_min_iters": 5, "partials_mb": 6, "tile64_max": 7,
           "tuned_split_min_rows": 8, "fuse_ln": 9, "ar_gemv": 11, "ar_gemv_wgs": 12}   # ED_OPT_* in include/evodiff_hip.h

# every symbol include/evodiff_hip.h declares
SYMBOLS = ("ed_abi_version", "ed_last_error", "ed_model_create", "ed_model_destroy", "ed_model_load_tensor",
           "ed_model_commit", "ed_model_num_params", "ed_forward", "ed_forward_collect", "ed_score_nll", "ed_ar_sample", "ed_oadm_sample",
           "ed_d3pm_sample", "ed_model_set_option", "ed_probe_config", "ed_probe_read", "ed_debug_gemm",
           "ed_debug_ln_act", "ed_debug_set_stamp_buffer", "ed_debug_gemm_splitk_noalloc", "ed_model_check_errors", "ed_model_get_config", "ed_ar_sample_scaffold")


class EdConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_tokens", "d_embed", "d_model", "d_hidden", "n_layers", "kernel_size",
                                         "r", "timesteps", "activation", "causal")]


class EdProbeStat(C.Structure):
    _fields_ = [("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double), ("launches", C.c_int64)]


class EvoDiffHipError(RuntimeError):
    pass


_lib = None


def load():
    """Load the library once; there is no fallback if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EvoDiffHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C evodiff_amd/csrc`). evodiff_amd has no CPU path.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    lib.ed_abi_version.restype = i32
    lib.ed_last_error.restype = C.c_char_p
    lib.ed_model_create.argtypes = [C.POINTER(EdConfig), i32, C.POINTER(vp)]
    lib.ed_model_destroy.argtypes = [vp]
    lib.ed_model_destroy.restype = None
    lib.ed_model_load_tensor.argtypes = [vp, C.c_char_p, vp, i64]
    lib.ed_model_commit.argtypes = [vp]
    lib.ed_model_num_params.argtypes = [vp]
    lib.ed_model_num_params.restype = i64
    lib.ed_forward.argtypes = [vp, vp, vp, i32, i32, vp, i32, vp]
    lib.ed_forward_collect.argtypes = [vp, vp, vp, i32, i32, vp, vp, i32, vp]
    lib.ed_ar_sample.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, i32, i32, vp]
    lib.ed_ar_sample_scaffold.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, i32, vp]
    lib.ed_score_nll.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp, i32, vp]
    lib.ed_oadm_sample.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, i32, vp]
    lib.ed_d3pm_sample.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, i32, vp]
    lib.ed_debug_gemm.argtypes = [vp, i64, i64, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.ed_debug_gemm_splitk_noalloc.argtypes = [vp, i64, i64, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32,
                                                 i32, vp]
    lib.ed_debug_ln_act.argtypes = [vp, i32, vp, i32, vp, vp, i32, i32, i32, vp]
    lib.ed_model_check_errors.argtypes = [vp, vp, i32]
    lib.ed_model_get_config.argtypes = [vp, C.POINTER(EdConfig)]
    lib.ed_probe_config.argtypes = [vp, i32]
    lib.ed_debug_set_stamp_buffer.argtypes = [vp, i64]
    lib.ed_model_set_option.argtypes = [vp, i32, i32]
    lib.ed_probe_read.argtypes = [vp, C.POINTER(EdProbeStat), vp]
    if lib.ed_abi_version() != 1:
        raise EvoDiffHipError("libevodiff_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise EvoDiffHipError(f"libevodiff_hip error {rc}: {load().ed_last_error().decode()}")
