"""ctypes binding of ``liballophant_amx.so`` (C ABI declared in ``include/allophant_amx.h``).

There is deliberately no fallback: if the HIP library has not been built (``python -c 'import __graft_entry__ as g;
g.build()'`` or ``make -C allophant_amd/csrc``) every compute entry point raises ``RuntimeError``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

# AMX_ABI_OVERRIDE: developer switch for same-box A/B runs against a library built from an OLDER revision (tools/ab_build.sh):
# ABI 5 and 6 added flags and entry points to ABI 4 without changing a struct, so an older build runs under this binding
AMX_ABI_VERSION = int(os.environ.get("AMX_ABI_OVERRIDE") or 6)
AMX_MAX_CONV = 8
AMX_MAX_DEPS = 64
AMX_NAME_LEN = 48

AMX_OK, AMX_EINVAL, AMX_EHIP, AMX_ESTATE, AMX_ENOMEM, AMX_ERANGE = 0, -1, -2, -3, -4, -5
PRECISIONS = {"bf16": 0, "f16": 1, "bf16x3": 2, "f16x3": 3}
FLAG_HOST_IO, FLAG_RAW_LOGITS, FLAG_KEEP_HIDDEN, FLAG_TIMING, FLAG_PADDED, FLAG_NO_PACK, FLAG_CONTINUE = 1, 2, 4, 8, 16, 32, 64
FLAG_NO_GRAPH, FLAG_NO_RANGE_CHECK = 128, 256
NORM_LAYER, NORM_GROUP = 0, 1
PASS_INFO = ["ln_fold", "packed", "graph", "rows", "id"]  # AMX_PASS_INFO_*
KERNEL_CLASSES = ["gemm_pp", "gemm_tile", "attention", "rownorm", "conv0", "other", "gemm_ln", "conv_tail"]
DEP_OUTPUT = -1

LIB_NAME = "liballophant_amx.so"
# AMX_LIB_PATH: developer switch (A/B builds of the library side by side); the default is the in-tree build
LIB_PATH = os.environ.get("AMX_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

# every symbol include/allophant_amx.h declares
EXPORTS = [
    "amx_create", "amx_destroy", "amx_last_error", "amx_set_inventory", "amx_output_layout", "amx_forward",
    "amx_synchronize", "amx_greedy_ctc", "amx_debug_fetch", "amx_device_bytes", "amx_timing_fetch",
    "amx_max_utterances", "amx_greedy_ctc_emissions", "amx_check_finite", "amx_gather_outputs", "amx_dist_last_error",
    "amx_graph_info", "amx_pass_info",
]


def dep_output_layer(i: int) -> int:
    return -2 - i


class AmxConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("n_conv", C.c_int32), ("conv_dim", C.c_int32),
        ("conv_kernel", C.c_int32 * AMX_MAX_CONV), ("conv_stride", C.c_int32 * AMX_MAX_CONV),
        ("hidden", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32), ("ffn", C.c_int32),
        ("pos_kernel", C.c_int32), ("pos_groups", C.c_int32), ("eps", C.c_float), ("do_normalize", C.c_int32),
        ("dependency_blanks", C.c_int32), ("embedding_size", C.c_int32), ("allophone_layer", C.c_int32),
        ("precision", C.c_int32), ("feat_extract_norm", C.c_int32), ("conv_bias", C.c_int32),
        ("stable_layer_norm", C.c_int32), ("use_attention_mask", C.c_int32),
    ]


class AmxClassDesc(C.Structure):
    _fields_ = [
        ("name", C.c_char * AMX_NAME_LEN), ("size", C.c_int32), ("out_features", C.c_int32), ("n_deps", C.c_int32),
        ("deps", C.c_int32 * AMX_MAX_DEPS), ("time_heads", C.c_int32), ("time_positional", C.c_int32),
    ]


class AmxTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("numel", C.c_int64)]


class AmxOutputDesc(C.Structure):
    _fields_ = [("name", C.c_char * AMX_NAME_LEN), ("classes", C.c_int32), ("offset", C.c_int64)]


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Loads the shared library and declares the prototypes; raises if it is missing (no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP extension first (make -C allophant_amd/csrc, or "
            "__graft_entry__.build()). allophant_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    lib.amx_create.argtypes = [C.POINTER(vp), i32, C.POINTER(AmxConfig), C.POINTER(AmxClassDesc), i32,
                               C.POINTER(AmxTensor), i32]
    lib.amx_create.restype = i32
    lib.amx_destroy.argtypes = [vp]
    lib.amx_destroy.restype = i32
    lib.amx_last_error.argtypes = [vp]
    lib.amx_last_error.restype = C.c_char_p
    lib.amx_set_inventory.argtypes = [vp, C.POINTER(i64), i32, i32, C.POINTER(i64), vp]
    lib.amx_set_inventory.restype = i32
    lib.amx_output_layout.argtypes = [vp, i32, i64, C.POINTER(AmxOutputDesc), C.POINTER(i32), C.POINTER(i64),
                                      C.POINTER(i64)]
    lib.amx_output_layout.restype = i32
    lib.amx_forward.argtypes = [vp, vp, C.POINTER(i64), i32, i64, vp, C.POINTER(i64), C.c_uint32, vp]
    lib.amx_forward.restype = i32
    lib.amx_synchronize.argtypes = [vp, vp]
    lib.amx_synchronize.restype = i32
    if hasattr(lib, "amx_graph_info") or AMX_ABI_VERSION >= 5:  # (absent from an ABI-4 build under AMX_ABI_OVERRIDE)
        lib.amx_graph_info.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
        lib.amx_graph_info.restype = i32
    if hasattr(lib, "amx_pass_info") or AMX_ABI_VERSION >= 6:  # (absent from older builds under AMX_ABI_OVERRIDE)
        lib.amx_pass_info.argtypes = [vp, C.POINTER(C.c_int32), i32]
        lib.amx_pass_info.restype = i32
    lib.amx_check_finite.argtypes = [vp, vp, C.POINTER(i64)]
    lib.amx_check_finite.restype = i32
    lib.amx_greedy_ctc.argtypes = [vp, vp, C.POINTER(i64), i32, i64, vp, vp, vp, vp, vp]
    lib.amx_greedy_ctc.restype = i32
    lib.amx_debug_fetch.argtypes = [vp, i32, i32, vp, i64, C.POINTER(i64)]
    lib.amx_debug_fetch.restype = i32
    lib.amx_timing_fetch.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int32), i32]
    lib.amx_timing_fetch.restype = i32
    lib.amx_device_bytes.argtypes = [vp]
    lib.amx_device_bytes.restype = i64
    lib.amx_max_utterances.argtypes = [vp, i64]
    lib.amx_max_utterances.restype = i64
    lib.amx_greedy_ctc_emissions.argtypes = [i32, vp, i64, i64, vp, i32, i64, i32, i32, vp, vp, vp, vp, vp]
    lib.amx_greedy_ctc_emissions.restype = i32
    lib.amx_gather_outputs.argtypes = [vp, i32, i32, i32, vp, i64, vp, vp, i32, vp, vp]
    lib.amx_gather_outputs.restype = i32
    lib.amx_dist_last_error.argtypes = []
    lib.amx_dist_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def check(lib: C.CDLL, handle, code: int) -> None:
    """Maps C status codes onto the exception types the reference raises (ValueError for configuration / argument
    problems, RuntimeError otherwise)."""
    if code == AMX_OK:
        return
    message = lib.amx_last_error(handle)
    message = message.decode() if message else f"liballophant_amx error {code}"
    if code == AMX_EINVAL:
        raise ValueError(message)
    if code == AMX_ENOMEM:
        raise MemoryError(message)
    if code == AMX_ERANGE:
        raise FloatingPointError(message)
    raise RuntimeError(message)
