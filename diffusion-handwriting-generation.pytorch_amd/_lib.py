"""ctypes binding of libdhw_hip.so (include/dhw.h, include/dhw_debug.h).

The HIP library IS the product: there is no CPU or PyTorch fallback.  If the
shared object is missing or does not load, every entry point raises
``RuntimeError`` with the build hint.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# DHW_LIB: alternative build of the SAME library (A/B runs of two builds on one GPU box); default = the in-tree build
LIB_PATH = os.environ.get("DHW_LIB") or os.path.join(HERE, "libdhw_hip.so")

DHW_F32, DHW_BF16, DHW_F16, DHW_F64 = 0, 1, 2, 3
PREC_BF16, PREC_F32 = 0, 1


class DhwDims(C.Structure):
    _fields_ = [("num_layers", C.c_int), ("c1", C.c_int), ("c2", C.c_int), ("c3", C.c_int),
                ("max_B", C.c_int), ("max_L", C.c_int), ("max_Lt", C.c_int), ("S", C.c_int),
                ("precision", C.c_int)]


class GemmDesc(C.Structure):   # include/dhw_train.h dhw_gemm_desc
    _fields_ = [("A", C.c_void_p), ("sam", C.c_longlong), ("sak", C.c_longlong), ("sazo", C.c_longlong), ("sazi", C.c_longlong), ("a_shift", C.c_int), ("a_tap_shift", C.c_int),
                ("B", C.c_void_p), ("sbk", C.c_longlong), ("sbn", C.c_longlong), ("sbzo", C.c_longlong), ("sbzi", C.c_longlong), ("sbt", C.c_longlong), ("b_shift", C.c_int), ("b_z_shift", C.c_int),
                ("C", C.c_void_p), ("scm", C.c_longlong), ("scn", C.c_longlong), ("sczo", C.c_longlong), ("sczi", C.c_longlong),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("nzo", C.c_int), ("nzi", C.c_int), ("lr", C.c_int), ("taps", C.c_int),
                ("bias", C.c_void_p), ("alpha", C.c_float), ("accumulate", C.c_int), ("bf16", C.c_int), ("act_out", C.c_void_p), ("addend", C.c_void_p), ("dsilu_of", C.c_void_p), ("rowsum", C.c_void_p),
                ("film_gamma", C.c_void_p), ("film_beta", C.c_void_p), ("film_pstride", C.c_longlong), ("film_rows", C.c_int), ("film_act", C.c_int),
                ("film_out", C.c_void_p), ("film_addend", C.c_void_p)]


class ConvBlockWeights(C.Structure):   # include/dhw_train.h dhw_convblock_weights (HOST pointers) / dhw_convblock_grads (DEVICE pointers)
    _fields_ = [(n, C.c_void_p) for n in ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "fc_w", "fc_b", "skip_w", "skip_b", "film_w", "film_b")]


class DhwError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libdhw_hip error {code}: {msg}")
        self.code = code


_lib = None

# every symbol include/dhw.h and include/dhw_debug.h declare: (restype, argtypes)
_P = C.c_void_p
_LL = C.c_longlong
SIGNATURES = {
    "dhw_create": (C.c_int, [C.POINTER(_P), C.POINTER(DhwDims), C.c_int]),
    "dhw_load": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "dhw_finalize": (C.c_int, [_P]),
    "dhw_num_keys": (C.c_int, [_P]),
    "dhw_key_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "dhw_forward": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "dhw_sample": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_uint64, C.c_int64, _P, _P]),
    "dhw_schedule": (C.c_int, [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "dhw_work": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dhw_last_error": (C.c_char_p, [_P]),
    "dhw_version": (C.c_char_p, []),
    "dhw_destroy": (None, [_P]),
    "dhw_debug_read": (C.c_int64, [_P, C.c_char_p, C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_int64)]),
    "dhw_profile_enable": (C.c_int, [_P, C.c_int]),
    "dhw_profile_reset": (C.c_int, [_P]),
    "dhw_profile_count": (C.c_int, [_P]),
    "dhw_profile_get": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                  C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dhw_set_graph": (C.c_int, [_P, C.c_int]),
    "dhw_debug_persist_plans": (C.c_int, [_P]),
    "dhw_debug_persist_trace": (C.c_int, [_P, _P, C.c_int64]),
    "dhw_debug_set_teacher": (C.c_int, [_P, _P, _P, C.c_int]),
    "dhw_debug_xcd_swizzle": (C.c_int, [C.c_int, C.c_int]),
    "dhw_debug_raise": (C.c_int, [_P, C.c_int]),
    "dhw_debug_attention_time": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), _P]),
    "dhw_set_streams": (C.c_int, [_P, C.c_int]),
    # include/dhw_style.h
    "dhw_style_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int]),
    "dhw_style_load": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "dhw_style_finalize": (C.c_int, [_P]),
    "dhw_style_num_keys": (C.c_int, [_P]),
    "dhw_style_key_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "dhw_style_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "dhw_style_debug_features": (C.c_int64, [_P, C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_int64)]),
    "dhw_style_last_error": (C.c_char_p, [_P]),
    "dhw_style_destroy": (None, [_P]),
    # include/dhw_train.h
    "dhw_train_perturb": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, _P]),
    "dhw_train_loss": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P]),
    "dhw_train_adam": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int64), C.c_float,
                                 C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, _P, _P]),
    "dhw_train_adam_dev": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int64), _P, _P, _P]),
    "dhw_train_draw": (C.c_int, [_P, C.c_int, C.c_int, _P, _LL, C.c_int, C.c_float, _P, _P]),
    "dhw_train_convblock": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.POINTER(ConvBlockWeights), _P, _P, _P,
                                      C.POINTER(ConvBlockWeights), _P]),
    "dhw_train_last_error": (C.c_char_p, []),
    "dhw_op_gemm": (C.c_int, [C.POINTER(GemmDesc), _P]),
    "dhw_op_gemm2": (C.c_int, [C.POINTER(GemmDesc), C.POINTER(GemmDesc), _P]),
    "dhw_op_gemm_group": (C.c_int, [C.POINTER(GemmDesc), C.c_int, _P]),
    "dhw_op_unary": (C.c_int, [C.c_int, _P, _LL, _P, _P]),
    "dhw_op_unary_bwd": (C.c_int, [C.c_int, _P, _P, _LL, _P, C.c_int, _P]),
    "dhw_op_add": (C.c_int, [_P, _P, _LL, _P, C.c_int, _P]),
    "dhw_op_add_rows": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "dhw_op_film": (C.c_int, [_P, _P, _P, _LL, C.c_int, C.c_int, C.c_int, _P, _P]),
    "dhw_op_film_bwd": (C.c_int, [_P, _P, _P, _LL, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P]),
    "dhw_op_film_act": (C.c_int, [_P, _P, _P, _LL, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "dhw_op_film_act_bwd": (C.c_int, [_P, _P, _P, _P, _LL, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P]),
    "dhw_op_ln_film": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _LL, _P, _P, _P, _P, _P, _P, _P, _P]),
    "dhw_op_ln_film_bwd": (C.c_int, [_P, _P, _P, _P, _P, _LL, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P]),
    "dhw_op_layernorm": (C.c_int, [_P, _LL, C.c_int, _P, _P, _P, _P]),
    "dhw_op_layernorm_bwd": (C.c_int, [_P, _P, _P, _LL, C.c_int, _P, C.c_int, _P]),
    "dhw_op_softmax": (C.c_int, [_P, _LL, C.c_int, _LL, _P, C.c_float, _P, _P]),
    "dhw_op_softmax_bwd": (C.c_int, [_P, _P, _LL, C.c_int, C.c_float, _P, _P]),
    "dhw_op_resample": (C.c_int, [C.c_int, _P, _LL, C.c_int, _P, C.c_int, _P]),
    "dhw_op_embedding": (C.c_int, [_P, _P, _LL, C.c_int, _P, _P]),
    "dhw_op_embedding_bwd": (C.c_int, [_P, _P, _LL, C.c_int, _P, _P]),
    "dhw_op_mask_mul": (C.c_int, [_P, _P, C.c_float, _LL, _P, C.c_int, _P]),
    "dhw_op_colsum": (C.c_int, [_P, _LL, C.c_int, _P, _P]),
    "dhw_op_film_table": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P, _P]),
    "dhw_op_film_table_bwd": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "dhw_op_keep_mask": (C.c_int, [_P, C.c_int, _LL, C.c_int, C.c_float, _P, _P]),
    "dhw_debug_randn": (C.c_int, [_P, C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
}


def lib():
    """Load (once) and return the ctypes library; raise loudly if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc, gfx950). There is no fallback path.")
    try:
        l = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise RuntimeError(f"failed to load {LIB_PATH}: {e}. There is no fallback path.") from e
    for name, (res, args) in SIGNATURES.items():
        f = getattr(l, name)
        f.restype = res
        f.argtypes = args
    _lib = l
    return l


def check(code: int, handle=None, style: bool = False):
    if code < 0:
        msg = (lib().dhw_style_last_error if style else lib().dhw_last_error)(handle)
        raise DhwError(code, msg.decode() if msg else "?")
    return code


def schedule(T: int = 60):
    """(beta[T], alpha_bar[T]) as numpy fp32 — host-only, works without a GPU."""
    import numpy as np
    beta = np.zeros(T, np.float32)
    alpha = np.zeros(T, np.float32)
    check(lib().dhw_schedule(T, beta.ctypes.data_as(C.POINTER(C.c_float)), alpha.ctypes.data_as(C.POINTER(C.c_float))))
    return beta, alpha
