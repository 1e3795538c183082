"""ctypes binding of liblob.so (include/lob.h).  Fails loudly when the library is absent:
there is no CPU fallback anywhere in the product path."""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LOB_LIB_PATH") or os.path.join(_PKG, "liblob.so")     # override: A/B of two builds

_lib = None

_f32p = C.c_void_p
_SIGS = {
    "lob_version": ([], C.c_int),
    "lob_build_id": ([], C.c_char_p),
    "lob_debug_set_variant": ([C.c_int, C.c_int], C.c_int),
    "lob_debug_get_variant": ([C.c_int], C.c_int),
    "lob_lstm_uses_fragment_layout": ([C.c_int], C.c_int),
    "lob_gemm_nt_f32": ([_f32p, C.c_int, _f32p, C.c_int, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                         C.c_int, C.c_void_p], C.c_int),
    "lob_gemm_tn_f32": ([_f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                         C.c_void_p], C.c_int),
    "lob_gemm_nt_f32_split": ([_f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p,
                               C.c_float, C.c_uint64, C.c_void_p], C.c_int),
    "lob_gemm_tn_f32_split": ([_f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p,
                               C.c_void_p], C.c_int),
    "lob_gate_gemm_x_f32": ([_f32p, C.c_int, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.c_int, _f32p, C.c_void_p], C.c_int),
    "lob_lstm_rec_fwd_f32": ([_f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                              _f32p, C.c_void_p], C.c_int),
    "lob_lstm_rec_fwd_f32_drop": ([_f32p, _f32p, _f32p, _f32p, C.c_float, C.c_uint64, _f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   _f32p, C.c_void_p], C.c_int),
    "lob_layernorm_act_f32": ([_f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                               C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_void_p], C.c_int),
    "lob_input_proj_bwd_bf16": ([_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, _f32p, C.c_int, _f32p, _f32p, _f32p, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, C.c_uint64, C.c_void_p], C.c_int),
    "lob_attn_ln_bwd_bf16": ([_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int,
                              C.c_int, C.c_int, C.c_float, C.c_void_p], C.c_int),
    "lob_attn_scores_bf16": ([_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int,
                              C.c_int, C.c_int, C.c_float, C.c_void_p], C.c_int),
    "lob_attn_scores_f32": ([_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int,
                             C.c_int, C.c_int, C.c_float, C.c_void_p], C.c_int),
    "lob_input_proj_ln_f32": ([_f32p, C.c_int, _f32p, C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_float, C.c_int, C.c_float, C.c_uint64, C.c_void_p], C.c_int),
    "lob_input_proj_ln_bf16": ([_f32p, C.c_int, _f32p, C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, _f32p, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, C.c_uint64, C.c_void_p], C.c_int),
    "lob_lstm_rec_bwd_f32": ([_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                              C.c_void_p], C.c_int),
    "lob_lstm_rec_bwd_f32_x": ([_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_void_p], C.c_int),
    "lob_lstm_rec_fwd_bf16": ([_f32p, C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_int, _f32p, _f32p, C.c_float, C.c_uint64, C.c_int,
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "lob_lstm_rec_bwd_bf16": ([_f32p, C.c_int, _f32p, C.c_int, _f32p, _f32p, _f32p, C.c_int, _f32p, _f32p, _f32p, C.c_int, C.c_int,
                               C.c_int, C.c_int, C.c_void_p], C.c_int),
    "lob_colsum_bf16": ([_f32p, C.c_int, C.c_int, C.c_int, _f32p, C.c_void_p], C.c_int),
    "lob_gemm_nt_bf16": ([_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, _f32p, _f32p, C.c_int, C.c_int, C.c_int,
                          C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_void_p], C.c_int),
    "lob_gate_gemm_x_bf16": ([_f32p, C.c_int, C.c_int, _f32p, C.c_int, _f32p, _f32p, C.c_int, C.c_int, C.c_int,
                              C.c_int, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "lob_gemm_tn_bf16": ([_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                          C.c_void_p], C.c_int),
    "lob_lstm_dw_bf16": ([_f32p, C.c_int, _f32p, C.c_int, C.c_int, _f32p, C.c_int, _f32p, _f32p, C.c_int, C.c_int,
                          C.c_int, C.c_int, C.c_void_p], C.c_int),
    "lob_colsum_f32": ([_f32p, C.c_int, C.c_int, C.c_int, _f32p, C.c_void_p], C.c_int),
    "lob_act_f32": ([_f32p, _f32p, C.c_int64, C.c_int, C.c_void_p], C.c_int),
    "lob_act_bwd_f32": ([_f32p, _f32p, _f32p, C.c_int64, C.c_int, C.c_void_p], C.c_int),
    "lob_layernorm_act_bwd_f32": ([_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_float,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint64, _f32p, _f32p,
                                   C.c_int, C.c_int, C.c_int, _f32p, C.c_void_p], C.c_int),
    "lob_attn_pool_bwd_f32": ([_f32p, C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, _f32p, _f32p] +
                              [C.c_int] * 5 + [_f32p, C.c_void_p], C.c_int),
    "lob_dropout_f32": ([_f32p, _f32p, C.c_int64, C.c_float, C.c_uint64, C.c_void_p], C.c_int),
    "lob_attn_pool_fwd_f32": ([_f32p, C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_void_p], C.c_int),
    "lob_softmax_rows_f32": ([_f32p, _f32p, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "lob_ode_rk4_f64": ([_f32p, C.c_void_p, C.POINTER(C.c_double), C.c_double, C.c_int, C.c_double, C.c_double,
                         C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "lob_prob_to_state_f64": ([_f32p, C.c_void_p, C.c_int, C.c_void_p], C.c_int),
    "lob_weighted_ce_f32": ([_f32p, C.c_void_p, _f32p, _f32p, _f32p, C.c_void_p, C.c_int, C.c_int, C.c_float,
                             C.c_void_p], C.c_int),
    "lob_sumsq_f32": ([_f32p, C.c_int64, _f32p, _f32p, C.c_void_p], C.c_int),
    "lob_clip_scale_f32": ([_f32p, C.c_int64, _f32p, C.c_float, C.c_void_p], C.c_int),
    "lob_adamw_f32": ([_f32p, _f32p, _f32p, _f32p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                       C.c_int64, _f32p, C.c_float, C.c_float, C.c_void_p], C.c_int),
    "lob_pad_cast_bf16": ([_f32p, _f32p, C.c_int64, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "lob_abs_colsum_f32": ([_f32p, C.c_int64, C.c_int, C.c_float, _f32p, C.c_void_p], C.c_int),
    "lob_prep_weights": ([C.c_void_p, C.c_int, C.c_void_p], C.c_int),
}


class PrepOp(C.Structure):
    """LobPrepOp of include/lob.h."""
    _fields_ = [("src", C.c_void_p), ("src2", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int),
                ("ld_src", C.c_int), ("ld_dst", C.c_int), ("pad_to", C.c_int), ("kind", C.c_int), ("blk0", C.c_int),
                ("reserved", C.c_int)]


PREP_MAX, PREP_TRANSPOSE, PREP_BF16, PREP_ABSMAX, PREP_LNBOUND = 64, 1, 2, 4, 8


class LobError(RuntimeError):
    pass


def lib():
    """The loaded library; raises if liblob.so has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LobError(
                f"{LIB_PATH} not found: the HIP extension is not built. "
                "Run `python -m lstm_ode_bci_amd.build` (needs hipcc). There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        missing = [name for name in _SIGS if not hasattr(l, name)]
        if missing:
            raise LobError(f"{LIB_PATH} is stale: it does not export {missing}. Rebuild: python -m lstm_ode_bci_amd.build")
        for name, (args, res) in _SIGS.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = res
        _check_build_id(l)
        _lib = l
    return _lib


def _check_build_id(l):
    """A prebuilt library must come from the sources next to it: compare lob_build_id() with the hash of csrc/*.hip,
    *.h, include/*.h and the compile flags (build.source_id).  LOB_LIB_PATH (A/B of two builds) and
    LOB_SKIP_BUILD_ID=1 opt out."""
    if os.environ.get("LOB_LIB_PATH") or os.environ.get("LOB_SKIP_BUILD_ID") == "1":
        return
    from . import build as _b
    have = l.lob_build_id().decode()
    want = _b.source_id()
    if have != want:
        raise LobError(f"{LIB_PATH} was built from other sources (build id {have}, sources hash to {want}). "
                       "Rebuild: python -m lstm_ode_bci_amd.build --force")


# kernel-variant switches (include/lob.h LOB_VAR_*): test-only
VAR = {"REC_BWD_DMA": 0, "NT_DMA": 1, "DMA_TILE": 2, "DMA_KT": 3, "NT_ADEEP": 4, "GATE_WS": 5, "REC_BF16_ROWS": 6,
       "F32_DMA": 7, "REC_FWD_ROWS": 8, "LN_LPR": 9, "NT_WGS": 10, "NT_TK": 11, "NT_STAGGER": 12, "FUSED_DW": 13,
       "F32_SPLIT": 14, "H256_LDSW": 15, "DX_KSPLIT": 16, "REC_FEW": 17, "GEMM_PP": 18, "REC_HALF": 19}


def get_variant(name):
    return lib().lob_debug_get_variant(VAR[name])


class variant:
    """with variant(REC_BWD_DMA=0): ...  -- run a block on a kernel twin (tests only)."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: lib().lob_debug_set_variant(VAR[k], int(v)) for k, v in self.kw.items()}
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            lib().lob_debug_set_variant(VAR[k], v)


_ERR = {-1: "LOB_E_ARG (null pointer / bad size)", -2: "LOB_E_SHAPE (unsupported shape)",
        -3: "LOB_E_ALIGN (pointer / leading dimension alignment)"}


def check(rc, what):
    if rc != 0:
        raise LobError(f"{what} failed: {_ERR.get(rc, f'hipError_t {rc}')}")
