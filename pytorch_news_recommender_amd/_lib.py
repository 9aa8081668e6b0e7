"""ctypes binding of libnrms_hip.so (the C ABI in include/nrms_hip.h).

There is NO fallback: if the library is missing or a call fails this raises.  The
product path never computes on the CPU and never imports ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NRMS_HIP_LIB") or os.path.join(PKG, "libnrms_hip.so")   # override: diagnostic builds only

NRMS_FLAG_PAD_ROW_ZERO = 1
NRMS_FLAG_DEFER_WQKV = 2
NRMS_FLAG_FWD_SCRATCH_KEPT = 4
NRMS_FLAG_FUSED_SEQ64 = 8
NRMS_PRECISION_FP32 = 0
NRMS_PRECISION_BF16X3 = 1
NRMS_PRECISION_BF16 = 2
NRMS_PRECISION_FP16 = 3
NRMS_DROPOUT_FIELDS16 = 0x100
NRMS_FP16_KP, NRMS_FP16_DP, NRMS_FP16_QP = 320, 320, 224     # fixed activation pitches of the fp16 mode (include/nrms_hip.h)
PRECISIONS = {"fp32": 0, "bf16x3": 1, "bf16": 2, "fp16": 3}


class NrmsError(RuntimeError):
    pass


class EncoderDesc(C.Structure):
    _fields_ = [("n_seq", C.c_int32), ("seq_len", C.c_int32), ("d_model", C.c_int32),
                ("n_heads", C.c_int32), ("q_dim", C.c_int32), ("vocab", C.c_int32),
                ("p_drop_embed", C.c_float), ("p_drop_ctx", C.c_float), ("precision", C.c_int32),
                ("use_output_proj", C.c_int32), ("mask_mode", C.c_int32), ("flags", C.c_int32),
                ("seed", C.c_uint64), ("loss_scale", C.c_float), ("p_drop_attn", C.c_float), ("seq_index", C.c_void_p)]


class EncoderWeights(C.Structure):
    _fields_ = [("table", C.c_void_p), ("w_qkv", C.c_void_p), ("b_qkv", C.c_void_p), ("w_o", C.c_void_p),
                ("b_o", C.c_void_p), ("w_add", C.c_void_p), ("b_add", C.c_void_p), ("q_vec", C.c_void_p)]


class EncoderGrads(C.Structure):
    _fields_ = [("table", C.c_void_p), ("w_qkv", C.c_void_p), ("b_qkv", C.c_void_p), ("w_o", C.c_void_p),
                ("b_o", C.c_void_p), ("w_add", C.c_void_p), ("b_add", C.c_void_p), ("q_vec", C.c_void_p)]


class EncoderActs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("qkv", C.c_void_p), ("attn", C.c_void_p), ("ctx", C.c_void_p),
                ("t", C.c_void_p), ("w", C.c_void_p), ("scratch", C.c_void_p)]


class SegPoolDesc(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_seg", C.c_int64), ("nnz", C.c_int64), ("d", C.c_int32), ("q", C.c_int32),
                ("precision", C.c_int32), ("flags", C.c_int32)]


NRMS_SEGPOOL_ROWS_UNIQUE = 1


class NewsFeatures(C.Structure):
    _fields_ = [("n", C.c_int64), ("d_text", C.c_int32), ("d_cat", C.c_int32), ("n_cat", C.c_int32), ("n_sub", C.c_int32),
                ("p_drop", C.c_float), ("seed", C.c_uint64), ("title_vec", C.c_void_p), ("abst_vec", C.c_void_p),
                ("cat_table", C.c_void_p), ("sub_table", C.c_void_p), ("categ", C.c_void_p), ("subcateg", C.c_void_p)]


# name -> (restype, argtypes).  Every symbol include/nrms_hip.h declares.
SIGNATURES = {
    "nrms_encoder_fwd": (C.c_int, [C.POINTER(EncoderDesc), C.POINTER(EncoderWeights), C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.POINTER(EncoderActs), C.c_void_p, C.c_void_p]),
    "nrms_encoder_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(EncoderDesc)]),
    "nrms_encoder_fwd_scratch_bytes": (C.c_size_t, [C.POINTER(EncoderDesc)]),
    "nrms_encoder_fused_qkv_bytes": (C.c_size_t, [C.POINTER(EncoderDesc)]),
    "nrms_encoder_bwd": (C.c_int, [C.POINTER(EncoderDesc), C.POINTER(EncoderWeights), C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.POINTER(EncoderActs), C.c_void_p, C.POINTER(EncoderGrads), C.c_void_p,
                                   C.c_void_p, C.c_size_t, C.c_void_p]),
    "nrms_encoder_bwd_wqkv": (C.c_int, [C.POINTER(EncoderDesc), C.c_void_p, C.c_void_p, C.POINTER(EncoderActs),
                                        C.POINTER(EncoderGrads), C.c_void_p, C.c_size_t, C.c_void_p]),
    "nrms_sanitize_ids": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "nrms_sanitize_ids_i32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "nrms_title_dedup": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "nrms_click_score_indexed": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrms_click_score_fwd": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "nrms_click_score_bwd": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrms_ce_loss_fwd_bwd": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                       C.c_void_p]),
    "nrms_adam_step": (C.c_int, [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                 C.c_double, C.c_double, C.c_double, C.c_int32, C.c_float, C.c_void_p]),
    "nrms_adam_step_guarded": (C.c_int, [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                         C.c_double, C.c_double, C.c_double, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "nrms_grad_guard": (C.c_int, [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrms_segment_pool_workspace_bytes": (C.c_size_t, [C.POINTER(SegPoolDesc)]),
    "nrms_segment_pool_fwd": (C.c_int, [C.POINTER(SegPoolDesc)] + [C.c_void_p] * 10 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "nrms_segment_pool_bwd": (C.c_int, [C.POINTER(SegPoolDesc)] + [C.c_void_p] * 12 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "nrms_hier_tree_scratch_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "nrms_hier_tree_build": (C.c_int, [C.c_int32, C.c_int32] + [C.c_void_p] * 16 + [C.c_size_t, C.c_void_p]),
    "nrms_hier_add_embedding_fwd": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrms_sequence_partition_count_ints": (C.c_size_t, [C.c_int32]),
    "nrms_sequence_partition": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrms_encoder_empty_workspace_bytes": (C.c_size_t, [C.POINTER(EncoderDesc)]),
    "nrms_encoder_empty_saved_bytes": (C.c_size_t, [C.POINTER(EncoderDesc)]),
    "nrms_encoder_empty_fwd": (C.c_int, [C.POINTER(EncoderDesc), C.POINTER(EncoderWeights), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.c_void_p]),
    "nrms_encoder_empty_bwd": (C.c_int, [C.POINTER(EncoderDesc), C.POINTER(EncoderWeights), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(EncoderGrads),
                                         C.c_void_p, C.c_size_t, C.c_void_p]),
    "nrms_csr_from_padded": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nrms_hier_add_embedding_bwd_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "nrms_hier_add_embedding_bwd": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_size_t, C.c_void_p]),
    "nrms_hier_match": (C.c_int, [C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 11 + [C.c_void_p]),
    "nrms_hier_score_fwd": (C.c_int, [C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 9 + [C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "nrms_hier_score_bwd": (C.c_int, [C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 9 + [C.c_float, C.c_float] + [C.c_void_p] * 5 + [C.c_void_p]),
    "nrms_impression_auc": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "nrms_dropout_keep_mask": (C.c_int, [C.c_uint64, C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_void_p,
                                         C.c_void_p]),
    "nrms_layernorm_fwd": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "nrms_layernorm_bwd_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "nrms_layernorm_bwd": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "nrms_news_features_fwd": (C.c_int, [C.POINTER(NewsFeatures), C.c_void_p, C.c_void_p]),
    "nrms_news_features_bwd": (C.c_int, [C.POINTER(NewsFeatures), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "nrms_timing_enable": (None, [C.c_int]),
    "nrms_timing_reset": (None, []),
    "nrms_timing_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "nrms_last_error": (C.c_char_p, []),
    "nrms_version": (C.c_char_p, []),
}

_lib = None


def load():
    """Load the library (once) and bind every declared symbol.  Raises NrmsError if the
    shared object is absent -- build it with ``python -m pytorch_news_recommender_amd.build``."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NrmsError(
            "libnrms_hip.so not found at %s: the HIP extension is required (no CPU fallback). "
            "Build it with `python -m pytorch_news_recommender_amd.build`." % LIB_PATH)
    # torch ships its own libamdhip64; it must be resident first so this library binds to the
    # SAME HIP runtime instance (streams / device pointers are shared with torch).  Loading
    # libnrms_hip.so before torch would pull in /opt/rocm's copy and split the process in two.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the header and the .so disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().nrms_last_error()
        raise NrmsError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())
