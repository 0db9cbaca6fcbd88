"""ctypes binding of libctrhip.so (the C ABI declared in include/ctrhip.h).

There is no CPU fallback: if the shared library is missing or a tensor is not on
a HIP device the call raises.  ``import torch`` must happen before the library
is loaded so that it binds to the HIP runtime PyTorch already mapped (same
SONAME ``libamdhip64.so.7``) and shares its streams.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch  # noqa: F401  (maps libamdhip64 first, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CTRHIP_LIB", os.path.join(_HERE, "libctrhip.so"))  # env override: A/B builds
ABI_VERSION = 33
DIN_TRIPLE, DIN_PAIR, DIN_H = 0, 1, 2  # layouts of the DIN attention operand (include/ctrhip.h)

CTR_MAX_FIELDS = 32
CTR_NCF_PROJ_MAX_ROWS = 16384
CTR_NCF_PROJ_COUNT_STRIDE = 16
CTR_ROWS1_MAX_ROWS = 32768
FIELD_ID_I64, FIELD_ID_F32, FIELD_BAG, FIELD_DENSE, FIELD_PROD_I64 = range(5)
ACT_NONE, ACT_RELU, ACT_SIGMOID = range(3)


class Field(C.Structure):
    """mirror of ``ctr_field_t``"""
    _fields_ = [
        ("kind", C.c_int32), ("width", C.c_int32), ("out_col", C.c_int32), ("src_col", C.c_int32),
        ("bag_size", C.c_int32), ("reserved", C.c_int32),
        ("vocab", C.c_int64), ("idx_stride", C.c_int64),
        ("idx", C.c_void_p), ("table", C.c_void_p), ("grad", C.c_void_p),
        ("idx2", C.c_void_p), ("table2", C.c_void_p), ("grad2", C.c_void_p),
        ("vocab2", C.c_int64),
    ]


class MlpHead(C.Structure):
    """mirror of ``ctr_mlp_head_t``"""
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int64), ("w", C.c_void_p), ("c", C.c_void_p), ("out", C.c_void_p),
                ("ldout", C.c_int64), ("p", C.c_int32), ("act", C.c_int32)]


class HeadFold(C.Structure):
    """mirror of ``ctr_head_fold_t``"""
    _fields_ = [("u_full", C.c_void_p), ("w", C.c_void_p), ("ldw", C.c_int64), ("b", C.c_void_p), ("b2", C.c_void_p),
                ("wfold", C.c_void_p), ("cfold", C.c_void_p), ("p", C.c_int32), ("n", C.c_int32), ("k", C.c_int32),
                ("reserved", C.c_int32)]


class HeadFoldGrad(C.Structure):
    """mirror of ``ctr_head_fold_grad_t``"""
    _fields_ = [("u_full", C.c_void_p), ("w", C.c_void_p), ("ldw", C.c_int64), ("b", C.c_void_p), ("gu_full", C.c_void_p),
                ("gw", C.c_void_p), ("ldgw", C.c_int64), ("gb", C.c_void_p), ("gb2", C.c_void_p), ("p", C.c_int32),
                ("n", C.c_int32), ("k", C.c_int32), ("reserved", C.c_int32)]


class MlpHeadGrad(C.Structure):
    """mirror of ``ctr_mlp_head_grad_t``"""
    _fields_ = [("prob", C.c_void_p), ("ldprob", C.c_int64), ("gprob", C.c_void_p), ("ldgprob", C.c_int64),
                ("x", C.c_void_p), ("ldx", C.c_int64), ("w", C.c_void_p), ("gx", C.c_void_p), ("ldgx", C.c_int64),
                ("gw", C.c_void_p), ("gc", C.c_void_p), ("p", C.c_int32), ("act", C.c_int32)]


class MlpLayer(C.Structure):
    """mirror of ``ctr_mlp_layer_t``"""
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("y", C.c_void_p), ("ldy", C.c_int64),
                ("gw", C.c_void_p), ("gb", C.c_void_p), ("n", C.c_int32), ("k", C.c_int32), ("act", C.c_int32),
                ("reserved", C.c_int32)]


class NcfProj(C.Structure):
    """mirror of ``ctr_ncf_proj_t``"""
    _fields_ = [("user_idx", C.c_void_p), ("user_stride", C.c_int64), ("item_idx", C.c_void_p), ("item_stride", C.c_int64),
                ("batch", C.c_int64), ("num_users", C.c_int64), ("num_items", C.c_int64),
                ("mlp_user", C.c_void_p), ("mlp_item", C.c_void_p), ("gmf_user", C.c_void_p), ("gmf_item", C.c_void_p),
                ("mlp_dim", C.c_int32), ("mf_dim", C.c_int32), ("layers", MlpLayer * 4),
                ("proj_w", C.c_void_p), ("ld_proj_w", C.c_int64), ("proj_b", C.c_void_p), ("proj_n", C.c_int32),
                ("proj_k", C.c_int32), ("head_w", C.c_void_p), ("head_b", C.c_void_p), ("head_act", C.c_int32),
                ("prob", C.c_void_p), ("ldprob", C.c_int64), ("err_flag", C.c_void_p),
                ("ptab", C.c_void_p), ("wfold", C.c_void_p), ("counts", C.c_void_p), ("ranks", C.c_void_p),
                ("training", C.c_int32), ("phases", C.c_int32)]


class NcfProjGrad(C.Structure):
    """mirror of ``ctr_ncf_proj_grad_t``"""
    _fields_ = [("gprob", C.c_void_p), ("ldgprob", C.c_int64), ("layers", MlpLayer * 4),
                ("g_mlp_user", C.c_void_p), ("g_mlp_item", C.c_void_p), ("g_gmf_user", C.c_void_p), ("g_gmf_item", C.c_void_p),
                ("g_proj_w", C.c_void_p), ("ld_g_proj_w", C.c_int64), ("g_proj_b", C.c_void_p), ("g_head_w", C.c_void_p),
                ("g_head_b", C.c_void_p), ("workspace", C.c_void_p), ("workspace_floats", C.c_int64),
                ("zero_buf", C.c_void_p), ("zero_floats", C.c_int64), ("phases", C.c_int32), ("reserved", C.c_int32)]


class AdamTensor(C.Structure):
    """mirror of ``ctr_adam_tensor_t``"""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("numel", C.c_int64)]


class RowsMark(C.Structure):
    """mirror of ``ctr_rows_mark_t``"""
    _fields_ = [("ids", C.c_void_p), ("stride", C.c_int64), ("n", C.c_int64), ("vocab", C.c_int64),
                ("flags", C.c_void_p), ("rows", C.c_void_p), ("count", C.c_void_p), ("ids_are_float", C.c_int32),
                ("reserved", C.c_int32)]


class RowsTable(C.Structure):
    """mirror of ``ctr_rows_table_t``"""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("flags", C.c_void_p), ("rows", C.c_void_p), ("count", C.c_void_p), ("done", C.c_void_p),
                ("dim", C.c_int32), ("reserved", C.c_int32)]


_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); must list every `ctr_*` symbol of include/ctrhip.h
SIGNATURES = {
    "ctr_version": (C.c_int, []),
    "ctr_strerror": (C.c_char_p, [_i]),
    "ctr_target_arch": (C.c_char_p, []),
    "ctr_embed_fwd": (_i, [C.POINTER(Field), _i, _p, _l, _l, _p, _l, _p, _p]),
    "ctr_embed_bwd": (_i, [C.POINTER(Field), _i, _p, _l, _l, _p, _l, _p, _l, _p]),
    "ctr_mf_fwd": (_i, [_p, _l, _p, _l, _i, _p, _p, _l, _p, _p, _p]),
    "ctr_mf_bwd": (_i, [_p, _l, _p, _l, _i, _p, _p, _l, _p, _p, _p, _p, _p]),
    "ctr_linear_fwd": (_i, [_p, _l, _p, _l, _p, _p, _l, _p, _l, _l, _i, _i, _i, _p]),
    "ctr_linear_bwd": (_i, [_p, _l, _p, _l, _p, _l, _p, _l, _p, _l, _i, _p, _l, _p, _l, _i, _i, _i, _p, _l, _p]),
    "ctr_allpairs_fwd": (_i, [_p, _l, _l, _i, _i, _p, _l, _p]),
    "ctr_allpairs_bwd": (_i, [_p, _l, _l, _i, _i, _p, _l, _p, _l, _i, _p]),
    "ctr_fields_pairs_fwd": (_i, [_p, _l, _l, _i, _i, _p, _l, _p]),
    "ctr_fields_pairs_bwd": (_i, [_p, _l, _l, _i, _i, _p, _l, _p, _l, _i, _p]),
    "ctr_fm_wide_fwd": (_i, [_p, _l, _l, _i, _i, _p, _l, _i, _i, _i, _i, _p, _l, _p, _l, _p, _p, _p, _l, _p, _p]),
    "ctr_fm_wide_bwd": (_i, [_p, _l, _l, _i, _i, _p, _l, _i, _i, _i, _i, _p, _l, _p, _l, _p, _p, _p, _l,
                             _p, _p, _p, _p, _p, _l, _i, _p, _l, _p]),
    "ctr_fields_fm_fwd": (_i, [_p, _l, _l, _i, _i, C.POINTER(_p), C.POINTER(_l), C.POINTER(_p), _p, _p, _l, _p, _l, _p, _p]),
    "ctr_fields_fm_bwd": (_i, [_p, _l, _l, _i, _i, C.POINTER(_l), _p, _l, _p, _l, _p, _l, C.POINTER(_p), C.POINTER(_p),
                               _p, _p, _l, _p]),
    "ctr_ffm_fused_fwd": (_i, [_p, _l, _l, _i, C.POINTER(_p), _l, _l, _p, _p, _p, _p, _p, _l, _p, _l, _p, _p]),
    "ctr_ffm_fused_bwd": (_i, [_p, _l, _l, _i, _p, _l, _l, _l, _p, _p, _l, _p, _l, _p, _p, _p, _p, _p, _l, _p, _l, _p]),
    "ctr_ffm_head_fwd": (_i, [_p, _l, _l, _i, _i, C.POINTER(C.c_int32), _i, _p, _l, _i, _i, _i, _i, _p, _l, _p, _l,
                              _p, _p, _p, _l, _p, _p]),
    "ctr_ffm_head_bwd": (_i, [_p, _l, _l, _i, _i, C.POINTER(C.c_int32), _i, _p, _l, _i, _i, _i, _i, _p, _l, _p, _l,
                              _p, _p, _p, _l, _p, _l, _p, _p, _p, _p, _p, _l, _p, _l, _p]),
    "ctr_act_bwd": (_i, [_p, _l, _p, _l, _p, _l, _l, _i, _i, _i, _p]),
    "ctr_biinteract_fwd": (_i, [_p, _l, _l, _i, _i, _p, _l, _p]),
    "ctr_biinteract_bwd": (_i, [_p, _l, _l, _i, _i, _p, _l, _p, _l, _i, _p]),
    "ctr_pairprod_fwd": (_i, [_p, _l, _l, _i, _i, _p, _l, _p]),
    "ctr_pairprod_bwd": (_i, [_p, _l, _l, _i, _i, _p, _l, _p, _p, _l, _p, _l, _i, _p]),
    "ctr_cross_fwd": (_i, [_p, _l, _p, _l, _p, _l, _p, _p, _l, _l, _i, _p]),
    "ctr_cross_bwd": (_i, [_p, _l, _p, _l, _p, _l, _p, _l, _p, _l, _p, _l, _i, _p, _l, _p]),
    "ctr_din_concat_fwd": (_i, [_p, _l, _i, _p, _p, _l, _i, _p, _l, _p, _l, _i, _p, _p]),
    "ctr_din_pool_fwd": (_i, [_p, _p, _l, _l, _i, _i, _p, _p, _l, _i, _p]),
    "ctr_din_pool_bwd": (_i, [_p, _p, _l, _l, _i, _i, _p, _l, _i, _p, _p]),
    "ctr_din_concat_bwd": (_i, [_p, _p, _l, _l, _i, _i, _p, _l, _p, _p, _l, _i, _p, _l, _i, _p, _p]),
    "ctr_gru_fwd": (_i, [_p, _l, _p, _p, _l, _i, _i, _p, _p, _l, _p]),
    "ctr_gru_bwd": (_i, [_p, _l, _p, _p, _p, _l, _i, _i, _p, _l, _p, _p, _p]),
    "ctr_mlp_fwd": (_i, [_p, _l, _l, C.POINTER(MlpLayer), _i, _p]),
    "ctr_mlp_head_fwd": (_i, [_p, _l, _l, C.POINTER(MlpLayer), _i, C.POINTER(MlpHead), _p]),
    "ctr_embed_mlp_head_fwd": (_i, [C.POINTER(Field), _i, _l, _p, _l, _p, _i, C.POINTER(MlpLayer), _i, C.POINTER(MlpHead),
                                    C.POINTER(HeadFold), _p]),
    "ctr_embed_mlp_head_bwd": (_i, [C.POINTER(Field), _i, _l, C.POINTER(MlpLayer), _i, C.POINTER(MlpHeadGrad),
                                    C.POINTER(HeadFoldGrad), _p, _l, _p, _l, _p, _l, _p]),
    "ctr_mlp_head_bwd": (_i, [_p, _l, _l, C.POINTER(MlpLayer), _i, C.POINTER(MlpHeadGrad), _p, _l, _p, _l, _p]),
    "ctr_rows_sum_act_fwd": (_i, [_p, _p, _l, _l, _p, _p, _l, _l, _l, _i, _i, _p, _l, _p, _p]),
    "ctr_act_mask_bwd": (_i, [_p, _l, _p, _l, _l, _i, _i, _p]),
    "ctr_ncf_proj_workspace_floats": (_i, [_l, _l, _l, C.POINTER(C.c_int64)]),
    "ctr_ncf_proj_fwd": (_i, [C.POINTER(NcfProj), _p]),
    "ctr_ncf_proj_bwd": (_i, [C.POINTER(NcfProj), C.POINTER(NcfProjGrad), _p]),
    "ctr_mlp_bwd": (_i, [_p, _l, _l, C.POINTER(MlpLayer), _i, _p, _l, _p, _l, _p, _l, _p]),
    "ctr_negative_sample": (_i, [_p, _l, _l, _l, _i, C.c_uint64, _p, _p, _p, _p]),
    "ctr_assemble_features": (_i, [_p, _p, _l, _p, _i, _l, _p, _i, _l, _p, _l, _p, _p]),
    "ctr_shard_bucket": (_i, [_p, _l, _i, _l, _p, _p, _p, _p, _p, _p]),
    "ctr_shard_bucket_padded": (_i, [_p, _l, _i, _l, _l, _p, _p, _p, _p, _p, _p]),
    "ctr_shard_recv_rows": (_i, [_p, _l, _l, _p, _p, _p, _p]),
    "ctr_rows_zero": (_i, [_p, _l, _l, _i, _p, _l, _p]),
    "ctr_topk_rows": (_i, [_p, _l, _l, _l, _l, _i, _p, _p, _p]),
    "ctr_rows1_scatter": (_i, [_p, _l, _i, _i, _p, _l, _p, _l, _l, _p, _l, _p, _l, _p]),
    "ctr_fold_head_fwd": (_i, [_p, _i, _p, _l, _p, _p, _i, _i, _p, _p, _p]),
    "ctr_fold_head_bwd": (_i, [_p, _i, _p, _l, _p, _i, _i, _p, _p, _p, _p, _l, _p, _p, _p]),
    "ctr_bce_fwd": (_i, [_p, _l, _p, _l, _l, _p, _p, _l, _p, _p, _p]),
    "ctr_bce_bwd": (_i, [_p, _l, _p, _l, _l, _p, _p, _l, _p]),
    "ctr_din_scatter_bwd": (_i, [_p, _l, _l, _i, _i, _p, _l, _p, _p, _l, _i, _p, _p]),
    "ctr_linear_group_fwd": (_i, [_p, _l, _p, _l, _p, _p, _l, _i, _p, _l, _p, _l, _l, _i, _i, _i, _p]),
    "ctr_linear_dx_masked": (_i, [_p, _l, _p, _l, _p, _l, _i, _p, _l, _i, _p, _l, _p, _l, _p, _l, _i, _l, _i, _i, _p]),
    "ctr_gru_fused_fwd": (_i, [_p, _l, _p, _p, _p, _p, _l, _i, _i, _p, _p, _l, _p]),
    "ctr_gru_fused_bwd": (_i, [_p, _l, _p, _p, _p, _p, _p, _l, _i, _i, _p, _l, _p, _l, _p, _p, _p, _p, _p, _l, _p]),
    "ctr_linear_fwd_dot": (_i, [_p, _l, _p, _l, _p, _p, _l, _p, _p, _p, _l, _l, _i, _i, _i, _p]),
    "ctr_linear_dx_scatter": (_i, [_p, _l, _p, _l, _p, _p, _p, _l, _i, _p, _l, _l, _i, _i, _p]),
    "ctr_linear_n1_bwd_masked": (_i, [_p, _l, _p, _p, _l, _i, _p, _l, _p, _p, _l, _i, _p, _l, _p]),
    "ctr_rows_mark": (_i, [C.POINTER(RowsMark), _i, _p]),
    "ctr_adam_rows": (_i, [C.POINTER(RowsTable), _i, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _l, _p]),
    "ctr_rows_discard": (_i, [C.POINTER(RowsTable), _i, _p]),
    "ctr_adam_step": (_i, [C.POINTER(AdamTensor), _i, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _l, _p]),
}

_lock = threading.Lock()
_lib = None


class CtrHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libctrhip.so once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise CtrHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C deeplearningrecommendationsystem_amd/csrc`). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        got = lib.ctr_version()
        if got != ABI_VERSION:
            raise CtrHipError(f"libctrhip ABI version {got}, python binding expects {ABI_VERSION}: rebuild")
        _lib = lib
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        raise CtrHipError(f"{what} failed: {load().ctr_strerror(code).decode()} ({code})")


def stream_ptr() -> int:
    """the HIP stream kernels must be enqueued on: torch's current stream"""
    return torch.cuda.current_stream().cuda_stream


def require_device(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise CtrHipError("libctrhip kernels need tensors on a HIP device (got a CPU tensor); "
                              "this backend has no CPU fallback")


def ptr(t):
    return None if t is None else t.data_ptr()
