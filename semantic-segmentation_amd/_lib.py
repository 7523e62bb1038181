"""ctypes binding of libssmi.so (the C ABI declared in include/ssmi.h).

The library is the only compute path: importing this module without a built libssmi.so raises, and there is no
CPU fallback.  Device memory, streams and process groups come from PyTorch-ROCm; nothing torch-typed crosses the ABI
(only raw device pointers, sizes and the hipStream_t handle).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libssmi.so")


class SsmiError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("n", "h", "w", "cin", "ho", "wo", "cout", "kh", "kw", "stride", "dil", "pad_t", "pad_l", "x_ld", "y_ld",
                 "dtype")] + [("w_fwd_pack", C.c_void_p), ("w_dgrad_pack", C.c_void_p)]


class MergeDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n", "h", "w", "c", "other_mode", "c_other", "dtype")]


class ExpandDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n", "h", "w", "cin", "cout", "dtype")]


class BnParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("mean", "invstd", "gamma", "beta")]


class ExpandNextApply(C.Structure):
    """ssmi_expand_next_apply"""
    _fields_ = [("y1", C.c_void_p), ("bn1", C.POINTER(BnParams)), ("alpha", C.c_void_p), ("coef", C.c_void_p)]


class Premerge(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dsum", "main", "mean", "invstd", "coef", "drop")] + [("hw", C.c_int64)]


_vp, _i32, _i64, _f32, _u64, _u32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64, C.c_uint32, C.c_size_t
_PD = C.POINTER(ConvDesc)
_PP = C.POINTER(Premerge)
_PM = C.POINTER(MergeDesc)
_PB = C.POINTER(BnParams)
_PE = C.POINTER(ExpandDesc)
_PNA = C.POINTER(ExpandNextApply)

# name -> (restype, argtypes); status-returning functions are wrapped to raise SsmiError
_PROTOS = {
    "ssmi_last_error": (C.c_char_p, []),
    "ssmi_last_kernel": (C.c_char_p, []),
    "ssmi_profile_begin": (_i32, []),
    "ssmi_profile_end": (_i32, [C.POINTER(_i32)]),
    "ssmi_profile_get": (_i32, [_i32, C.c_char_p, _i32, C.POINTER(_f32), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ssmi_version": (_i32, []),
    "ssmi_same_pad": (_i32, [_i32, _i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "ssmi_conv2d_fwd": (_i32, [_PD, _vp, _vp, _vp, _vp, _vp]),
    "ssmi_conv2d_fwd_bn_act_infer": (_i32, [_PD, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _f32, _vp, _vp]),
    "ssmi_conv2d_fwd_bn_add_act_infer": (_i32, [_PD, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _f32, _vp,
                                                _vp]),
    "ssmi_conv2d_fwd_bn_stats_workspace_bytes": (_sz, [_PD]),
    "ssmi_conv2d_fwd_bn_stats": (_i32, [_PD, _vp, _vp, _vp, _vp, _f32, _f32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssmi_conv2d_prebn_supported": (_i32, [_PD, _i32]),
    "ssmi_conv2d_fwd_bn_stats_prebn": (_i32, [_PD, _vp, _PB, _vp, _vp, _vp, _vp, _f32, _f32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssmi_conv2d_bwd_weight_prebn": (_i32, [_PD, _vp, _PB, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssmi_conv2d_bwd_data": (_i32, [_PD, _vp, _vp, _vp, _vp, _vp]),
    "ssmi_conv2d_bwd_data_workspace_bytes": (_sz, [_PD]),
    "ssmi_conv2d_bwd_data_ws": (_i32, [_PD, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssmi_conv2d_bwd_data_bn_bwd_workspace_bytes": (_sz, [_PD]),
    "ssmi_conv2d_bwd_data_bn_bwd": (_i32, [_PD, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _f32, _vp, _i64,
                                           _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssmi_wg_reduce_defer_begin": (_i32, []),
    "ssmi_wg_reduce_defer_flush": (_i32, [_i32, _vp]),
    "ssmi_conv2d_bwd_data_bn_bwd_deferred_supported": (_i32, [_PD, _i32]),
    "ssmi_conv2d_bwd_data_bn_bwd_rot_deferred": (_i32, [_PD, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _f32, _vp, _vp, _vp,
                                                        _vp, _vp, _sz, _vp]),
    "ssmi_conv2d_bwd_data_bn_bwd_rot": (_i32, [_PD, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _f32, _vp, _i64,
                                               _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssmi_conv2d_bwd_data_rot": (_i32, [_PD, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssmi_conv2d_bwd_data_uses_rotated": (_i32, [_PD]),
    "ssmi_comm_unique_id": (_i32, [_vp]),
    "ssmi_comm_init": (_i32, [_vp, _i32, _i32, _vp]),
    "ssmi_comm_destroy": (_i32, [_vp]),
    "ssmi_comm_size": (_i32, [_vp]),
    "ssmi_comm_rank": (_i32, [_vp]),
    "ssmi_comm_allreduce_sum_f32": (_i32, [_vp, _vp, _i64, _vp]),
    "ssmi_comm_allreduce_sum_i64": (_i32, [_vp, _vp, _i64, _vp]),
    "ssmi_comm_broadcast_f32": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "ssmi_conv_rotate_weights": (_i32, [_vp, _i32, _vp, _vp, _vp]),
    "ssmi_conv_pack_weights_bf16": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp]),
    "ssmi_conv2d_pack_usage": (_i32, [_PD]),
    "ssmi_conv2d_bwd_weight_workspace_bytes": (_sz, [_PD]),
    "ssmi_conv2d_bwd_weight": (_i32, [_PD, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssmi_conv2d_transpose_fwd": (_i32, [_PD, _vp, _vp, _vp, _vp, _vp]),
    "ssmi_conv2d_transpose_bwd_data": (_i32, [_PD, _vp, _vp, _vp, _vp]),
    "ssmi_conv2d_transpose_bwd_weight_workspace_bytes": (_sz, [_PD]),
    "ssmi_conv2d_transpose_bwd_weight": (_i32, [_PD, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssmi_bn_workspace_bytes": (_sz, [_i64, _i32]),
    "ssmi_bn_train_stats": (_i32, [_vp, _i64, _i32, _i32, _f32, _f32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssmi_bn_infer_stats": (_i32, [_vp, _vp, _i32, _f32, _vp, _vp, _vp]),
    "ssmi_bn_infer_stats_arena": (_i32, [_vp, _vp, _i64, _f32, _vp, _vp]),
    "ssmi_bn_act_fwd": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _f32, _vp, _i64, _i32, _vp]),
    "ssmi_bn_act_bwd": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _f32, _vp, _vp, _vp, _vp, _i64, _i32, _vp,
                               _sz, _vp]),
    "ssmi_bn_act_drop_fwd": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _f32, _vp, _i64, _vp, _i32, _i64, _i32, _vp]),
    "ssmi_bn_act_drop_bwd": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _f32, _vp, _i64, _vp, _vp, _vp, _vp,
                                    _i64, _i32, _vp, _sz, _vp]),
    "ssmi_merge_fwd": (_i32, [_PM, _vp, _PB, _vp, _vp, _PB, _vp, _vp, _vp, _vp]),
    "ssmi_merge_workspace_bytes": (_sz, [_PM]),
    "ssmi_merge_bwd_deferred": (_i32, [_PM, _vp, _vp, _PB, _vp, _vp, _PB, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssmi_merge_bwd_apply": (_i32, [_PM, _vp, _vp, _PB, _vp, _vp, _vp, _vp]),
    "ssmi_conv2d_bwd_data_premerge_supported": (_i32, [_PD]),
    "ssmi_conv2d_bwd_data_bn_bwd_premerge": (_i32, [_PD, _PP, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _f32, _vp, _vp,
                                             _vp, _vp, _vp, _sz, _vp]),
    "ssmi_expand_supported": (_i32, [_PE]),
    "ssmi_expand_workspace_bytes": (_sz, [_PE]),
    "ssmi_expand_bn_stats": (_i32, [_PE, _vp, _PB, _vp, _vp, _f32, _f32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssmi_expand_merge_fwd": (_i32, [_PE, _vp, _PB, _vp, _vp, _PB, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "ssmi_expand_merge_fwd_through": (_i32, [_PE, _vp, _PB, _vp, _vp, _PB, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _i32, _vp, _vp,
                                             _vp, _vp, _vp, _sz, _vp]),
    "ssmi_expand_merge_fwd_through_infer": (_i32, [_PE, _vp, _PB, _vp, _vp, _PB, _vp, _vp, _i32, _vp, _vp, _vp, _PB, _vp, _vp, _vp]),
    "ssmi_expand_merge_bwd": (_i32, [_PE, _vp, _vp, _PB, _vp, _vp, _PB, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssmi_expand_merge_bwd_chained": (_i32, [_PE, _vp, _vp, _vp, _vp, _PB, _vp, _vp, _PB, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                             _sz, _vp]),
    "ssmi_expand_merge_bwd_through": (_i32, [_PE, _vp, _vp, _vp, _vp, _vp, _PB, _vp, _vp, _PB, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                             _vp, _vp, _vp, _vp, _vp, _vp, _PNA, _vp, _sz, _vp]),
    "ssmi_expand_weight_partial_bytes": (_sz, [_PE]),
    "ssmi_expand_weight_reduce": (_i32, [_PE, _vp, _vp, _vp]),
    "ssmi_merge_bwd": (_i32, [_PM, _vp, _vp, _PB, _vp, _vp, _PB, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ssmi_maxpool_argmax_fwd": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp]),
    "ssmi_maxpool_argmax_bwd": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "ssmi_maxunpool_fwd": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ssmi_maxunpool_local_fwd": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ssmi_maxunpool_bwd": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ssmi_idx_f32_to_i32": (_i32, [_vp, _vp, _i64, _vp]),
    "ssmi_softmax_xent_workspace_bytes": (_sz, [_i64]),
    "ssmi_logits_tconv2x2_xent_supported": (_i32, [_i32, _i32]),
    "ssmi_logits_tconv2x2_xent_fwd_bwd": (_i32, [_i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _i64, _f32, _vp,
                                          _vp, _vp, _vp, _i64, _u64, _vp, _vp, _vp, _sz, _vp]),
    "ssmi_softmax_xent_fwd_bwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _f32, _vp, _vp, _vp, _i64, _u64, _vp, _vp, _vp, _sz,
                                         _vp]),
    "ssmi_softmax_xent_static_max_pixels": (_i64, []),
    "ssmi_softmax_xent_fwd_bwd_static": (_i32, [_vp, _vp, _vp, _i64, _i32, _f32, _vp, _vp, _vp, _i64, _u64, _vp, _vp, _vp]),
    "ssmi_mt_consistency_workspace_bytes": (_sz, [_i64]),
    "ssmi_mt_consistency_fwd_bwd": (_i32, [_vp, _vp, _i64, _i32, _vp, _f32, _i64, _vp, _vp, _i32, _vp, _vp, _sz, _vp]),
    "ssmi_superpixel_workspace_bytes": (_sz, [_i64, _i32]),
    "ssmi_superpixel_cost_fwd_bwd": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _f32, _f32, _vp, _i32, _vp, _i32, _vp, _vp,
                                            _sz, _vp]),
    "ssmi_classification_xent_fwd_bwd": (_i32, [_vp, _vp, _vp, _i32, _i32, _f32, _f32, _vp, _vp, _vp]),
    "ssmi_global_avgpool_fwd": (_i32, [_vp, _i32, _i64, _i32, _vp, _vp]),
    "ssmi_global_avgpool_bwd": (_i32, [_vp, _i32, _i64, _i32, _vp, _vp]),
    "ssmi_ema_update": (_i32, [_vp, _vp, C.c_double, _i64, _vp]),
    "ssmi_metrics_from_cm": (_i32, [_vp, _i32, _u64, C.POINTER(C.c_double)]),
    "ssmi_adam_step": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _f32, _f32, _f32, _f32, _f32, _f32, _vp, _vp]),
    "ssmi_sgd_step": (_i32, [_vp, _vp, _vp, _i64, _vp, _f32, _f32, _f32, _i32, _f32, _vp, _vp]),
    "ssmi_adam_step_hp": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "ssmi_sgd_step_hp": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "ssmi_dropout_mask": (_i32, [_vp, _i64, _f32, _u64, _vp, _u32, _vp]),
    "ssmi_fill_f32": (_i32, [_vp, _f32, _i64, _vp]),
    "ssmi_upsample_nn2x_fwd": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "ssmi_upsample_nn2x_bwd": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ssmi_copy_ld": (_i32, [_vp, _i32, _vp, _i32, _i64, _i32, _i32, _vp]),
    "ssmi_scale_nc": (_i32, [_vp, _vp, _vp, _i32, _i64, _i32, _vp]),
    "ssmi_preprocess_batch": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ssmi_predict_prepare": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ssmi_predict_accumulate": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp]),
    "ssmi_predict_topk": (_i32, [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ssmi_profile_bracket_overhead_ms": (_f32, []),
    "ssmi_device_timestamp": (_i32, [_vp, _vp]),
    "ssmi_device_timestamp_khz": (_i32, []),
    "ssmi_graph_begin_capture": (_i32, [_vp]),
    "ssmi_graph_end_capture": (_i32, [_vp, C.POINTER(_vp)]),
    "ssmi_graph_launch": (_i32, [_vp, _vp]),
    "ssmi_graph_destroy": (_i32, [_vp]),
}

# bfloat16 twins: same prototypes as the float32 entry points (pointers are void* either way)
for _n in ("bn_train_stats", "bn_act_fwd", "bn_act_bwd", "bn_act_drop_fwd", "bn_act_drop_bwd", "maxpool_argmax_fwd",
           "maxpool_argmax_bwd", "upsample_nn2x_fwd", "upsample_nn2x_bwd", "copy_ld", "scale_nc", "softmax_xent_fwd_bwd"):
    _PROTOS["ssmi_" + _n + "_bf16"] = _PROTOS["ssmi_" + _n]
_PROTOS["ssmi_cast_f32_to_bf16"] = (_i32, [_vp, _vp, _i64, _vp])
_PROTOS["ssmi_cast_bf16_to_f32"] = (_i32, [_vp, _vp, _i64, _vp])

F32, BF16 = 0, 1      # ssmi dtype codes of activation tensors

EXPORTED_SYMBOLS = tuple(_PROTOS.keys())
_NOT_STATUS = {"ssmi_softmax_xent_static_max_pixels", "ssmi_conv2d_bwd_data_bn_bwd_deferred_supported", "ssmi_expand_supported", "ssmi_expand_workspace_bytes", "ssmi_expand_weight_partial_bytes", "ssmi_conv2d_bwd_data_premerge_supported", "ssmi_conv2d_prebn_supported", "ssmi_logits_tconv2x2_xent_supported", "ssmi_profile_bracket_overhead_ms", "ssmi_device_timestamp_khz", "ssmi_last_error", "ssmi_last_kernel", "ssmi_conv2d_fwd_bn_stats_workspace_bytes", "ssmi_conv2d_bwd_data_bn_bwd_workspace_bytes", "ssmi_conv2d_bwd_data_workspace_bytes", "ssmi_conv2d_bwd_data_uses_rotated", "ssmi_conv2d_pack_usage", "ssmi_comm_size", "ssmi_comm_rank", "ssmi_version", "ssmi_conv2d_bwd_weight_workspace_bytes",
               "ssmi_conv2d_transpose_bwd_weight_workspace_bytes", "ssmi_bn_workspace_bytes", "ssmi_merge_workspace_bytes",
               "ssmi_softmax_xent_workspace_bytes", "ssmi_mt_consistency_workspace_bytes", "ssmi_superpixel_workspace_bytes"}


class _Lib:
    def __init__(self, path: str):
        if not os.path.exists(path):
            raise SsmiError(
                "libssmi.so not found at {} - build it with `make -C semantic-segmentation_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback".format(path))
        # torch must be imported first so that libamdhip64 resolves to the copy PyTorch-ROCm already loaded
        import torch  # noqa: F401
        self._dll = C.CDLL(path, mode=C.RTLD_GLOBAL)
        self.raw = {}
        for name, (res, args) in _PROTOS.items():
            fn = getattr(self._dll, name)
            fn.restype = res
            fn.argtypes = args
            self.raw[name] = fn
            short = name[len("ssmi_"):]
            setattr(self, short, fn if name in _NOT_STATUS else self._checked(fn, name))

    def _checked(self, fn, name):
        last_error = self._dll.ssmi_last_error
        last_error.restype = C.c_char_p

        def call(*a):
            st = fn(*a)
            if st != 0:
                raise SsmiError("{} failed ({}): {}".format(name, st, (last_error() or b"").decode()))
            return st
        call.__name__ = name
        return call


class _TypedView:
    """The library as seen by a network whose activations are bfloat16: every entry point that has a `_bf16` twin resolves
    to it, everything else (descriptor-typed convolutions / merges, optimizers, ...) to the plain entry point."""

    def __init__(self, base: "_Lib"):
        self._base = base

    def __getattr__(self, name):
        base = self.__dict__["_base"]
        fn = getattr(base, name + "_bf16", None) or getattr(base, name)
        setattr(self, name, fn)
        return fn


_lib = None


def lib_for(dtype_code: int):
    """lib() for float32 activations, its bfloat16 view for BF16."""
    L = lib()
    if dtype_code == F32:
        return L
    if dtype_code == BF16:
        if not hasattr(L, "_bf16_view"):
            L._bf16_view = _TypedView(L)
        return L._bf16_view
    raise ValueError("unknown activation dtype code {}".format(dtype_code))


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib(LIB_PATH)
    return _lib
