"""ctypes binding of ``libfpsg_hip.so`` (C ABI: ``include/fpsg_hip.h``).

There is NO fallback: if the HIP library is missing, or a tensor is not a contiguous fp32
/ int32 tensor on a ROCm device, the call raises.  Tensors are passed as raw device
pointers together with torch's current HIP stream, so the kernels are ordered with the
surrounding PyTorch work without any synchronisation.
"""
from __future__ import annotations

import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# FPSG_HIP_LIB: another build of the same library (same-box A/B of kernel versions, tools/ab_lib.sh); never a fallback
LIB_PATH = os.environ.get("FPSG_HIP_LIB") or os.path.join(_HERE, "libfpsg_hip.so")

_c_f32p = ctypes.c_void_p
_c_i32p = ctypes.c_void_p
_c_int = ctypes.c_int
_c_stream = ctypes.c_void_p

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/fpsg_hip.h
SIGNATURES = {
    "fpsg_version": [],
    "fpsg_last_error": [],
    "fpsg_chamfer_fwd": [_c_f32p, _c_f32p, _c_int, _c_int, _c_int,
                         _c_f32p, _c_i32p, _c_f32p, _c_i32p, _c_stream],
    "fpsg_chamfer_bwd": [_c_f32p, _c_f32p, _c_i32p, _c_i32p, _c_f32p, _c_f32p,
                         _c_int, _c_int, _c_int, _c_f32p, _c_f32p, _c_stream],
    "fpsg_chamfer_losses": [_c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, ctypes.c_float, ctypes.c_float, _c_f32p,
                            _c_stream],
    "fpsg_chamfer_loss_grads": [_c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, ctypes.c_float,
                                ctypes.c_float, _c_f32p, _c_f32p, _c_stream],
    "fpsg_chamfer_fwd_variant": [_c_f32p, _c_f32p, _c_int, _c_int, _c_int,
                                 _c_f32p, _c_i32p, _c_f32p, _c_i32p, _c_int, _c_stream],
    "fpsg_chamfer_workspace_bytes": [_c_int, _c_int, _c_int, _c_int],
    "fpsg_chamfer_fwd_tiled": [_c_f32p, _c_f32p, _c_int, _c_int, _c_int,
                               _c_f32p, _c_i32p, _c_f32p, _c_i32p, ctypes.c_void_p, ctypes.c_size_t, _c_int,
                               _c_stream],
    "fpsg_chamfer_fwd_tiled_losses": [_c_f32p, _c_f32p, _c_int, _c_int, _c_int,
                                      _c_f32p, _c_i32p, _c_f32p, _c_i32p, ctypes.c_void_p, ctypes.c_size_t, _c_int,
                                      _c_int, ctypes.c_float, ctypes.c_float, _c_f32p, _c_stream],
    "fpsg_chamfer_bwd_losses": [_c_f32p, _c_f32p, _c_i32p, _c_i32p, _c_f32p, _c_f32p, _c_f32p,
                                _c_int, _c_int, _c_int, _c_int, ctypes.c_float, ctypes.c_float, _c_f32p, _c_f32p,
                                _c_stream],
    "fpsg_chamfer_bwd_sorted": [_c_f32p, _c_f32p, _c_i32p, _c_i32p, _c_f32p, _c_f32p,
                                _c_int, _c_int, _c_int, _c_f32p, _c_f32p, _c_stream],
    "fpsg_chamfer_bwd_scan": [_c_f32p, _c_f32p, _c_i32p, _c_i32p, _c_f32p, _c_f32p,
                              _c_int, _c_int, _c_int, _c_f32p, _c_f32p, _c_stream],
    "fpsg_knn_workspace_floats": [_c_int, _c_int, _c_int],
    "fpsg_knn": [_c_f32p, _c_int, _c_int, _c_int, _c_int, _c_i32p, _c_f32p, _c_stream],
    "fpsg_knn_ex": [_c_f32p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_i32p, _c_f32p, _c_int, _c_stream],
    "fpsg_edge_feature_fwd": [_c_f32p, _c_i32p, _c_int, _c_int, _c_int, _c_int, _c_f32p, _c_stream],
    "fpsg_edge_feature_bwd": [_c_f32p, _c_i32p, _c_int, _c_int, _c_int, _c_int, _c_f32p, _c_stream],
    "fpsg_edgeconv_blocks": [_c_int, _c_int, _c_int],
    "fpsg_edgeconv_reverse_graph_fits": [_c_int, _c_int],
    "fpsg_edgeconv_reverse_graph": [_c_i32p, _c_int, _c_int, _c_int, _c_i32p, _c_i32p, _c_stream],
    "fpsg_edgeconv_prep_blocks": [ctypes.c_long],
    "fpsg_edgeconv_stats_finalize": [_c_f32p, _c_int, _c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_float, ctypes.c_float,
                                     ctypes.c_double, _c_int, _c_int, _c_f32p, _c_stream],
    "fpsg_edgeconv_stats_ws_floats": [_c_int, _c_int],
    "fpsg_edgeconv_stats_finalize_ws": [_c_f32p, _c_int, _c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_float, ctypes.c_float,
                                        ctypes.c_double, _c_int, _c_int, _c_f32p, _c_f32p, _c_stream],
    "fpsg_edgeconv_bwd_finalize": [_c_f32p, _c_int, _c_f32p, ctypes.c_double, _c_int, _c_int, _c_f32p, _c_f32p, _c_f32p,
                                   _c_stream],
    "fpsg_edgeconv_act": [_c_f32p, _c_f32p, _c_f32p, ctypes.c_float, ctypes.c_long, _c_int, _c_f32p, _c_stream],
    "fpsg_edgeconv_bwd_prep": [_c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_float, ctypes.c_long, _c_int, _c_f32p, _c_f32p,
                               _c_stream],
    "fpsg_edgeconv_fwd": [_c_f32p, _c_i32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_f32p, _c_f32p,
                          _c_f32p, _c_f32p, _c_stream],
    "fpsg_edgeconv_bwd": [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_i32p, _c_i32p, _c_f32p, _c_int, _c_int,
                          _c_int, _c_int, _c_f32p, _c_stream],
    "fpsg_softmin": [_c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, ctypes.c_float, _c_f32p, _c_stream],
    "fpsg_sinkhorn_workspace_floats": [_c_int, _c_int, _c_int],
    "fpsg_sinkhorn_divergence": [_c_f32p, _c_f32p, _c_int, _c_int, _c_int, ctypes.c_void_p, _c_int, _c_f32p, _c_f32p,
                                 _c_stream],
    "fpsg_dec1_tiles": [_c_int],
    "fpsg_dec1_fwd": [_c_f32p, _c_f32p, _c_int, _c_int, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_int, _c_int,
                      _c_int, _c_int, _c_int, ctypes.c_float, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_stream],
    "fpsg_dec1_bwd": [_c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int,
                      _c_int, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_stream],
    "fpsg_dec1_fwd_ld": [_c_f32p, _c_int, _c_f32p, _c_int, _c_int, _c_f32p, _c_int, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_int,
                         _c_int, _c_int, _c_int, _c_int, ctypes.c_float, _c_f32p, _c_int, _c_f32p, _c_f32p, _c_f32p,
                         _c_stream],
    "fpsg_dec1_bwd_ld": [_c_f32p, _c_int, _c_f32p, _c_int, _c_f32p, _c_int, _c_int, _c_f32p, _c_int, _c_f32p, _c_int, _c_int,
                         _c_int, _c_int, _c_int, _c_int, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_stream],
    "fpsg_bn_act_rows_fwd": [_c_f32p, _c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), _c_int,
                             _c_f32p, _c_f32p, _c_f32p, _c_int, ctypes.c_float, _c_int, ctypes.c_float, _c_f32p,
                             _c_f32p, _c_f32p, _c_stream],
    "fpsg_bn_act_rows_bwd": [_c_f32p, _c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), _c_int,
                             _c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, ctypes.c_float, _c_f32p, _c_f32p, _c_f32p,
                             _c_f32p, _c_stream],
    "fpsg_bn_workspace_floats": [_c_int, _c_int, _c_int],
    "fpsg_bn_act_fwd": [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_float, _c_int, _c_int, _c_int, _c_int,
                        ctypes.c_float, _c_int, ctypes.c_float, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p,
                        _c_stream],
    "fpsg_bn_act_bwd": [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_int,
                        ctypes.c_float, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_stream],
    "fpsg_bn_act_bwd_coef": [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_int, ctypes.c_float,
                             _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_stream],
    "fpsg_bn_act_bwd_parts": [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_int,
                        ctypes.c_float, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_int, _c_stream],
    "fpsg_bn_pool_workspace_floats": [_c_int, _c_int, _c_int, _c_int],
    "fpsg_bn_act_pool_fwd": [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_float, _c_int, _c_int, _c_int, _c_int,
                             _c_int, ctypes.c_float, _c_int, ctypes.c_float, _c_f32p, _c_f32p, _c_f32p, _c_f32p,
                             _c_f32p, _c_f32p, _c_int, _c_stream],
    "fpsg_bn_act_pool_bwd": [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                             ctypes.c_float, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_stream],
    "fpsg_bn_max_workspace_floats": [_c_int, _c_int, _c_int],
    "fpsg_bn_max_dz_offset": [_c_int, _c_int, _c_int],
    "fpsg_bn_act_max_fwd": [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_float, _c_int, _c_int, _c_int, _c_int,
                            ctypes.c_float, _c_int, ctypes.c_float, _c_f32p, _c_i32p, _c_f32p, _c_f32p, _c_f32p,
                            _c_f32p, _c_stream],
    "fpsg_bn_act_max_bwd": [_c_f32p, _c_f32p, _c_f32p, _c_i32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_int,
                            ctypes.c_float, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_stream],
    "fpsg_max_bwd_prep": [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_f32p, _c_f32p,
                          _c_f32p, _c_f32p, _c_stream],
    "fpsg_max_bwd_dw": [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_f32p, _c_stream],
    "fpsg_max_bwd_gather": [_c_f32p, _c_f32p, _c_i32p, _c_int, _c_int, _c_int, _c_int, _c_f32p, _c_f32p, _c_stream],
    "fpsg_max_bwd_scatter": [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_i32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_f32p,
                             _c_stream],
    "fpsg_max_bwd_scatter_workspace_floats": [_c_int, _c_int, _c_int],
    "fpsg_bn_act_max_bwd_coef": [_c_f32p, _c_f32p, _c_f32p, _c_i32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_int,
                                 ctypes.c_float, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_stream],
    "fpsg_wino_input_transform": [_c_int, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_f32p, ctypes.c_long, _c_stream],
    "fpsg_wino_output_transform": [_c_int, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_f32p, ctypes.c_long, _c_stream],
    "fpsg_wino_stats_parts": [_c_int, _c_int, _c_int, _c_int],
    "fpsg_wino_output_transform_stats": [_c_int, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_f32p, _c_f32p, _c_f32p,
                                         ctypes.c_long, _c_stream],
    "fpsg_wino_grad_output_transform": [_c_int, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_f32p, ctypes.c_long, _c_stream],
    "fpsg_wino_grad_transforms": [_c_int, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_f32p, _c_f32p, ctypes.c_long, _c_stream],
    "fpsg_wino_filter_transform": [_c_int, _c_f32p, _c_int, _c_int, _c_int, _c_f32p, _c_stream],
    "fpsg_wino_output_transform_bwd_stats": [_c_int, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_f32p, _c_f32p, _c_f32p,
                                             _c_f32p, _c_f32p, ctypes.c_long, _c_stream],
    "fpsg_wino_filter_transform_batch": [ctypes.c_void_p, _c_int, ctypes.c_long, _c_stream],
    "fpsg_wino_filter_grad_transform": [_c_int, _c_f32p, _c_int, _c_int, _c_f32p, _c_stream],
    "fpsg_wino_conv_fused": [_c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f32p, _c_stream],
    "fpsg_wino_input_transform_act": [_c_int, _c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_f32p, ctypes.c_long,
                                      _c_stream],
    "fpsg_wino_conv_fused_act": [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f32p,
                                 _c_stream],
    "fpsg_wino_dw_fused_workspace_floats": [_c_int, _c_int, _c_int, _c_int],
    "fpsg_wino_dw_fused": [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f32p, _c_f32p,
                           _c_stream],
    "fpsg_wino_conv_fused_parts": [_c_int, _c_int, _c_int, _c_int],
    "fpsg_wino_conv_fused_stats": [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f32p,
                                   _c_f32p, _c_f32p, _c_stream],
    "fpsg_bn_stats": [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_float, _c_int, _c_int, _c_int, _c_int,
                      ctypes.c_float, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_int, _c_stream],
    "fpsg_conv_first_dw_workspace_floats": [_c_int, _c_int, _c_int],
    "fpsg_conv_first_parts": [_c_int, _c_int, _c_int],
    "fpsg_conv_first_fwd": [_c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f32p, _c_f32p, _c_f32p, _c_stream],
    "fpsg_conv_first_dw": [_c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f32p, _c_f32p, _c_stream],
    "fpsg_conv_first_dw_fold": [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_int,
                                _c_f32p, _c_f32p, _c_stream],
    "fpsg_adam_step": [_c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_size_t, ctypes.c_float, ctypes.c_float,
                       ctypes.c_float, ctypes.c_float, _c_int, ctypes.c_float, _c_stream],
    "fpsg_flat_accumulate_segments": [_c_f32p, ctypes.c_void_p, ctypes.c_void_p, _c_int, ctypes.c_size_t, _c_int, _c_stream],
    "fpsg_flat_accumulate_tables": [_c_f32p, ctypes.c_void_p, ctypes.c_void_p, _c_int, _c_int, ctypes.c_size_t, _c_int,
                                    _c_stream],
    "fpsg_adam_step_segments": [_c_f32p, ctypes.c_void_p, ctypes.c_void_p, _c_int, _c_f32p, _c_f32p, ctypes.c_size_t,
                                ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _c_int, ctypes.c_float,
                                _c_stream],
    "fpsg_gemm_split_workspace_floats": [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int],
    "fpsg_gemm_split": [_c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, ctypes.c_long,
                        ctypes.c_long, ctypes.c_long, _c_int, _c_int, _c_f32p, ctypes.c_size_t, _c_stream],
    "fpsg_gemm_split_packed_a_bytes": [_c_int, _c_int, _c_int, _c_int],
    "fpsg_gemm_split_pack_a": [_c_f32p, _c_int, _c_int, _c_int, _c_int, ctypes.c_long, _c_int, ctypes.c_void_p, _c_stream],
    "fpsg_gemm_split_nn_packed": [ctypes.c_void_p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                  ctypes.c_long, ctypes.c_long, _c_int, _c_stream],
    "fpsg_gemm_split_nn_persistent": [ctypes.c_void_p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                      ctypes.c_long, ctypes.c_long, _c_int, _c_stream],
    "fpsg_gemm_f32_nn": [_c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                         ctypes.c_long, ctypes.c_long, ctypes.c_long, _c_int, _c_stream],
    "fpsg_emd_workspace_floats": [_c_int, _c_int, _c_int],
    "fpsg_emd_approx": [_c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_f32p, _c_f32p, _c_f32p, _c_f32p,
                        _c_stream],
    "fpsg_emd_approx_variant": [_c_f32p, _c_f32p, _c_int, _c_int, _c_int, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_int,
                                _c_stream],
}
_RESTYPES = {"fpsg_last_error": ctypes.c_char_p, "fpsg_chamfer_workspace_bytes": ctypes.c_size_t,
             "fpsg_sinkhorn_workspace_floats": ctypes.c_size_t,
             "fpsg_knn_workspace_floats": ctypes.c_size_t,
             "fpsg_bn_workspace_floats": ctypes.c_size_t, "fpsg_bn_pool_workspace_floats": ctypes.c_size_t, "fpsg_bn_max_workspace_floats": ctypes.c_size_t, "fpsg_bn_max_dz_offset": ctypes.c_size_t, "fpsg_conv_first_dw_workspace_floats": ctypes.c_size_t, "fpsg_emd_workspace_floats": ctypes.c_size_t, "fpsg_max_bwd_scatter_workspace_floats": ctypes.c_size_t,
             "fpsg_wino_dw_fused_workspace_floats": ctypes.c_size_t, "fpsg_gemm_split_workspace_floats": ctypes.c_size_t, "fpsg_gemm_split_packed_a_bytes": ctypes.c_size_t, "fpsg_edgeconv_stats_ws_floats": ctypes.c_size_t}

_lib = None
_lock = threading.Lock()


class FpsgHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Loads the library once; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise FpsgHipError(
                        f"{LIB_PATH} not found: build it with `make -C fpsg_amd/csrc` "
                        "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                        "fpsg_amd has no CPU or PyTorch fallback for its HIP ops.")
                lib = ctypes.CDLL(LIB_PATH)
                for name, argtypes in SIGNATURES.items():
                    fn = getattr(lib, name)
                    fn.argtypes = argtypes
                    fn.restype = _RESTYPES.get(name, ctypes.c_int)
                _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().fpsg_last_error()
        raise FpsgHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def dev_tensor(t: torch.Tensor, dtype: torch.dtype, name: str) -> torch.Tensor:
    """Validates that `t` is something the C ABI accepts."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if t.device.type != "cuda":
        raise FpsgHipError(
            f"{name}: tensor is on '{t.device}'; fpsg_amd HIP ops run on a ROCm GPU only "
            "(no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t


def stream_of(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def ptr(t: torch.Tensor) -> int:
    return t.data_ptr()
