"""ctypes binding of libopnet_hip.so (the C ABI in include/opnet_hip.h).

There is no CPU or eager-PyTorch fallback: if the shared library is missing, loading fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_size_t, c_void_p

from . import build as _build

_LIB = None

OPNET_OK = 0
ABI_VERSION = 9                            # include/opnet_hip.h OPNET_HIP_ABI_VERSION
NO_OFFSET = ctypes.c_size_t(-1).value      # opnet_*_status_offset: "this shape never runs a persistent kernel"


class OpnetHipError(RuntimeError):
    pass


def _declare(lib):
    fp = c_void_p  # device pointers travel as integers (tensor.data_ptr())
    lib.opnet_hip_abi_version.restype = c_int
    lib.opnet_hip_abi_version.argtypes = []
    lib.opnet_last_error.restype = c_char_p
    lib.opnet_last_error.argtypes = []
    lib.opnet_packed_weights_bytes.restype = c_size_t
    lib.opnet_packed_weights_bytes.argtypes = [c_int, c_int]
    lib.opnet_pack_weights_f32.restype = c_int
    lib.opnet_pack_weights_f32.argtypes = [fp, fp, fp, fp, fp, fp, fp, c_size_t, c_int, c_int, c_void_p]
    lib.opnet_workspace_bytes.restype = c_size_t
    lib.opnet_workspace_bytes.argtypes = [c_int, c_int, c_int, c_int]
    lib.opnet_forward_f32.restype = c_int
    lib.opnet_forward_f32.argtypes = [fp, fp, fp, fp, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_void_p]
    lib.opnet_xcd_max_batch.restype = c_int
    lib.opnet_xcd_max_batch.argtypes = []
    lib.opnet_xcd_supported.restype = c_int
    lib.opnet_xcd_supported.argtypes = [c_int, c_int]
    lib.opnet_xcd_workspace_bytes.restype = c_size_t
    lib.opnet_xcd_workspace_bytes.argtypes = [c_int, c_int, c_int, c_int]
    lib.opnet_xcd_forward_f32.restype = c_int
    lib.opnet_xcd_forward_f32.argtypes = [fp, fp, fp, fp, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_void_p]
    lib.opnet_xcd_forward_multi_f32.restype = c_int
    lib.opnet_xcd_forward_multi_f32.argtypes = [POINTER(c_void_p), POINTER(c_int), c_int, fp, fp, fp, c_void_p, c_size_t,
                                                c_int, c_int, c_int, c_void_p]
    lib.opnet_xcd_profile.restype = c_int
    lib.opnet_xcd_profile.argtypes = [c_int]
    lib.opnet_xcd_profile_read.restype = c_int
    lib.opnet_xcd_profile_read.argtypes = [POINTER(c_double), POINTER(c_int)]
    lib.opnet_kernel_profile_read.restype = c_int
    lib.opnet_kernel_profile_read.argtypes = [c_int, POINTER(c_double), POINTER(c_int)]
    lib.opnet_xcd_set_trace.restype = None
    lib.opnet_xcd_set_trace.argtypes = [c_void_p]
    lib.opnet_xcd4_max_batch.restype = c_int
    lib.opnet_xcd4_max_batch.argtypes = []
    lib.opnet_xcd4_packed_weights_bytes.restype = c_size_t
    lib.opnet_xcd4_packed_weights_bytes.argtypes = [c_int, c_int]
    lib.opnet_xcd4_workspace_bytes.restype = c_size_t
    lib.opnet_xcd4_workspace_bytes.argtypes = [c_int, c_int, c_int, c_int]
    lib.opnet_xcd4_pack_weights_f32.restype = c_int
    lib.opnet_xcd4_pack_weights_f32.argtypes = [fp, fp, fp, fp, fp, fp, fp, c_size_t, c_int, c_int, c_void_p]
    lib.opnet_xcd4_forward_f32.restype = c_int
    lib.opnet_xcd4_forward_f32.argtypes = [fp, fp, fp, fp, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_void_p]
    lib.opnet_xcd4_set_trace.restype = None
    lib.opnet_xcd4_set_trace.argtypes = [c_void_p]
    lib.opnet_xcd4_last_status.restype = c_int
    lib.opnet_xcd4_last_status.argtypes = [POINTER(ctypes.c_uint)]
    lib.opnet_plan_create.restype = c_int
    lib.opnet_plan_create.argtypes = [POINTER(c_void_p), c_int, c_int, c_int, c_int]
    lib.opnet_plan_forward.restype = c_int
    lib.opnet_plan_forward.argtypes = [c_void_p, fp, fp, fp, fp, c_void_p, c_size_t, c_void_p]
    lib.opnet_plan_destroy.restype = None
    lib.opnet_plan_destroy.argtypes = [c_void_p]
    lib.opnet_train_packed_weights_bytes.restype = c_size_t
    lib.opnet_train_packed_weights_bytes.argtypes = [c_int, c_int]
    lib.opnet_train_pack_weights_f32.restype = c_int
    lib.opnet_train_pack_weights_f32.argtypes = [fp, fp, fp, fp, fp, fp, fp, c_size_t, c_int, c_int, c_void_p]
    lib.opnet_train_workspace_bytes.restype = c_size_t
    lib.opnet_train_workspace_bytes.argtypes = [c_int, c_int, c_int, c_int]
    lib.opnet_train_forward_f32.restype = c_int
    lib.opnet_train_forward_f32.argtypes = [fp, fp, fp, fp, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_void_p]
    lib.opnet_train_backward_f32.restype = c_int
    lib.opnet_train_backward_f32.argtypes = [fp, fp, c_void_p, c_size_t, fp, fp, fp, fp, fp, fp,
                                             c_int, c_int, c_int, c_int, c_void_p]
    lib.opnet_l1_loss_f32.restype = c_int
    lib.opnet_l1_loss_f32.argtypes = [fp, fp, fp, fp, ctypes.c_long, c_void_p, c_size_t, c_void_p]
    lib.opnet_smooth_l1_loss_f32.restype = c_int
    lib.opnet_smooth_l1_loss_f32.argtypes = [fp, fp, fp, fp, ctypes.c_long, c_float, c_void_p, c_size_t, c_void_p]
    lib.opnet_adam_multi_step_f32.restype = c_int
    lib.opnet_adam_multi_step_f32.argtypes = [c_int, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                              POINTER(ctypes.c_long), c_float, c_float, c_float, c_float, c_int, c_float, c_void_p]
    lib.opnet_adam_multi_step_guarded_f32.restype = c_int
    lib.opnet_adam_multi_step_guarded_f32.argtypes = lib.opnet_adam_multi_step_f32.argtypes[:-1] + [fp, fp, fp, c_void_p]
    lib.opnet_xcd4_status_offset.restype = c_size_t
    lib.opnet_xcd4_status_offset.argtypes = [c_int, c_int, c_int, c_int]
    lib.opnet_train_status_offset.restype = c_size_t
    lib.opnet_train_status_offset.argtypes = [c_int, c_int, c_int, c_int]
    lib.opnet_xcd4_enable.restype = None
    lib.opnet_xcd4_enable.argtypes = [c_int]
    lib.opnet_xcd4_enabled.restype = c_int
    lib.opnet_xcd4_enabled.argtypes = []
    lib.opnet_adam_step_f32.restype = c_int
    lib.opnet_adam_step_f32.argtypes = [fp, fp, fp, fp, ctypes.c_long, c_float, c_float, c_float, c_float,
                                        c_int, c_float, c_void_p]
    lib.opnet_mlp_pack_weights_f32.restype = c_int
    lib.opnet_mlp_pack_weights_f32.argtypes = [fp, fp, fp, fp, fp, fp, c_size_t, c_int, c_int, c_void_p]
    lib.opnet_mlp_forward_f32.restype = c_int
    lib.opnet_mlp_forward_f32.argtypes = [fp, fp, fp, fp, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_void_p]
    lib.opnet_mlp_train_pack_weights_f32.restype = c_int
    lib.opnet_mlp_train_pack_weights_f32.argtypes = [fp, fp, fp, fp, fp, fp, c_size_t, fp, c_int, c_int, c_void_p]
    lib.opnet_mlp_train_forward_f32.restype = c_int
    lib.opnet_mlp_train_forward_f32.argtypes = [fp, fp, fp, fp, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_void_p]
    lib.opnet_mlp_train_backward_f32.restype = c_int
    lib.opnet_mlp_train_backward_f32.argtypes = [fp, fp, c_void_p, c_size_t, fp, fp, fp, fp, fp,
                                                 c_int, c_int, c_int, c_int, c_void_p]
    lib.opseq_lstm_stack_packed_bytes.restype = c_size_t
    lib.opseq_lstm_stack_packed_bytes.argtypes = [c_int, c_int, c_int]
    lib.opseq_lstm_stack_workspace_bytes.restype = c_size_t
    lib.opseq_lstm_stack_workspace_bytes.argtypes = [c_int, c_int, c_int, c_int, c_int]
    lib.opseq_lstm_stack_pack_weights_f32.restype = c_int
    lib.opseq_lstm_stack_pack_weights_f32.argtypes = [POINTER(c_void_p), POINTER(c_void_p), fp, fp, c_size_t,
                                                      c_int, c_int, c_int, c_void_p]
    lib.opseq_lstm_stack_forward_f32.restype = c_int
    lib.opseq_lstm_stack_forward_f32.argtypes = [fp, fp, fp, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.opseq_lstm_stack_forward_graph_f32.restype = c_int
    lib.opseq_lstm_stack_forward_graph_f32.argtypes = lib.opseq_lstm_stack_forward_f32.argtypes
    lib.opseq_xcd_supported.restype = c_int
    lib.opseq_xcd_supported.argtypes = [c_int, c_int, c_int]
    lib.opseq_xcd_enable.restype = None
    lib.opseq_xcd_enable.argtypes = [c_int]
    lib.opseq_xcd_max_batch.restype = c_int
    lib.opseq_xcd_max_batch.argtypes = [c_int]
    lib.opseq_xcd_packed_bytes.restype = c_size_t
    lib.opseq_xcd_packed_bytes.argtypes = [c_int, c_int, c_int]
    lib.opseq_xcd_workspace_bytes.restype = c_size_t
    lib.opseq_xcd_workspace_bytes.argtypes = [c_int, c_int, c_int, c_int, c_int]
    lib.opseq_xcd_status_offset.restype = c_size_t
    lib.opseq_xcd_status_offset.argtypes = [c_int, c_int, c_int, c_int, c_int]
    lib.opseq_xcd_pack_weights_f32.restype = c_int
    lib.opseq_xcd_pack_weights_f32.argtypes = [POINTER(c_void_p), POINTER(c_void_p), fp, c_size_t, c_int, c_int, c_int, c_void_p]
    lib.opseq_xcd_forward_f32.restype = c_int
    lib.opseq_xcd_forward_f32.argtypes = [fp, fp, fp, fp, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.opseq_lstm_stack_train_status_offset.restype = c_size_t
    lib.opseq_lstm_stack_train_status_offset.argtypes = [c_int, c_int, c_int, c_int, c_int]
    lib.opseq_xcdt_supported.restype = c_int
    lib.opseq_xcdt_supported.argtypes = [c_int, c_int, c_int]
    lib.opseq_xcdt_enable.restype = None
    lib.opseq_xcdt_enable.argtypes = [c_int]
    lib.opseq_xcdt_max_batch.restype = c_int
    lib.opseq_xcdt_max_batch.argtypes = [c_int, c_int, c_int, c_int]
    lib.opseq_xcdt_packed_bytes.restype = c_size_t
    lib.opseq_xcdt_packed_bytes.argtypes = [c_int, c_int, c_int]
    lib.opseq_xcdt_workspace_bytes.restype = c_size_t
    lib.opseq_xcdt_workspace_bytes.argtypes = [c_int, c_int, c_int, c_int, c_int]
    lib.opseq_xcdt_status_offset.restype = c_size_t
    lib.opseq_xcdt_status_offset.argtypes = [c_int, c_int, c_int, c_int, c_int]
    lib.opseq_xcdt_pack_weights_f32.restype = c_int
    lib.opseq_xcdt_pack_weights_f32.argtypes = [POINTER(c_void_p), POINTER(c_void_p), fp, c_size_t, c_int, c_int, c_int, c_void_p]
    lib.opseq_xcdt_forward_f32.restype = c_int
    lib.opseq_xcdt_forward_f32.argtypes = [fp, fp, fp, fp, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.opseq_xcdt_set_trace.restype = None
    lib.opseq_xcdt_set_trace.argtypes = [c_void_p]
    lib.opseq_xcd_set_trace.restype = None
    lib.opseq_xcd_set_trace.argtypes = [c_void_p]
    lib.opseq_graph_cache_clear.restype = None
    lib.opseq_graph_cache_clear.argtypes = []
    lib.opseq_lstm_stack_train_packed_bytes.restype = c_size_t
    lib.opseq_lstm_stack_train_packed_bytes.argtypes = [c_int, c_int, c_int]
    lib.opseq_lstm_stack_train_workspace_bytes.restype = c_size_t
    lib.opseq_lstm_stack_train_workspace_bytes.argtypes = [c_int, c_int, c_int, c_int, c_int]
    lib.opseq_lstm_stack_train_pack_weights_f32.restype = c_int
    lib.opseq_lstm_stack_train_pack_weights_f32.argtypes = lib.opseq_lstm_stack_pack_weights_f32.argtypes
    lib.opseq_lstm_stack_train_forward_f32.restype = c_int
    lib.opseq_lstm_stack_train_forward_f32.argtypes = lib.opseq_lstm_stack_forward_f32.argtypes
    lib.opseq_lstm_stack_train_backward_f32.restype = c_int
    lib.opseq_lstm_stack_train_backward_f32.argtypes = [fp, fp, c_void_p, c_size_t, POINTER(c_void_p), POINTER(c_void_p),
                                                        fp, fp, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.opseq_slot_embed_relu_bwd_f32.restype = c_int
    lib.opseq_slot_embed_relu_bwd_f32.argtypes = [fp, fp, fp, fp, ctypes.c_long, c_int, c_int, c_void_p]
    lib.opseq_slot_embed_bwd_workspace_bytes.restype = c_size_t
    lib.opseq_slot_embed_bwd_workspace_bytes.argtypes = [ctypes.c_long, c_int, c_int]
    lib.opseq_slot_embed_relu_bwd_ws_f32.restype = c_int
    lib.opseq_slot_embed_relu_bwd_ws_f32.argtypes = [fp, fp, fp, fp, ctypes.c_long, c_int, c_int, c_void_p, c_size_t, c_void_p]
    lib.opseq_slot_embed_relu_f32.restype = c_int
    lib.opseq_slot_embed_relu_f32.argtypes = [fp, fp, fp, ctypes.c_long, c_int, c_int, c_void_p]
    lib.opseq_encoder_workspace_bytes.restype = c_size_t
    lib.opseq_encoder_workspace_bytes.argtypes = [ctypes.c_long, c_int, c_int, c_int]
    lib.opseq_attention_f32.restype = c_int
    lib.opseq_attention_f32.argtypes = [fp, fp, ctypes.c_long, c_int, c_int, c_void_p, c_size_t, c_void_p]
    lib.opseq_attention_workspace_bytes.restype = c_size_t
    lib.opseq_attention_workspace_bytes.argtypes = [ctypes.c_long, c_int, c_int]
    lib.opseq_encoder_train_saved_bytes.restype = c_size_t
    lib.opseq_encoder_train_saved_bytes.argtypes = [ctypes.c_long, c_int, c_int, c_int]
    lib.opseq_encoder_train_scratch_bytes.restype = c_size_t
    lib.opseq_encoder_train_scratch_bytes.argtypes = [ctypes.c_long, c_int, c_int, c_int]
    lib.opseq_encoder_layer_train_forward_f32.restype = c_int
    lib.opseq_encoder_layer_train_forward_f32.argtypes = [fp] * 14 + [c_void_p, c_size_t, c_void_p, c_size_t, ctypes.c_long,
                                                          c_int, c_int, c_int, c_float, ctypes.c_ulonglong, c_void_p]
    lib.opseq_encoder_layer_train_backward_f32.restype = c_int
    lib.opseq_encoder_layer_train_backward_f32.argtypes = [fp] * 20 + [c_void_p, c_size_t, c_void_p, c_size_t, ctypes.c_long,
                                                           c_int, c_int, c_int, c_float, ctypes.c_ulonglong, c_void_p]
    lib.opseq_encoder_test_masks_set.restype = c_int
    lib.opseq_encoder_test_masks_set.argtypes = [c_int, ctypes.c_ulonglong, fp, fp, fp, fp]
    lib.opseq_encoder_test_masks_clear.restype = c_int
    lib.opseq_encoder_test_masks_clear.argtypes = []
    lib.opdet_test_sort_scratch_bytes.restype = c_size_t
    lib.opdet_test_sort_scratch_bytes.argtypes = [ctypes.c_long]
    lib.opdet_test_sort_pairs.restype = c_int
    lib.opdet_test_sort_pairs.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_long, c_int, c_int, c_void_p, c_size_t,
                                          ctypes.POINTER(c_int), c_void_p]
    lib.opseq_encoder_layer_f32.restype = c_int
    lib.opseq_encoder_layer_f32.argtypes = [fp] * 13 + [c_void_p, c_size_t, ctypes.c_long, c_int, c_int, c_int, c_void_p]
    lib.opseq_encoder_layer_segmented_f32.restype = c_int
    lib.opseq_encoder_layer_segmented_f32.argtypes = [fp] * 13 + [c_void_p, c_size_t, ctypes.c_long, c_int, c_int, c_int, c_int,
                                                      c_void_p]
    lib.opseq_encoder_layer_batched_f32.restype = c_int
    lib.opseq_encoder_layer_batched_f32.argtypes = [fp] * 13 + [c_void_p, c_size_t, ctypes.c_long, c_int, c_int, c_int, c_int,
                                                    c_void_p]
    lib.opseq_ffn_fused_supported.restype = c_int
    lib.opseq_ffn_fused_supported.argtypes = [ctypes.c_long, c_int, c_int]
    lib.opseq_ffn_fused_plan.restype = c_int
    lib.opseq_ffn_fused_plan.argtypes = [ctypes.c_long, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_uint)]
    lib.opseq_ffn_fused_f32.restype = c_int
    lib.opseq_ffn_fused_f32.argtypes = [fp] * 6 + [ctypes.c_long, c_int, c_int, c_void_p]
    lib.opdet_conv2d_f32.restype = c_int
    lib.opdet_conv2d_f32.argtypes = [fp, fp, fp, fp, fp] + [c_int] * 11 + [c_void_p]
    lib.opdet_conv2d_workspace_bytes.restype = c_size_t
    lib.opdet_conv2d_workspace_bytes.argtypes = [c_int] * 10
    lib.opdet_conv2d_dual_workspace_bytes.restype = ctypes.c_longlong
    lib.opdet_conv2d_dual_workspace_bytes.argtypes = [c_int] * 9
    lib.opdet_conv2d_dual_f32.restype = c_int
    lib.opdet_conv2d_dual_f32.argtypes = [fp, fp, fp, fp, fp] + [c_int] * 10 + [c_void_p, c_size_t, c_void_p]
    lib.opdet_conv2d_up_f32.restype = c_int
    lib.opdet_conv2d_up_f32.argtypes = [fp, fp, fp, fp, fp] + [c_int] * 12 + [c_void_p, c_size_t, c_void_p]
    lib.opdet_conv2d_ws_f32.restype = c_int
    lib.opdet_conv2d_ws_f32.argtypes = [fp, fp, fp, fp, fp] + [c_int] * 11 + [c_void_p, c_size_t, c_void_p]
    lib.opdet_wino_weights_bytes.restype = c_size_t
    lib.opdet_wino_weights_bytes.argtypes = [c_int, c_int]
    lib.opdet_wino_weights_f32.restype = c_int
    lib.opdet_wino_weights_f32.argtypes = [fp, fp, c_int, c_int, c_int, c_void_p]
    lib.opdet_conv2d_wino_workspace_bytes.restype = c_size_t
    lib.opdet_conv2d_wino_workspace_bytes.argtypes = [c_int] * 5
    lib.opdet_conv2d_wino_f32.restype = c_int
    lib.opdet_conv2d_wino_f32.argtypes = [fp, fp, fp, fp] + [c_int] * 6 + [c_void_p, c_size_t, c_void_p]
    lib.opdet_maxpool3x3s2_f32.restype = c_int
    lib.opdet_maxpool3x3s2_f32.argtypes = [fp, fp, c_int, c_int, c_int, c_int, c_void_p]
    lib.opdet_subsample2_f32.restype = c_int
    lib.opdet_subsample2_f32.argtypes = [fp, fp, c_int, c_int, c_int, c_int, c_void_p]
    lib.opdet_upsample_add_f32.restype = c_int
    lib.opdet_upsample_add_f32.argtypes = [fp, fp, fp, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.opdet_preprocess_frame_f32.restype = c_int
    lib.opdet_preprocess_frame_f32.argtypes = [fp, fp, c_int, c_int, c_int, c_int, c_int, c_int,
                                               POINTER(c_float), POINTER(c_float), c_void_p]
    ip = POINTER(c_int)
    lib.opdet_rpn_workspace_bytes.restype = c_size_t
    lib.opdet_rpn_workspace_bytes.argtypes = [c_int, ip, ip, ip, c_int, c_int, c_int]
    lib.opdet_rpn_proposals_f32.restype = c_int
    lib.opdet_rpn_proposals_f32.argtypes = [POINTER(c_void_p), c_int, ip, ip, ip, c_int, c_int, c_int, c_int, c_int, c_int,
                                            c_float, c_float, fp, fp, fp, c_void_p, c_size_t, c_void_p]
    lib.opdet_roi_align_f32.restype = c_int
    lib.opdet_roi_align_f32.argtypes = [POINTER(c_void_p), ip, ip, c_int, c_int, fp, fp, c_int, fp, c_void_p]
    lib.opdet_detections_workspace_bytes.restype = c_size_t
    lib.opdet_detections_workspace_bytes.argtypes = [c_int, c_int]
    lib.opdet_detections_f32.restype = c_int
    lib.opdet_detections_f32.argtypes = [fp, fp, fp, fp, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_int,
                                         fp, fp, fp, fp, c_void_p, c_size_t, c_void_p]
    lib.opdet_rpn_workspace_bytes_batch.restype = c_size_t
    lib.opdet_rpn_workspace_bytes_batch.argtypes = [c_int, c_int, ip, ip, ip, c_int, c_int, c_int]
    lib.opdet_rpn_proposals_batch_f32.restype = c_int
    lib.opdet_rpn_proposals_batch_f32.argtypes = [POINTER(c_void_p), c_int, c_int, ip, ip, ip, c_int, c_int, c_int, c_int, c_int, c_int,
                                                  c_float, c_float, fp, fp, fp, c_void_p, c_size_t, c_void_p]
    lib.opdet_roi_align_batch_f32.restype = c_int
    lib.opdet_roi_align_batch_f32.argtypes = [POINTER(c_void_p), c_int, ip, ip, c_int, c_int, fp, fp, c_int, fp, c_void_p]
    lib.opdet_detections_workspace_bytes_batch.restype = c_size_t
    lib.opdet_detections_workspace_bytes_batch.argtypes = [c_int, c_int, c_int]
    lib.opdet_detections_batch_f32.restype = c_int
    lib.opdet_detections_batch_f32.argtypes = [fp, fp, fp, fp] + [c_int] * 9 + [c_float, c_float, c_int,
                                               fp, fp, fp, fp, c_void_p, c_size_t, c_void_p]
    lib.opnet_encode_clips_f32.restype = c_int
    lib.opnet_encode_clips_f32.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int,
                                           c_void_p, c_void_p]
    lib.opnet_load_clips_f32.restype = c_int
    lib.opnet_load_clips_f32.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                         c_char_p, c_int]
    lib.opnet_postprocess_iou.restype = c_int
    lib.opnet_postprocess_iou.argtypes = [fp, fp, fp, fp, fp, c_int, c_int, c_void_p]
    lib.opnet_dp_guard_f32.restype = c_int
    lib.opnet_dp_guard_f32.argtypes = [fp, fp, fp, c_float, c_void_p]


EXPORTS = [
    "opnet_hip_abi_version", "opnet_last_error", "opnet_packed_weights_bytes", "opnet_pack_weights_f32",
    "opnet_workspace_bytes", "opnet_forward_f32", "opnet_plan_create", "opnet_plan_forward",
    "opnet_plan_destroy", "opnet_postprocess_iou", "opnet_encode_clips_f32", "opnet_load_clips_f32",
    "opnet_xcd_max_batch", "opnet_xcd_supported", "opnet_xcd_workspace_bytes", "opnet_xcd_forward_f32", "opnet_xcd_forward_multi_f32", "opnet_xcd_set_trace", "opnet_xcd4_set_trace", "opnet_xcd4_last_status", "opnet_xcd4_max_batch", "opnet_xcd4_packed_weights_bytes",
    "opnet_xcd4_workspace_bytes", "opnet_xcd4_pack_weights_f32", "opnet_xcd4_forward_f32",
    "opnet_xcd_profile", "opnet_xcd_profile_read", "opnet_kernel_profile_read",
    "opnet_train_packed_weights_bytes", "opnet_train_pack_weights_f32", "opnet_train_workspace_bytes",
    "opnet_train_forward_f32", "opnet_train_backward_f32", "opnet_l1_loss_f32", "opnet_smooth_l1_loss_f32", "opnet_adam_step_f32", "opnet_adam_multi_step_f32",
    "opnet_adam_multi_step_guarded_f32", "opnet_dp_guard_f32", "opnet_xcd4_status_offset", "opnet_train_status_offset", "opnet_xcd4_enable", "opnet_xcd4_enabled",
    "opnet_mlp_pack_weights_f32", "opnet_mlp_forward_f32",
    "opnet_mlp_train_pack_weights_f32", "opnet_mlp_train_forward_f32", "opnet_mlp_train_backward_f32",
    "opseq_lstm_stack_packed_bytes", "opseq_lstm_stack_workspace_bytes", "opseq_lstm_stack_pack_weights_f32",
    "opseq_lstm_stack_forward_f32", "opseq_lstm_stack_forward_graph_f32", "opseq_graph_cache_clear",
    "opseq_xcd_supported", "opseq_xcd_enable", "opseq_xcd_max_batch", "opseq_xcd_packed_bytes", "opseq_xcd_workspace_bytes",
    "opseq_xcd_status_offset", "opseq_xcd_pack_weights_f32", "opseq_xcd_forward_f32", "opseq_lstm_stack_train_status_offset", "opseq_xcd_set_trace",
    "opseq_xcdt_supported", "opseq_xcdt_enable", "opseq_xcdt_max_batch", "opseq_xcdt_packed_bytes", "opseq_xcdt_workspace_bytes",
    "opseq_xcdt_status_offset", "opseq_xcdt_pack_weights_f32", "opseq_xcdt_forward_f32", "opseq_xcdt_set_trace",
    "opseq_slot_embed_relu_f32", "opseq_slot_embed_relu_bwd_f32", "opseq_slot_embed_bwd_workspace_bytes", "opseq_slot_embed_relu_bwd_ws_f32",
    "opseq_lstm_stack_train_packed_bytes", "opseq_lstm_stack_train_workspace_bytes",
    "opseq_lstm_stack_train_pack_weights_f32", "opseq_lstm_stack_train_forward_f32",
    "opseq_lstm_stack_train_backward_f32",
    "opseq_encoder_workspace_bytes",
    "opseq_encoder_layer_f32", "opseq_encoder_layer_segmented_f32", "opseq_encoder_layer_batched_f32", "opseq_ffn_fused_supported", "opseq_ffn_fused_plan", "opseq_ffn_fused_f32", "opseq_attention_f32", "opseq_attention_workspace_bytes",
    "opseq_encoder_train_saved_bytes", "opseq_encoder_train_scratch_bytes", "opseq_encoder_layer_train_forward_f32",
    "opseq_encoder_layer_train_backward_f32", "opseq_encoder_test_masks_set", "opseq_encoder_test_masks_clear",
    "opdet_conv2d_f32", "opdet_conv2d_workspace_bytes", "opdet_conv2d_ws_f32", "opdet_wino_weights_bytes", "opdet_wino_weights_f32", "opdet_conv2d_wino_workspace_bytes", "opdet_conv2d_wino_f32", "opdet_conv2d_up_f32", "opdet_conv2d_dual_workspace_bytes", "opdet_conv2d_dual_f32", "opdet_maxpool3x3s2_f32", "opdet_subsample2_f32", "opdet_upsample_add_f32",
    "opdet_preprocess_frame_f32", "opdet_rpn_workspace_bytes", "opdet_rpn_proposals_f32", "opdet_roi_align_f32",
    "opdet_detections_workspace_bytes", "opdet_detections_f32", "opdet_rpn_workspace_bytes_batch", "opdet_rpn_proposals_batch_f32",
    "opdet_roi_align_batch_f32", "opdet_detections_workspace_bytes_batch", "opdet_detections_batch_f32", "opdet_test_sort_scratch_bytes", "opdet_test_sort_pairs",
]


def lib_path() -> str:
    # OPNET_HIP_LIB: load another build of the same ABI (kernel-variant experiments, tools/)
    return os.environ.get("OPNET_HIP_LIB") or _build.LIB


def load():
    """Load (once) and return the ctypes handle. Raises OpnetHipError if the library is absent."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise OpnetHipError(
                f"{path} not found: build it with `python -m objectpermanence_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for this path.")
        lib = ctypes.CDLL(path)
        lib.opnet_hip_abi_version.restype = c_int
        lib.opnet_hip_abi_version.argtypes = []
        found = int(lib.opnet_hip_abi_version())
        if found != ABI_VERSION:
            raise OpnetHipError(f"{path} implements ABI version {found}, this package needs {ABI_VERSION}: rebuild it with "
                                "`python -m objectpermanence_amd.build`")
        _declare(lib)
        _LIB = lib
    return _LIB


def check(rc: int, what: str) -> None:
    if rc != OPNET_OK:
        msg = load().opnet_last_error().decode(errors="replace")
        raise OpnetHipError(f"{what} failed (code {rc}): {msg}")
