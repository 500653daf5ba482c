"""ctypes binding of libs2d_hip.so (C-ABI declared in include/s2d.h).

The product path has NO CPU fallback: if the shared object is missing or does not export a
symbol, importing/using the ops raises immediately (build it with `python -m sparse2dense_amd.build`).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libs2d_hip.so")

c_i32p = ctypes.c_void_p
c_f32p = ctypes.c_void_p
_I3 = ctypes.c_int32 * 3
_PFN_GEO = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float]   # pillars, slots, ndim, vx, vy, x/y offset
_F3 = ctypes.c_float * 3
_F6 = ctypes.c_float * 6

# name -> (restype, argtypes); mirrors include/s2d.h one to one
SIGNATURES = {
    "s2d_version": (ctypes.c_int, []),
    "s2d_last_error": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_size_t]),
    "s2d_build_info": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_size_t]),
    "s2d_debug_lds_fill": (ctypes.c_int, [ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_voxelize_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "s2d_voxelize_run": (ctypes.c_int, [c_f32p, ctypes.c_int64, ctypes.c_int, _F6, _F3, ctypes.c_int, ctypes.c_int,
                                        c_f32p, c_i32p, c_i32p, c_f32p, c_i32p, ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.c_void_p]),
    "s2d_voxelize_batch_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "s2d_voxelize_batch_run": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, _F6, _F3, ctypes.c_int, ctypes.c_int,
                                              c_f32p, c_i32p, c_i32p, c_f32p, c_i32p, c_i32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_rulebook_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, _I3, ctypes.c_int64]),
    "s2d_rulebook_subm_build": (ctypes.c_int, [c_i32p, ctypes.c_int64, ctypes.c_int, _I3, _I3, _I3, c_i32p, c_i32p,
                                               ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_rulebook_conv_count": (ctypes.c_int, [c_i32p, ctypes.c_int64, ctypes.c_int, _I3, _I3, _I3, _I3, _I3, c_i32p,
                                               ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_rulebook_conv_fill": (ctypes.c_int, [c_i32p, ctypes.c_int64, ctypes.c_int, _I3, _I3, _I3, _I3, _I3,
                                              ctypes.c_int64, c_i32p, c_i32p, c_i32p, c_i32p, ctypes.c_void_p,
                                              ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_conv2d_pack_batch_bf16": (ctypes.c_int, [ctypes.c_int] + [ctypes.c_void_p] * 9),
    "s2d_pfn_supported": (ctypes.c_int, [ctypes.c_int] * 4),
    "s2d_pfn_blocks": (ctypes.c_int, [ctypes.c_int64]),
    "s2d_pfn_bwd_cols": (ctypes.c_int, []),
    "s2d_pfn_stats_f32": (ctypes.c_int, [c_f32p, c_i32p, c_i32p, c_f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                         ctypes.c_float, ctypes.c_float, c_f32p, ctypes.c_void_p]),
    "s2d_pfn_apply_max_f32": (ctypes.c_int, [c_f32p, c_i32p, c_i32p, c_f32p, c_f32p, c_f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                             ctypes.c_float, ctypes.c_float, ctypes.c_float, c_f32p, ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_pfn_bwd_f32": (ctypes.c_int, [c_f32p, c_i32p, c_i32p, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                       ctypes.c_float, ctypes.c_float, ctypes.c_float, c_f32p, ctypes.c_void_p]),
    "s2d_pfn2_supported": (ctypes.c_int, [ctypes.c_int] * 5),
    "s2d_pfn2_bwd_rows": (ctypes.c_int, []),
    "s2d_pfn2_bwd_cols": (ctypes.c_int, []),
    "s2d_pfn2_stats1_f32": (ctypes.c_int, [c_f32p, c_i32p, c_i32p, c_f32p] + _PFN_GEO + [c_f32p, ctypes.c_void_p]),
    "s2d_pfn2_stats2_f32": (ctypes.c_int, [c_f32p, c_i32p, c_i32p, c_f32p, c_f32p, c_f32p] + _PFN_GEO + [c_f32p, ctypes.c_void_p]),
    "s2d_pfn2_apply_max_f32": (ctypes.c_int, [c_f32p, c_i32p, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p] + _PFN_GEO
                               + [c_f32p, ctypes.c_void_p, c_f32p, ctypes.c_void_p]),
    "s2d_pfn2_bwd_f32": (ctypes.c_int, [c_f32p, c_i32p, c_i32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_void_p] + _PFN_GEO
                         + [c_f32p, ctypes.c_void_p]),
    "s2d_rulebook_chain_supported": (ctypes.c_int, [ctypes.c_int, _I3, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_rulebook_chain_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, _I3, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                             ctypes.c_void_p]),
    "s2d_rulebook_chain_plan": (ctypes.c_int, [c_i32p, ctypes.c_int64, ctypes.c_int, _I3, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p, c_i32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_rulebook_chain_fill": (ctypes.c_int, [c_i32p, ctypes.c_int64, ctypes.c_int, _I3, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p, c_i32p, c_i32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_spconv_fwd_f32": (ctypes.c_int, [c_f32p, ctypes.c_int64, c_f32p, c_f32p, c_i32p, ctypes.c_int64, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, c_f32p, ctypes.c_void_p]),
    "s2d_spconv_bf16_supported": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "s2d_spconv_pack_weights_bf16": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                    ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_spconv_fwd_bf16": (ctypes.c_int, [c_f32p, ctypes.c_int64, ctypes.c_void_p, c_f32p, c_i32p, ctypes.c_int64,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, ctypes.c_void_p]),
    "s2d_spconv_wgrad_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "s2d_spconv_wgrad_f32": (ctypes.c_int, [c_f32p, ctypes.c_int64, c_f32p, c_i32p, ctypes.c_int64, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, c_f32p, ctypes.c_void_p, ctypes.c_size_t,
                                            ctypes.c_void_p]),
    "s2d_spconv_wgrad_bf16": (ctypes.c_int, [c_f32p, ctypes.c_int64, c_f32p, c_i32p, ctypes.c_int64, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, c_f32p, ctypes.c_void_p, ctypes.c_size_t,
                                             ctypes.c_void_p]),
    "s2d_bn1d_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int]),
    "s2d_bn1d_stats_f32": (ctypes.c_int, [c_f32p, ctypes.c_int64, ctypes.c_int, c_f32p, ctypes.c_void_p,
                                          ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_bn1d_finalize_fwd_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_float, ctypes.c_float, ctypes.c_int,
                                                 c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_void_p,
                                                 ctypes.c_void_p]),
    "s2d_bn1d_finalize_bwd_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int, c_f32p,
                                                 c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_void_p]),
    "s2d_bn1d_stats_finalize_f32": (ctypes.c_int, [c_f32p, ctypes.c_int64, ctypes.c_int, c_f32p, c_f32p, ctypes.c_float,
                                                   ctypes.c_float, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_bn1d_bwd_reduce_finalize_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                                        c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                                        c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_bn1d_apply_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                          c_f32p, ctypes.c_void_p]),
    "s2d_bn1d_bwd_reduce_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                               c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_bn1d_bwd_apply_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int64, ctypes.c_int,
                                              c_f32p, ctypes.c_void_p]),
    "s2d_pointwise_conv_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int64, c_f32p, ctypes.c_void_p]),
    "s2d_comm_load_library": (ctypes.c_int, [ctypes.c_char_p]),
    "s2d_comm_available": (ctypes.c_int, []),
    "s2d_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "s2d_comm_init": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "s2d_comm_ranks": (ctypes.c_int, []),
    "s2d_comm_shutdown": (ctypes.c_int, []),
    "s2d_convt3d_mfma_supported": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "s2d_convt3d_mfma_packed_elems": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "s2d_convt3d_mfma_pack_weights": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_convt3d_mfma_fwd": (ctypes.c_int, [c_f32p, ctypes.c_void_p, c_f32p] + [ctypes.c_int] * 6 + [c_f32p, ctypes.c_void_p]),
    "s2d_convt3d_mfma_stats_tiles": (ctypes.c_int64, [ctypes.c_int] * 5),
    "s2d_convt3d_mfma_fwd_stats": (ctypes.c_int, [c_f32p, ctypes.c_void_p, c_f32p] + [ctypes.c_int] * 6 + [c_f32p, c_f32p, ctypes.c_void_p]),
    "s2d_convt3d_mfma_fwd_stats_y16": (ctypes.c_int, [c_f32p, ctypes.c_void_p, c_f32p] + [ctypes.c_int] * 6 + [c_f32p, c_f32p, ctypes.c_void_p]),
    "s2d_convt3d_mfma_d16_supported": (ctypes.c_int, [ctypes.c_int] * 5),
    "s2d_convt3d_mfma_dgrad_d16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 6 + [c_f32p, ctypes.c_void_p]),
    "s2d_convt3d_mfma_wgrad_d16": (ctypes.c_int, [c_f32p, ctypes.c_void_p] + [ctypes.c_int] * 6 + [c_f32p, ctypes.c_void_p, ctypes.c_size_t,
                                                                                                   ctypes.c_void_p]),
    "s2d_convt3d_mfma_norm_supported": (ctypes.c_int, [ctypes.c_int] * 5),
    "s2d_convt3d_mfma_fwd_stats_y16_norm": (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_void_p, c_f32p] + [ctypes.c_int] * 6 + [ctypes.c_void_p, c_f32p,
                                                                                                                  ctypes.c_void_p]),
    "s2d_convt3d_mfma_wgrad_d16_norm": (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_void_p] + [ctypes.c_int] * 6 + [c_f32p, ctypes.c_void_p, ctypes.c_size_t,
                                                                                                              ctypes.c_void_p]),
    "s2d_convt3d_mfma_dgrad": (ctypes.c_int, [c_f32p, ctypes.c_void_p] + [ctypes.c_int] * 6 + [c_f32p, ctypes.c_void_p]),
    "s2d_convt3d_mfma_wgrad_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 6),
    "s2d_convt3d_mfma_wgrad": (ctypes.c_int, [c_f32p, c_f32p] + [ctypes.c_int] * 6 + [c_f32p, ctypes.c_void_p, ctypes.c_size_t,
                                                                                     ctypes.c_void_p]),
    "s2d_pointwise_conv_wgrad_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "s2d_pointwise_conv_wgrad_f32": (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, c_f32p, c_f32p,
                                                    ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_pointwise_conv_wgrad_bf16_supported": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int64]),
    "s2d_pointwise_conv_wgrad_norm_bf16": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, c_f32p,
                                                          c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_pointwise_conv_wgrad_norm_x16": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, c_f32p,
                                                          c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_pointwise_conv_wgrad_bf16": (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, c_f32p, c_f32p,
                                                     ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_pcr_loss_workspace_bytes": (ctypes.c_size_t, []),
    "s2d_pcr_loss_fwd_f32": (ctypes.c_int, [c_f32p, c_f32p, c_i32p, c_f32p, ctypes.c_int64] + [ctypes.c_int] * 4 +
                             [c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_pcr_loss_bwd_f32": (ctypes.c_int, [c_f32p, c_f32p, c_i32p, c_f32p, ctypes.c_int64] + [ctypes.c_int] * 4 +
                             [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_void_p]),
    "s2d_nhwc_bf16_to_nchw_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, c_f32p, ctypes.c_void_p]),
    "s2d_nchw_f32_to_nhwc_bf16": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_nhwc_bf16_to_nchw_f32_ld": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, c_f32p, ctypes.c_void_p]),
    "s2d_nchw_f32_to_nhwc_bf16_ld": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_conv2d1x1_pack_weights_bf16": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_conv2d3x3_pack_weights_pair_bf16": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                            ctypes.c_void_p]),
    "s2d_conv2d1x1_pack_weights_pair_bf16": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_conv2d1x1_stats_tiles": (ctypes.c_int64, [ctypes.c_int] * 5),
    "s2d_conv2d1x1_nhwc_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_f32p, ctypes.c_void_p] + [ctypes.c_int] * 5 +
                                [ctypes.c_void_p, c_f32p, ctypes.c_void_p]),
    "s2d_conv2d1x1_wgrad_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 5),
    "s2d_conv2d1x1_wgrad_nhwc_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 +
                                      [c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_smallconv3x3_supported": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "s2d_smallconv3x3_fwd": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p] + [ctypes.c_int] * 5 + [c_f32p, ctypes.c_void_p]),
    "s2d_smallconv3x3_dgrad": (ctypes.c_int, [c_f32p, c_f32p] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_smallconv3x3_wgrad_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "s2d_smallconv3x3_wgrad": (ctypes.c_int, [ctypes.c_void_p, c_f32p] + [ctypes.c_int] * 5 +
                               [c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_conv2d2x2s2_pack_weights_bf16": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_conv2d2x2s2_stats_tiles": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "s2d_conv2d2x2s2_nhwc_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_f32p, ctypes.c_void_p] + [ctypes.c_int] * 5 +
                                  [ctypes.c_void_p, c_f32p, ctypes.c_void_p]),
    "s2d_convup_supported": (ctypes.c_int, [ctypes.c_int] * 3),
    "s2d_convup_pack_weights_bf16": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_convup_stats_tiles": (ctypes.c_int64, [ctypes.c_int] * 6),
    "s2d_convup_nhwc_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_f32p, ctypes.c_void_p] + [ctypes.c_int] * 6 +
                             [ctypes.c_void_p, c_f32p, ctypes.c_void_p]),
    "s2d_conv2d4x4s2_pack_weights_bf16": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_conv2d4x4s2_nhwc_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 +
                                  [ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_conv2d_s2_wgrad_supported": (ctypes.c_int, [ctypes.c_int] * 3),
    "s2d_conv2d_s2_wgrad_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 6),
    "s2d_conv2d_s2_wgrad_nhwc_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 6 +
                                      [c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_rows_wgrad_supported": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "s2d_rows_wgrad_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "s2d_rows_wgrad_f32": (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, c_f32p, ctypes.c_void_p, ctypes.c_size_t,
                                          ctypes.c_void_p]),
    "s2d_lnwide_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "s2d_lnwide_fwd_bf16": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, c_f32p,
                                           ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_lnwide_bwd_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, c_f32p,
                                           c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_focal_workspace_bytes": (ctypes.c_size_t, []),
    "s2d_focal_fwd": (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                     ctypes.c_int, c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_focal_bwd": (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                     ctypes.c_int, c_f32p, c_f32p, c_f32p, ctypes.c_void_p]),
    "s2d_regloss_fwd": (ctypes.c_int, [c_f32p, ctypes.c_void_p, ctypes.c_void_p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                       c_f32p, ctypes.c_void_p]),
    "s2d_regloss_bwd": (ctypes.c_int, [c_f32p, ctypes.c_void_p, ctypes.c_void_p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                       c_f32p, c_f32p, c_f32p, ctypes.c_void_p]),
    "s2d_masked_mse_workspace_bytes": (ctypes.c_size_t, []),
    "s2d_masked_mse_fwd": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_float,
                                          c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_masked_mse_bwd": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, c_f32p, c_f32p,
                                          ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_pcr_heads_supported": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int64]),
    "s2d_pcr_heads_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "s2d_pcr_heads_fwd_f32": (ctypes.c_int, [c_f32p, c_f32p, c_i32p, c_f32p, ctypes.c_int64] + [ctypes.c_int] * 5 +
                              [c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_pcr_heads_bwd_f32": (ctypes.c_int, [c_f32p, c_f32p, c_i32p, c_f32p, ctypes.c_int64] + [ctypes.c_int] * 5 +
                              [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int] + [c_f32p] * 5 +
                              [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_pcr_level_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "s2d_pcr_level_fwd_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, ctypes.c_int64] + [ctypes.c_int] * 6 +
                              [c_f32p, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_pcr_level_fwd_y16": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, ctypes.c_int64] + [ctypes.c_int] * 6 +
                              [c_f32p, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_pcr_level_bwd_sums_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, ctypes.c_int64] + [ctypes.c_int] * 5 +
                                   [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_size_t,
                                    ctypes.c_void_p]),
    "s2d_pcr_level_bwd_sums_y16": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, ctypes.c_int64] + [ctypes.c_int] * 5 +
                                   [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_size_t,
                                    ctypes.c_void_p]),
    "s2d_pcr_level_bwd_apply_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, ctypes.c_int64] + [ctypes.c_int] * 5 +
                                    [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int, c_f32p, c_f32p, ctypes.c_void_p]),
    "s2d_pcr_level_bwd_apply_y16": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, ctypes.c_int64] + [ctypes.c_int] * 5 +
                                    [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int, c_f32p, c_f32p, ctypes.c_void_p]),
    "s2d_pcr_level_bwd_apply_y16_d16": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, ctypes.c_int64] + [ctypes.c_int] * 5 +
                                    [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int, c_f32p, c_f32p, ctypes.c_void_p]),
    "s2d_bev_iou_f32": (ctypes.c_int, [c_f32p, ctypes.c_int, c_f32p, ctypes.c_int, c_f32p, ctypes.c_void_p]),
    "s2d_nms_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "s2d_nms_rotated_bev": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_nms_circle": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_assign_label": (ctypes.c_int, [c_f32p, c_i32p, ctypes.c_int, ctypes.c_int, ctypes.c_float * 2, ctypes.c_float * 2] + [ctypes.c_int] * 5 +
                         [ctypes.c_double, ctypes.c_int] + [ctypes.c_void_p] * 7),
    "s2d_adam_max_tensors": (ctypes.c_int, []),
    "s2d_adam_step_f32": (ctypes.c_int, [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_float] * 5 + [ctypes.c_int, c_f32p, ctypes.c_void_p]),
    "s2d_grad_norm_workspace_floats": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_void_p]),
    "s2d_grad_sumsq_f32": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, c_f32p, ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_grad_norm_finalize_f32": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_float, c_f32p, ctypes.c_void_p]),
    "s2d_dwconv7_supported": (ctypes.c_int, [ctypes.c_int]),
    "s2d_dwconv7_nhwc_bf16": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_dwconv7_wgrad_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "s2d_dwconv7_wgrad_nhwc_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 +
                                    [c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_comm_allreduce_sum_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "s2d_conv2d3x3_supported": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "s2d_conv2d3x3_stats_tiles": (ctypes.c_int64, [ctypes.c_int] * 7),
    "s2d_conv2d3x3_pack_weights_bf16": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                       ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_conv2d3x3_tile_rows": (ctypes.c_int, [ctypes.c_int] * 7),
    "s2d_conv2d3x3_nhwc_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, c_f32p, ctypes.c_void_p] +
                                [ctypes.c_int] * 7 + [ctypes.c_void_p, c_f32p, ctypes.c_void_p]),
    "s2d_bn_partials_finalize_f32": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, c_f32p, c_f32p,
                                                    ctypes.c_float, ctypes.c_float, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                                    ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_bn_partials_finalize_ws_f32": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, c_f32p, c_f32p,
                                                       ctypes.c_float, ctypes.c_float, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_conv2d3x3_bnbwd_supported": (ctypes.c_int, [ctypes.c_int] * 4),
    "s2d_conv2d3x3_nhwc_bf16_bnbwd": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_void_p, c_f32p, c_f32p,
                                                                                                    ctypes.c_int, c_f32p, ctypes.c_void_p]),
    "s2d_conv2d1x1_bnbwd_supported": (ctypes.c_int, [ctypes.c_int] * 2),
    "s2d_conv2d1x1_nhwc_bf16_bnbwd": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p, c_f32p, c_f32p,
                                                                                                    ctypes.c_int, c_f32p, ctypes.c_void_p]),
    "s2d_bn_partials_bwd_finalize_ws_f32": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, c_f32p, c_f32p, c_f32p, c_f32p,
                                                           c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_bn_partials_sum_f32": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, c_f32p, ctypes.c_int,
                                               ctypes.c_void_p]),
    "s2d_bn_partials_sum_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "s2d_bn_partials_sum_ws_f32": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, c_f32p, ctypes.c_int,
                                                  ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_bnrow_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int]),
    "s2d_bnrow_stats_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, c_f32p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_bnrow_stats_finalize_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, c_f32p, c_f32p, ctypes.c_float, ctypes.c_float, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_bnrow_apply_bf16": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_bnrow_bwd_reduce_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_bnrow_bwd_reduce_finalize_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_bnrow_bwd_apply_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_int, c_f32p, c_f32p, c_f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_bnrow_apply_ld_bf16": (ctypes.c_int, [ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p,
                                               ctypes.c_int, ctypes.c_void_p]),
    "s2d_bnrow_bwd_reduce_ld_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int64,
                                                    ctypes.c_int, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_bnrow_bwd_reduce_finalize_ld_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_int,
                                                             ctypes.c_int64, ctypes.c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                                             ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_bnrow_bwd_apply_ld_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_int, c_f32p, c_f32p,
                                                   c_f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_upsample2x_nhwc_bf16": (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_upsample2x_bwd_nhwc_bf16": (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_maxpool2x2_nhwc_bf16": (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_maxpool2x2_bwd_nhwc_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_spconv_s16_supported": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "s2d_debug_rg_trace": (None, [ctypes.c_void_p]),
    "s2d_spconv_s16_packed_elems": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "s2d_spconv_s16_pack_weights": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_spconv_s16_pack_weights_pair": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                                        ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_spconv_s16_fwd": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, c_f32p, c_i32p, ctypes.c_int64,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p]),
    "s2d_spconv_s16_stats_tiles": (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "s2d_spconv_s16_fwd_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, c_f32p, c_i32p, ctypes.c_int64,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, c_f32p,
                                                ctypes.c_void_p]),
    "s2d_rulebook_sort_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64]),
    "s2d_rulebook_sort_chunk_rows": (ctypes.c_int64, [ctypes.c_int64]),
    "s2d_rulebook_sort_by_mask": (ctypes.c_int, [c_i32p, ctypes.c_int, ctypes.c_int64, c_i32p, ctypes.c_void_p, c_i32p, ctypes.c_void_p, ctypes.c_size_t,
                                                 ctypes.c_void_p]),
    "s2d_spconv_s16_set_sorted_rows": (ctypes.c_int, [ctypes.c_int]),
    "s2d_spconv_s16_sorted_supported": (ctypes.c_int, [ctypes.c_int] * 3),
    "s2d_spconv_s16_fwd_sorted": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, c_f32p, c_i32p, c_i32p, ctypes.c_void_p, ctypes.c_int64,
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, c_f32p, ctypes.c_void_p]),
    "s2d_conv2d3x3_wgrad_supported": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "s2d_conv2d3x3_wgrad_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 6),
    "s2d_conv2d3x3_wgrad_nhwc_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 6 +
                                      [c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_convt3d_k4s2p1_fwd_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, ctypes.c_void_p]),
    "s2d_convt3d_k4s2p1_dgrad_f32": (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p, ctypes.c_void_p]),
    "s2d_convt3d_k4s2p1_wgrad_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 6),
    "s2d_convt3d_k4s2p1_wgrad_f32": (ctypes.c_int, [c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                    ctypes.c_int, ctypes.c_int, c_f32p, ctypes.c_void_p, ctypes.c_size_t,
                                                    ctypes.c_void_p]),
    "s2d_bncm_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int64]),
    "s2d_bncm_stats_f32": (ctypes.c_int, [c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, c_f32p, ctypes.c_void_p,
                                          ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_bncm_apply_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int64, c_f32p, ctypes.c_void_p]),
    "s2d_bncm_bwd_reduce_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int64, c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_bncm_bwd_reduce_x_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, c_f32p, ctypes.c_void_p,
                                                 ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_bncm_bwd_apply_x_f32": (ctypes.c_int, [c_f32p] * 7 + [ctypes.c_int, ctypes.c_int, ctypes.c_int64, c_f32p, ctypes.c_void_p]),
    # r06: bf16-stored z / dz / dx' of the PCR head (storage flags: 0 fp32, 1 bf16)
    "s2d_bncm_bwd_reduce_x_typed": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int64, c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_bncm_bwd_apply_x_typed": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "s2d_convt3d_mfma_x16_supported": (ctypes.c_int, [ctypes.c_int] * 5),
    "s2d_convt3d_mfma_dgrad_d16_x16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_convt3d_mfma_fwd_stats_y16_norm_x16": (ctypes.c_int, [ctypes.c_void_p, c_f32p, ctypes.c_void_p, c_f32p] + [ctypes.c_int] * 6 +
                                                [ctypes.c_void_p, c_f32p, ctypes.c_void_p]),
    "s2d_convt3d_mfma_wgrad_d16_norm_x16": (ctypes.c_int, [ctypes.c_void_p, c_f32p, ctypes.c_void_p] + [ctypes.c_int] * 6 +
                                            [c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_pcr_level_site_cache": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]),
    "s2d_pcr_level_fwd_y16_z16": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, ctypes.c_int64] + [ctypes.c_int] * 6 +
                                  [ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_pcr_level_bwd_sums_y16_z16": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, ctypes.c_int64] + [ctypes.c_int] * 5 +
                                       [c_f32p, c_f32p, c_f32p, ctypes.c_void_p, c_f32p, ctypes.c_int, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.c_void_p]),
    "s2d_pcr_level_bwd_apply_y16_d16_z16": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, ctypes.c_int64] + [ctypes.c_int] * 5 +
                                            [c_f32p, c_f32p, c_f32p, ctypes.c_void_p, c_f32p, ctypes.c_int, c_f32p, ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_pointwise_conv_wgrad_norm_x16_d16": (ctypes.c_int, [ctypes.c_void_p, c_f32p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                              ctypes.c_int64, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_space_depth2_nhwc_bf16": (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_bncm_bwd_apply_f32": (ctypes.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int64, c_f32p, ctypes.c_void_p]),
    "s2d_densify_bev_fwd_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_i32p, ctypes.c_int64, ctypes.c_int, _I3,
                                                ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "s2d_densify_bev_bwd_bf16": (ctypes.c_int, [ctypes.c_void_p, c_i32p, ctypes.c_int64, ctypes.c_int, _I3, ctypes.c_int,
                                                ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "s2d_spconv_s16_wgrad": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, c_i32p, ctypes.c_int64, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, c_f32p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "s2d_densify_fwd_f32": (ctypes.c_int, [c_f32p, c_i32p, ctypes.c_int64, ctypes.c_int, _I3, ctypes.c_int, c_f32p,
                                           ctypes.c_void_p]),
    "s2d_densify_bwd_f32": (ctypes.c_int, [c_f32p, c_i32p, ctypes.c_int64, ctypes.c_int, _I3, ctypes.c_int, c_f32p,
                                           ctypes.c_void_p]),
}

_lib = None


class S2DError(RuntimeError):
    pass


def _memoised(fn):
    cache = {}

    def query(*a):
        r = cache.get(a)
        if r is None:
            r = cache[a] = fn(*a)
        return r
    query.__wrapped__ = fn
    return query


def load():
    """dlopen libs2d_hip.so and type every entry point.  Raises if the build is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise S2DError(
            f"{LIB_PATH} not found: the HIP extension is not built (python -m sparse2dense_amd.build). "
            "There is no CPU fallback for the sparse2dense_amd ops.")
    # torch first: PyTorch-ROCm ships its own libamdhip64; the process must hold ONE HIP runtime (the one that owns the
    # tensors' memory and streams), and the loader binds us to whichever copy of that SONAME is already mapped.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise S2DError(f"{LIB_PATH} does not export {name}")
        fn.restype = res
        fn.argtypes = args
    # Pure shape queries (workspace sizes, launch plans, *_supported) are asked again with the same integers by every layer call of every step
    # - ~150 ctypes round trips of 3-5 us per training step on the launch thread.  They depend on their integer arguments and on
    # environment switches the library reads once, so the answers are memoised per argument tuple.  (Not the queries that follow a
    # run-time mode: the sparse bf16-storage kernels' plans change with the sorted-row switch.)
    _int_types = (ctypes.c_int, ctypes.c_int64, ctypes.c_size_t)
    for name, (res, args) in SIGNATURES.items():
        pure = name.endswith(("_workspace_bytes", "_supported", "_stats_tiles", "_tile_rows", "_packed_elems")) and "sort" not in name and "spconv_s16" not in name
        if pure and args and all(a in _int_types for a in args):
            setattr(lib, name, _memoised(getattr(lib, name)))
    _lib = lib
    return lib


def last_error():
    buf = ctypes.create_string_buffer(512)
    load().s2d_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def build_info():
    """compiler / HIP headers / build date of the loaded library (s2d_build_info)"""
    buf = ctypes.create_string_buffer(256)
    load().s2d_build_info(buf, 256)
    return buf.value.decode(errors="replace")


def check(rc, what=""):
    """Mirror of the reference's AT_ERROR/TORCH_CHECK convention: non-zero -> RuntimeError."""
    if rc != 0:
        raise S2DError(f"{what} failed (rc={rc}): {last_error()}")


def i3(v):
    if isinstance(v, int):
        v = (v, v, v)
    v = [int(x) for x in v]
    assert len(v) == 3
    return _I3(*v)


def f3(v):
    return _F3(*[float(x) for x in v])


def f6(v):
    return _F6(*[float(x) for x in v])
