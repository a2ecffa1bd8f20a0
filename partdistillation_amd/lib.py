"""ctypes binding of libpd_hip.so — the C-ABI declared in include/*.h.

The HIP library is the product: there is NO fallback.  If the shared object is
missing or a symbol is absent this module raises, and every op built on it
raises with it (a GPU box must never silently run something else).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PD_LIB_PATH") or os.path.join(_HERE, "libpd_hip.so")   # PD_LIB_PATH: development A/B of two builds (tools/ab_bench.sh)
CSRC = os.path.join(_HERE, "csrc")

PD_F32, PD_F64, PD_BF16 = 0, 1, 2
ABI_VERSION = 39

_c_int, _c_vp = ctypes.c_int, ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol include/*.h declares
SIGNATURES = {
    "pd_msda_forward": (_c_int, [_c_vp] * 6 + [_c_int] * 9 + [_c_vp]),
    "pd_msda_backward": (_c_int, [_c_vp] * 9 + [_c_int] * 9 + [_c_vp]),
    "pd_msda_backward_last_gate": (_c_int, [_c_vp]),
    "pd_stem7x7_fwd": (_c_int, [_c_vp, _c_int] + [_c_vp] * 4 + [_c_int] * 4 + [_c_vp]),
    "pd_stem_wgrad_workspace_floats": (ctypes.c_int64, []),
    "pd_stem7x7_wgrad": (_c_int, [_c_vp, _c_int] + [_c_vp] * 4 + [_c_int, _c_vp] + [_c_int] * 4 + [_c_vp]),
    "pd_msda_fused_supported": (_c_int, [_c_int] * 7),
    "pd_msda_fused_forward": (_c_int, [_c_vp] * 4 + [_c_int] + [_c_vp] * 4 + [_c_int] * 7 + [_c_vp]),
    "pd_msda_fused_backward": (_c_int, [_c_vp] * 4 + [_c_int] + [_c_vp] * 6 + [_c_int, _c_vp, _c_vp] + [_c_int] * 7 + [_c_vp]),
    "pd_lsa_batched": (_c_int, [_c_vp] * 4 + [_c_int] * 3 + [_c_vp]),
    "pd_sumsq_accumulate": (_c_int, [_c_vp, ctypes.c_int64, _c_int, _c_vp, _c_vp]),
    "pd_adamw_clipped": (_c_int, [_c_vp] * 4 + [ctypes.c_int64, _c_int] + [ctypes.c_double] * 5 + [_c_int, _c_vp, ctypes.c_double, _c_vp, _c_vp]),
    "pd_gemm_tn_f32": (_c_int, [_c_vp] * 4 + [_c_int] * 7 + [_c_vp]),
    "pd_gemm_tn_f32x3": (_c_int, [_c_vp] * 4 + [_c_int] * 7 + [_c_vp]),
    "pd_gemm_wgrad_f16x2_takes_wide_tiles": (_c_int, [_c_int] * 2),
    "pd_gemm_wgrad_f16x2_ws_floats": (ctypes.c_int64, [_c_int] * 2),
    "pd_gemm_wgrad_f16x2_grouped_ws_floats": (ctypes.c_int64, [_c_vp, _c_int]),
    "pd_gemm_wgrad_acc_f16x2_ws": (_c_int, [_c_vp] * 7 + [ctypes.c_int64] + [_c_int] * 6 + [_c_vp]),
    "pd_gemm_wgrad_f16x2_grouped": (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, ctypes.c_int64, _c_vp]),
    "pd_conv3x3_wgrad_nhwc_f16x2": (_c_int, [_c_vp] * 7 + [ctypes.c_int64] + [_c_int] * 5 + [_c_vp]),
    "pd_conv3x3_nhwc_f16x2": (_c_int, [_c_vp] * 7 + [_c_int] * 5 + [_c_vp]),
    "pd_gemm_tn_f16x2_which": (_c_int, [_c_int] * 6),
    "pd_gemm_tn_f16x2_bits_words": (ctypes.c_int64, [_c_int] * 2),
    "pd_gemm_tn_f16x2": (_c_int, [_c_vp] * 9 + [_c_int] * 7 + [_c_vp]),
    "pd_gemm_tn_f16x2_bf16out": (_c_int, [_c_vp] * 6 + [_c_int] * 6 + [_c_vp]),
    "pd_cast_bf16_f32_amax": (_c_int, [_c_vp, _c_int, _c_int, _c_vp, _c_vp, _c_vp]),
    "pd_row_amax_f32": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp]),
    "pd_gemm_tn_f32x3_relumask": (_c_int, [_c_vp] * 5 + [_c_int] * 6 + [_c_vp]),
    "pd_gemm_tn_f32x3_relu_bits": (_c_int, [_c_vp] * 5 + [_c_int] * 6 + [_c_vp]),
    "pd_gemm_tn_f32x3_relu_bits_words": (ctypes.c_int64, [_c_int] * 2),
    "pd_split3_bf16": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp, _c_vp]),
    "pd_gemm_tn_f32x3_pre": (_c_int, [_c_vp] * 6 + [_c_int] * 6 + [_c_vp]),
    "pd_gemm_wgrad_f32": (_c_int, [_c_vp] * 4 + [_c_int] * 6 + [_c_vp]),
    "pd_gemm_wgrad_acc_f32": (_c_int, [_c_vp] * 4 + [_c_int] * 6 + [_c_vp]),
    "pd_conv3x3_nhwc_f32x3": (_c_int, [_c_vp] * 4 + [_c_int] * 5 + [_c_vp]),
    "pd_gemm_wgrad_acc_f32x3": (_c_int, [_c_vp] * 4 + [_c_int] * 6 + [_c_vp]),
    "pd_conv3x3_wgrad_nhwc_f32x3": (_c_int, [_c_vp] * 5 + [ctypes.c_int64] + [_c_int] * 5 + [_c_vp]),
    "pd_gemm_wgrad_f32x3_ws_floats": (ctypes.c_int64, [_c_int] * 2),
    "pd_gemm_wgrad_f32x3_grouped_table_bytes": (ctypes.c_int64, [_c_int]),
    "pd_gemm_wgrad_f32x3_grouped_ws_floats": (ctypes.c_int64, [_c_vp, _c_int]),
    "pd_gemm_wgrad_f32x3_grouped": (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, ctypes.c_int64, _c_vp]),
    "pd_gemm_wgrad_acc_f32x3_ws": (_c_int, [_c_vp] * 5 + [ctypes.c_int64] + [_c_int] * 6 + [_c_vp]),
    "pd_adamw_clipped_shadow": (_c_int, [_c_vp] * 5 + [ctypes.c_int64] + [ctypes.c_double] * 5 + [_c_int, _c_vp, ctypes.c_double, _c_vp, _c_vp]),
    "pd_conv_bf16_supported": (_c_int, [_c_int] * 5),
    "pd_conv_bf16_fwd": (_c_int, [_c_vp] * 6 + [_c_int] * 11 + [_c_vp]),
    "pd_conv_bf16_wgrad_workspace_floats": (ctypes.c_int64, [_c_int] * 6),
    "pd_conv_bf16_wgrad": (_c_int, [_c_vp] * 4 + [ctypes.c_int64] + [_c_int] * 10 + [_c_vp]),
    "pd_conv_bf16_wgrad_grouped_table_bytes": (ctypes.c_int64, [_c_int]),
    "pd_conv_bf16_wgrad_grouped_workspace_floats": (ctypes.c_int64, [_c_vp, _c_int]),
    "pd_conv_bf16_wgrad_grouped": (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, ctypes.c_int64, _c_vp]),
    "pd_conv_bf16_dgrad": (_c_int, [_c_vp] * 4 + [_c_int] * 10 + [_c_vp]),
    "pd_maxpool3s2_fwd_bf16": (_c_int, [_c_vp] * 3 + [_c_int] * 4 + [_c_vp]),
    "pd_maxpool3s2_bwd_bf16": (_c_int, [_c_vp] * 3 + [_c_int] * 4 + [_c_vp]),
    "pd_affine_act_fwd_bf16": (_c_int, [_c_vp] * 5 + [ctypes.c_int64, _c_int, _c_int, _c_vp]),
    "pd_affine_act_bwd_bf16": (_c_int, [_c_vp] * 5 + [ctypes.c_int64, _c_int, _c_int, _c_vp]),
    "pd_affine_act_bwd2_bf16": (_c_int, [_c_vp] * 6 + [ctypes.c_int64, _c_int, _c_int, _c_vp]),
    "pd_multi_gather_sumsq": (_c_int, [_c_vp] * 8 + [_c_int, _c_int, _c_vp]),
    "pd_attn_workspace_floats": (ctypes.c_int64, [_c_int] * 4),
    "pd_attn_fwd_d32": (_c_int, [_c_vp] * 7 + [_c_int] * 4 + [ctypes.c_float, _c_int, _c_vp]),
    "pd_attn_bwd_d32": (_c_int, [_c_vp] * 11 + [_c_int] * 4 + [ctypes.c_float, _c_int, _c_vp]),
    "pd_attn_fwd_d32_ld": (_c_int, [_c_vp] * 7 + [_c_int] * 4 + [ctypes.c_float, _c_int, _c_int, _c_vp]),
    "pd_attn_bwd_d32_ld": (_c_int, [_c_vp] * 11 + [_c_int] * 4 + [ctypes.c_float, _c_int, _c_int, _c_vp]),
    "pd_nc_sums_f32": (_c_int, [_c_vp] * 6 + [_c_int] * 5 + [_c_vp]),
    "pd_nc_affine_f32": (_c_int, [_c_vp] * 4 + [_c_int] * 4 + [_c_vp]),
    "pd_nc_affine_amax_f32": (_c_int, [_c_vp] * 5 + [_c_int] * 4 + [_c_vp]),
    "pd_nc_affine2_amax_f32": (_c_int, [_c_vp] * 8 + [_c_int] * 4 + [_c_vp]),
    "pd_upsample_add_amax_nhwc_f32": (_c_int, [_c_vp, ctypes.c_int64] + [_c_vp] * 3 + [_c_int] * 6 + [_c_vp]),
    "pd_nc_affine2_f32": (_c_int, [_c_vp] * 7 + [_c_int] * 4 + [_c_vp]),
    "pd_add_layernorm_fwd": (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, ctypes.c_float, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_vp,
                                      _c_int, _c_vp, _c_vp, _c_int, _c_int, _c_vp]),
    "pd_add_layernorm_bwd": (_c_int, [_c_vp] * 4 + [_c_int] + [_c_vp] * 6 + [_c_int] + [_c_vp] * 4 + [_c_int] * 3 + [_c_vp]),
    "pd_add_layernorm_fwd_amax": (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp, ctypes.c_float, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_vp,
                                           _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_vp]),
    "pd_add_layernorm_bwd_amax": (_c_int, [_c_vp] * 4 + [_c_int] + [_c_vp] * 6 + [_c_int] + [_c_vp] * 4 + [_c_int] + [_c_vp] + [_c_int] * 2 + [_c_vp]),
    "pd_colsum_acc": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp]),
    "pd_relu_bwd_colsum": (_c_int, [_c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp]),
    "pd_mem_prep_fwd": (_c_int, [_c_vp, ctypes.c_int64, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp]),
    "pd_mem_prep_bwd": (_c_int, [_c_vp, _c_vp, _c_int, _c_vp, ctypes.c_int64, _c_int, _c_int, _c_int, _c_vp]),
    "pd_attn_mask_u8": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp]),
    "pd_add_rows_amax_f32": (_c_int, [_c_vp] * 6 + [_c_int, _c_int, _c_vp]),
    "pd_sum3_sum2_f32": (_c_int, [_c_vp] * 6 + [ctypes.c_int64, _c_vp]),
    "pd_transpose_batched_f32": (_c_int, [_c_vp, _c_int, _c_vp]),
    "pd_normalize_u8_nhwc": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp]),
    "pd_resize_bilinear_nhwc_f32": (_c_int, [_c_vp] + [_c_int] * 4 + [_c_vp] * 3 + [_c_int, _c_int, _c_vp]),
    "pd_matcher_point_terms": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp]),
    "pd_match_point_logits": (_c_int, [_c_vp] * 4 + [_c_int] * 7 + [_c_vp]),
    "pd_matcher_costs": (_c_int, [_c_vp, _c_int, _c_vp] + [ctypes.c_int64] * 3 + [_c_vp] * 3 + [_c_int] * 6 + [ctypes.c_float] * 3 + [_c_vp]),
    "pd_mask_point_losses_fwd": (_c_int, [_c_vp] * 5 + [_c_int, _c_int, _c_vp]),
    "pd_mask_point_losses_bwd": (_c_int, [_c_vp] * 6 + [_c_int, _c_int, _c_vp]),
    "pd_uncertain_points": (_c_int, [_c_vp] * 4 + [_c_int] * 4 + [_c_vp]),
    "pd_point_sample_u8": (_c_int, [_c_vp] * 4 + [_c_int] * 5 + [_c_vp]),
    "pd_skinny_linear_fwd": (_c_int, [_c_vp] * 3 + [_c_int, _c_vp] + [_c_int] * 3 + [_c_vp]),
    "pd_skinny_linear_partial_floats": (ctypes.c_int64, [_c_int] * 3),
    "pd_skinny_linear_bwd": (_c_int, [_c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_vp] + [_c_int] * 4 + [_c_vp]),
    "pd_loss_vectors_fwd": (_c_int, [_c_vp, ctypes.c_int64, ctypes.c_int64] + [_c_vp] * 9 + [_c_int] * 5 + [_c_vp]),
    "pd_loss_vectors_bwd": (_c_int, [_c_vp, ctypes.c_int64, ctypes.c_int64] + [_c_vp] * 10 + [_c_int] * 5 + [_c_vp]),
    "pd_pair_logits_fwd": (_c_int, [_c_vp] * 5 + [_c_int] * 4 + [_c_vp]),
    "pd_pair_logits_bwd_tok": (_c_int, [_c_vp] * 5 + [_c_int] * 4 + [_c_vp]),
    "pd_pair_logits_workspace_floats": (ctypes.c_int64, [_c_int] * 3),
    "pd_pair_logits_bwd_rows": (_c_int, [_c_vp] * 6 + [_c_int] * 4 + [_c_vp]),
    "pd_msda_prep_fwd": (_c_int, [_c_vp] * 6 + [ctypes.c_int64] + [_c_int] * 5 + [_c_vp]),
    "pd_msda_prep_bwd_amax": (_c_int, [_c_vp] * 7 + [ctypes.c_int64] + [_c_int] * 5 + [_c_vp]),
    "pd_msda_forward_amax": (_c_int, [_c_vp] * 7 + [_c_int] * 9 + [_c_vp]),
    "pd_msda_prep_bwd": (_c_int, [_c_vp] * 6 + [ctypes.c_int64] + [_c_int] * 5 + [_c_vp]),
    "pd_sgemm_tn_batched_bf16": (_c_int, [_c_vp] * 3 + [_c_int] * 7 + [ctypes.c_int64] * 3 + [_c_vp]),
    "pd_sgemm_tn_multi_bf16": (_c_int, [_c_vp, _c_int, _c_int, _c_vp]),
    "pd_decoder_head_bf16": (_c_int, [_c_vp] * 3 + [ctypes.c_float] + [_c_vp] * 10 + [_c_int] * 4 + [_c_vp]),
    # include/pd_declayer.h
    "pd_dec_fwd_a": (_c_int, [_c_vp] * 3 + [_c_int] + [_c_vp] * 4 + [ctypes.c_float] + [_c_vp] * 10 + [_c_int] + [_c_vp]),
    "pd_dec_fwd_b": (_c_int, [_c_vp] * 3 + [_c_int] + [_c_vp] * 20 + [ctypes.c_float] + [_c_vp] * 13 + [_c_int] * 2 + [_c_vp]),
    "pd_dec_bwd_b": (_c_int, [_c_vp] * 14 + [_c_int] + [_c_vp] * 15 + [_c_int] + [_c_vp]),
    "pd_dec_split": (_c_int, []),
    "pd_dec_workspace_bytes": (ctypes.c_int64, [_c_int]),
    "pd_dec_pack_table_bytes": (ctypes.c_int64, [_c_int]),
    "pd_dec_pack_grouped": (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp]),
    "pd_dec_bwd_a": (_c_int, [_c_vp] * 11 + [_c_int] + [_c_vp] * 4 + [_c_int] + [_c_vp]),
    "pd_sgemm_split_workspace_floats": (ctypes.c_int64, [_c_int] * 3),
    "pd_sgemm_split_tickets": (ctypes.c_int64, [_c_int] * 2),
    "pd_sgemm_tn_splitk_bf16": (_c_int, [_c_vp] * 5 + [ctypes.c_int64, _c_vp] + [_c_int] * 7 + [_c_vp]),
    "pd_sgemm_nn_splitn_bf16": (_c_int, [_c_vp] * 5 + [ctypes.c_int64, _c_vp] + [_c_int] * 7 + [_c_vp]),
    "pd_sgemm_tn_bf16": (_c_int, [_c_vp] * 4 + [_c_int] * 7 + [_c_vp]),
    "pd_sgemm_nn_bf16": (_c_int, [_c_vp] * 4 + [_c_int] * 7 + [_c_vp]),
    "pd_sgemm_wgrad_bf16": (_c_int, [_c_vp] * 4 + [_c_int] * 6 + [_c_vp]),
    "pd_gn_coeffs_fwd": (_c_int, [_c_vp] * 3 + [_c_int] * 4 + [ctypes.c_float] + [_c_vp] * 6),
    "pd_gn_coeffs_bwd": (_c_int, [_c_vp] * 4 + [_c_int] * 4 + [_c_vp] * 6),
    "pd_point_sample_nhwc_f32": (_c_int, [_c_vp] * 3 + [_c_int] * 5 + [_c_vp]),
    "pd_point_sample_nhwc_f32_bf16": (_c_int, [_c_vp] * 3 + [_c_int] * 5 + [_c_vp]),
    "pd_point_sample_planar_f32": (_c_int, [_c_vp] * 3 + [_c_int] * 5 + [_c_vp]),
    "pd_point_sample_planar_bwd_needs_zero": (_c_int, [_c_int] * 3),
    "pd_point_sample_planar_bwd_needs_zero_n": (_c_int, [_c_int] * 4),
    "pd_point_sample_planar_bwd_f32": (_c_int, [_c_vp] * 3 + [_c_int] * 5 + [_c_vp]),
    "pd_upsample_add_nhwc_f32": (_c_int, [_c_vp, ctypes.c_int64] + [_c_vp] * 2 + [_c_int] * 6 + [_c_vp]),
    "pd_upsample2x_bwd_nhwc_f32": (_c_int, [_c_vp] * 2 + [_c_int] * 4 + [_c_vp]),
    "pd_scores_argmax_u8": (_c_int, [_c_vp] * 3 + [_c_int] * 7 + [_c_vp]),
    "pd_kmeans_assign": (_c_int, [_c_vp, _c_vp, _c_int] + [_c_vp] * 7 + [_c_int, _c_int, _c_vp]),
    "pd_kmeans_assign_partial": (_c_int, [_c_vp, _c_vp, _c_int] + [_c_vp] * 7 + [_c_int, _c_int, _c_vp]),
    "pd_kmeans_reduce": (_c_int, [_c_vp] * 6 + [_c_int] * 3 + [_c_vp]),
    "pd_kmeans_reduce_update_scratch_floats": (ctypes.c_int64, [_c_int] * 3),
    "pd_kmeans_reduce_update": (_c_int, [_c_vp] * 11 + [_c_int] * 3 + [_c_vp]),
    "pd_kmeans_assign_bounded": (_c_int, [_c_vp, _c_vp, _c_int] + [_c_vp] * 11 + [_c_int, _c_int, _c_vp]),
    "pd_kmeans_reduce_update_shift": (_c_int, [_c_vp] * 12 + [_c_int] * 3 + [_c_vp]),
    "pd_kmeans_update": (_c_int, [_c_vp] * 8 + [_c_int] * 3 + [_c_vp]),
    "pd_mask_assign": (_c_int, [_c_vp] * 6 + [_c_int] * 7 + [_c_vp]),
    "pd_window_attn_fwd_w12": (_c_int, [_c_vp] * 6 + [_c_int] * 3 + [ctypes.c_float, _c_vp, _c_vp, _c_int, _c_vp]),
    "pd_window_attn_bwd_w12": (_c_int, [_c_vp] * 9 + [_c_int] * 3 + [ctypes.c_float, _c_vp, _c_vp, _c_int, _c_vp]),
    "pd_layernorm_rows_f32_fwd": (_c_int, [_c_vp] * 3 + [ctypes.c_float] + [_c_vp] * 3 + [ctypes.c_int64, _c_int, _c_vp]),
    "pd_layernorm_rows_f32_bwd": (_c_int, [_c_vp] * 8 + [ctypes.c_int64, _c_int, _c_vp]),
    "pd_swin_tail_ln_fwd": (_c_int, [_c_vp] * 3 + [_c_int] + [_c_vp] * 2 + [ctypes.c_float] + [_c_vp] * 4 + [ctypes.c_int64, _c_int, _c_vp]),
    "pd_swin_tail_ln_bwd": (_c_int, [_c_vp] * 7 + [_c_int] + [_c_vp] * 4 + [ctypes.c_int64, _c_int, _c_vp]),
    "pd_swin_merge_ln_fwd": (_c_int, [_c_vp] * 3 + [ctypes.c_float] + [_c_vp] * 3 + [_c_int] * 4 + [_c_vp]),
    "pd_swin_merge_ln_bwd": (_c_int, [_c_vp] * 8 + [_c_int] * 4 + [_c_vp]),
    "pd_swin_ln_fwd": (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_vp, ctypes.c_float, _c_vp, _c_vp, _c_vp, _c_int, _c_vp,
                                _c_int, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_int, _c_vp]),
    "pd_swin_ln_bwd": (_c_int, [_c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp,
                                _c_int, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_int, _c_int, ctypes.c_int64, _c_vp]),
    "pd_resample_rows_u8": (_c_int, [_c_vp] + [_c_int] * 6 + [_c_vp] * 3 + [_c_int, _c_int, _c_vp, _c_vp]),
    "pd_resample_cols_u8": (_c_int, [_c_vp] + [_c_int] * 3 + [_c_vp] * 3 + [_c_int] * 5 + [_c_vp, _c_vp]),
    "pd_rle_sample_u8": (_c_int, [_c_vp, _c_vp] + [_c_int] * 4 + [_c_vp, _c_vp] + [_c_int] * 3 + [_c_vp, _c_vp, _c_vp]),
    "pd_sgemm_wgrad_grouped_table_bytes": (ctypes.c_int64, [_c_int]),
    "pd_sgemm_wgrad_grouped_bf16": (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp]),
    "pd_sgemm_wgrad_split_workspace": (ctypes.c_int64, [_c_int] * 3),
    "pd_sgemm_wgrad_split_bf16": (_c_int, [_c_vp] * 5 + [_c_int] * 6 + [_c_vp]),
    "pd_igemm_bf16_supported": (_c_int, [_c_int] * 5),
    "pd_igemm_bf16_workspace_bytes": (ctypes.c_int64, [_c_vp]),
    "pd_igemm_bf16": (_c_int, [_c_vp, _c_vp, ctypes.c_int64, _c_vp]),
    "pd_igemm_bf16_time": (_c_int, [_c_vp, _c_vp, ctypes.c_int64, _c_int, _c_vp, _c_vp]),
    "pd_igemm_bf16_seq_workspace_bytes": (ctypes.c_int64, [_c_vp, _c_int]),
    "pd_igemm_bf16_seq": (_c_int, [_c_vp, _c_int, _c_vp, ctypes.c_int64, _c_vp]),
    "pd_wgrad_bf16_workspace_bytes": (ctypes.c_int64, [_c_vp]),
    "pd_wgrad_bf16": (_c_int, [_c_vp, _c_vp, ctypes.c_int64, _c_vp]),
    "pd_wgrad_bf16_seq_workspace_bytes": (ctypes.c_int64, [_c_vp, _c_int]),
    "pd_wgrad_bf16_seq": (_c_int, [_c_vp, _c_int, _c_vp, ctypes.c_int64, _c_vp]),
    "pd_wgrad_bf16_time": (_c_int, [_c_vp, _c_vp, ctypes.c_int64, _c_int, _c_vp, _c_vp]),
    "pd_mx8_quantize_bf16": (_c_int, [_c_vp, ctypes.c_int64, _c_int, ctypes.c_int64, _c_int, _c_vp, _c_vp, _c_vp]),
    "pd_mx8_quantize_table_bytes": (ctypes.c_int64, [_c_int]),
    "pd_mx8_quantize_grouped": (_c_int, [_c_vp, _c_int, _c_int, _c_vp, _c_vp, _c_vp]),
    "pd_mx8_gemm": (_c_int, [_c_vp, _c_vp]),
    "pd_mx8_gemm_supported": (_c_int, [_c_int] * 3),
    "pd_filter_transpose_table_bytes": (ctypes.c_int64, [_c_int]),
    "pd_filter_transpose_grouped": (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp]),
    "pd_cmd_fn_index": (_c_int, [ctypes.c_char_p]),
    "pd_cmd_fn_nargs": (_c_int, [_c_int]),
    "pd_cmd_replay": (_c_int, [_c_vp, _c_int, _c_vp, _c_int, _c_vp]),
    "pd_memset_async": (_c_int, [_c_vp, _c_int, ctypes.c_int64, _c_vp]),
    "pd_memcpy_d2d_async": (_c_int, [_c_vp, _c_vp, ctypes.c_int64, _c_vp]),
    "pd_copy_segments": (_c_int, [_c_vp, _c_int, _c_vp]),
    "pd_last_error": (ctypes.c_char_p, []),
    "pd_abi_version": (_c_int, []),
    "pd_debug_set": (_c_int, [ctypes.c_char_p, _c_int]),
}

_lib = None
_PROXY = None            # cmdbuf.Recording: while a region is being recorded load() hands out its recording proxy


class PdHipError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile every csrc/*.hip for gfx950 into libpd_hip.so (in-tree)."""
    args = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    out = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
    if out.returncode != 0:
        raise PdHipError("building libpd_hip.so failed")
    return LIB_PATH


def real():
    """the library itself, never the recording proxy"""
    return _lib if _lib is not None else load()


def load():
    global _lib
    if _PROXY is not None:
        return _PROXY
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PdHipError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C partdistillation_amd/csrc`). "
            "partdistillation_amd has no CPU or PyTorch fallback.")
    # PyTorch ships its own libamdhip64 / libhsa-runtime64; libpd_hip.so is linked against the same SONAME.  Import torch
    # FIRST so that the process has ONE HIP runtime - the one torch's streams and allocations live in.  (Loaded the other way
    # round, the system runtime wins and every launch fails with "no ROCm-capable device is detected".)
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise PdHipError(f"libpd_hip.so does not export `{name}` (stale build?)") from e
        fn.restype, fn.argtypes = res, args
    if lib.pd_abi_version() != ABI_VERSION:
        raise PdHipError(f"libpd_hip.so ABI {lib.pd_abi_version()} != binding {ABI_VERSION}: rebuild")
    # development knobs of the kernels (tools/ab_bench.sh): PD_DEBUG_SET="key=value,key=value" -> pd_debug_set at load
    for kv in filter(None, os.environ.get("PD_DEBUG_SET", "").split(",")):
        k, v = kv.split("=")
        if lib.pd_debug_set(k.encode(), int(v)) != 0:
            raise PdHipError(f"PD_DEBUG_SET: unknown key `{k}`")
    _lib = lib
    return lib


def current_stream():
    """raw hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream().cuda_stream costs
    ~9 us of host time per call (Stream object construction); the two C calls below ~0.5 us - it is called once per launch."""
    import torch
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def check(rc):
    if rc != 0:
        raise PdHipError(load().pd_last_error().decode() or f"libpd_hip error {rc}")
