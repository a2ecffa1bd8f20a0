// Command buffers: a recorded sequence of C-ABI calls of this library replayed from C++ by ONE call (include/pd_cmdbuf.h).
//
// Why: the fused transformer cores (functions/encoder_core.py, decoder_core.py) issue 50-450 launches per step through Python faces
// that allocate, assert and marshal ~15 arguments each: ~7-15 us of host time per launch where the launch itself costs ~3.  Whole-step
// hipGraph replay is ruled out on ROCm 7.2 ("hipGraph": packet-captured graphs go stale, the safe mode costs more host
// time than eager issue).  A command buffer is the same idea one level up: the FIRST execution of a region runs the ordinary
// Python faces while a recorder (partdistillation_amd/cmdbuf.py) notes every pd_* call with its argument words — buffers come from a
// persistent arena, so every address is either stable, or an offset into one of the region's input tensors ("slot", patched at
// replay), or the stream — and every later execution is pd_cmd_replay: plain launches on the current stream, in order, no graph.
//
// Each recordable function gets a typed thunk generated from its real prototype (floats / doubles / 64-bit integers travel in the
// registers the C ABI assigns them; nothing is guessed from the argument words).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <type_traits>
#include <utility>

#include "pd_common.h"
#include "pd_cmdbuf.h"
#include "pd_attention.h"
#include "pd_conv.h"
#include "pd_declayer.h"
#include "pd_criterion.h"
#include "pd_fused.h"
#include "pd_gemm.h"
#include "pd_grouping.h"
#include "pd_igemm.h"
#include "pd_input.h"
#include "pd_kmeans.h"
#include "pd_msda.h"
#include "pd_mx8.h"
#include "pd_optim.h"
#include "pd_rowwise.h"
#include "pd_smallgemm.h"
#include "pd_stem.h"
#include "pd_swin.h"
#include "pd_window_attention.h"

namespace {

template <class T>
inline T word_to(uint64_t w)
{
  if constexpr (std::is_pointer_v<T>) {
    return reinterpret_cast<T>(static_cast<uintptr_t>(w));
  } else if constexpr (std::is_same_v<T, float>) {
    float f; uint32_t u = (uint32_t)w; memcpy(&f, &u, 4); return f;
  } else if constexpr (std::is_same_v<T, double>) {
    double d; memcpy(&d, &w, 8); return d;
  } else {
    return static_cast<T>(static_cast<int64_t>(w));
  }
}

template <auto F> struct Thunk;
template <class... A, int (*F)(A...)>
struct Thunk<F> {
  static constexpr int nargs = (int)sizeof...(A);
  template <size_t... I>
  static int invoke(const uint64_t *w, std::index_sequence<I...>) { return F(word_to<A>(w[I])...); }
  static int call(const uint64_t *w) { return invoke(w, std::index_sequence_for<A...>{}); }
};

struct Entry { const char *name; int (*call)(const uint64_t *); int nargs; };
#define PD_E(f) {#f, &Thunk<&f>::call, Thunk<&f>::nargs}
const Entry kTable[] = {
  PD_E(pd_adamw_clipped),
  PD_E(pd_adamw_clipped_shadow),
  PD_E(pd_add_layernorm_bwd),
  PD_E(pd_add_layernorm_bwd_amax),
  PD_E(pd_add_layernorm_fwd),
  PD_E(pd_add_layernorm_fwd_amax),
  PD_E(pd_add_rows_amax_f32),
  PD_E(pd_affine_act_bwd2_bf16),
  PD_E(pd_affine_act_bwd_bf16),
  PD_E(pd_affine_act_fwd_bf16),
  PD_E(pd_attn_bwd_d32),
  PD_E(pd_attn_fwd_d32),
  PD_E(pd_attn_mask_u8),
  PD_E(pd_cast_bf16_f32_amax),
  PD_E(pd_colsum_acc),
  PD_E(pd_conv3x3_nhwc_f16x2),
  PD_E(pd_conv3x3_nhwc_f32x3),
  PD_E(pd_conv3x3_wgrad_nhwc_f16x2),
  PD_E(pd_conv3x3_wgrad_nhwc_f32x3),
  PD_E(pd_conv_bf16_dgrad),
  PD_E(pd_conv_bf16_fwd),
  PD_E(pd_conv_bf16_wgrad),
  PD_E(pd_conv_bf16_wgrad_grouped),
  PD_E(pd_dec_bwd_a),
  PD_E(pd_dec_bwd_b),
  PD_E(pd_dec_fwd_a),
  PD_E(pd_dec_fwd_b),
  PD_E(pd_dec_pack_grouped),
  PD_E(pd_decoder_head_bf16),
  PD_E(pd_filter_transpose_grouped),
  PD_E(pd_gemm_tn_f16x2),
  PD_E(pd_gemm_tn_f16x2_bf16out),
  PD_E(pd_gemm_tn_f32),
  PD_E(pd_gemm_tn_f32x3),
  PD_E(pd_gemm_tn_f32x3_pre),
  PD_E(pd_gemm_tn_f32x3_relu_bits),
  PD_E(pd_gemm_tn_f32x3_relumask),
  PD_E(pd_gemm_wgrad_acc_f16x2_ws),
  PD_E(pd_gemm_wgrad_acc_f32),
  PD_E(pd_gemm_wgrad_acc_f32x3),
  PD_E(pd_gemm_wgrad_acc_f32x3_ws),
  PD_E(pd_gemm_wgrad_f16x2_grouped),
  PD_E(pd_gemm_wgrad_f32),
  PD_E(pd_gemm_wgrad_f32x3_grouped),
  PD_E(pd_gn_coeffs_bwd),
  PD_E(pd_gn_coeffs_fwd),
  PD_E(pd_igemm_bf16),
  PD_E(pd_igemm_bf16_seq),
  PD_E(pd_kmeans_assign),
  PD_E(pd_kmeans_assign_bounded),
  PD_E(pd_kmeans_assign_partial),
  PD_E(pd_kmeans_reduce),
  PD_E(pd_kmeans_reduce_update),
  PD_E(pd_kmeans_reduce_update_shift),
  PD_E(pd_kmeans_update),
  PD_E(pd_layernorm_rows_f32_bwd),
  PD_E(pd_layernorm_rows_f32_fwd),
  PD_E(pd_loss_vectors_bwd),
  PD_E(pd_loss_vectors_fwd),
  PD_E(pd_lsa_batched),
  PD_E(pd_mask_assign),
  PD_E(pd_mask_point_losses_bwd),
  PD_E(pd_mask_point_losses_fwd),
  PD_E(pd_matcher_costs),
  PD_E(pd_copy_segments),
  PD_E(pd_attn_fwd_d32_ld),
  PD_E(pd_attn_bwd_d32_ld),
  PD_E(pd_match_point_logits),
  PD_E(pd_swin_merge_ln_fwd),
  PD_E(pd_swin_tail_ln_fwd),
  PD_E(pd_swin_tail_ln_bwd),
  PD_E(pd_swin_merge_ln_bwd),
  PD_E(pd_matcher_point_terms),
  PD_E(pd_maxpool3s2_bwd_bf16),
  PD_E(pd_maxpool3s2_fwd_bf16),
  PD_E(pd_mem_prep_bwd),
  PD_E(pd_mem_prep_fwd),
  PD_E(pd_memcpy_d2d_async),
  PD_E(pd_memset_async),
  PD_E(pd_msda_backward),
  PD_E(pd_msda_forward),
  PD_E(pd_msda_forward_amax),
  PD_E(pd_msda_fused_backward),
  PD_E(pd_msda_fused_forward),
  PD_E(pd_msda_prep_bwd),
  PD_E(pd_msda_prep_bwd_amax),
  PD_E(pd_msda_prep_fwd),
  PD_E(pd_multi_gather_sumsq),
  PD_E(pd_mx8_gemm),
  PD_E(pd_mx8_quantize_bf16),
  PD_E(pd_mx8_quantize_grouped),
  PD_E(pd_nc_affine2_amax_f32),
  PD_E(pd_nc_affine2_f32),
  PD_E(pd_nc_affine_amax_f32),
  PD_E(pd_nc_affine_f32),
  PD_E(pd_nc_sums_f32),
  PD_E(pd_normalize_u8_nhwc),
  PD_E(pd_pair_logits_bwd_rows),
  PD_E(pd_pair_logits_bwd_tok),
  PD_E(pd_pair_logits_fwd),
  PD_E(pd_point_sample_nhwc_f32),
  PD_E(pd_point_sample_nhwc_f32_bf16),
  PD_E(pd_point_sample_planar_bwd_f32),
  PD_E(pd_point_sample_planar_f32),
  PD_E(pd_point_sample_u8),
  PD_E(pd_relu_bwd_colsum),
  PD_E(pd_resample_cols_u8),
  PD_E(pd_resample_rows_u8),
  PD_E(pd_resize_bilinear_nhwc_f32),
  PD_E(pd_rle_sample_u8),
  PD_E(pd_row_amax_f32),
  PD_E(pd_scores_argmax_u8),
  PD_E(pd_sgemm_nn_bf16),
  PD_E(pd_sgemm_nn_splitn_bf16),
  PD_E(pd_sgemm_tn_batched_bf16),
  PD_E(pd_sgemm_tn_bf16),
  PD_E(pd_sgemm_tn_multi_bf16),
  PD_E(pd_sgemm_tn_splitk_bf16),
  PD_E(pd_sgemm_wgrad_bf16),
  PD_E(pd_sgemm_wgrad_grouped_bf16),
  PD_E(pd_sgemm_wgrad_split_bf16),
  PD_E(pd_skinny_linear_bwd),
  PD_E(pd_skinny_linear_fwd),
  PD_E(pd_split3_bf16),
  PD_E(pd_stem7x7_fwd),
  PD_E(pd_stem7x7_wgrad),
  PD_E(pd_sum3_sum2_f32),
  PD_E(pd_sumsq_accumulate),
  PD_E(pd_swin_ln_bwd),
  PD_E(pd_swin_ln_fwd),
  PD_E(pd_transpose_batched_f32),
  PD_E(pd_uncertain_points),
  PD_E(pd_upsample2x_bwd_nhwc_f32),
  PD_E(pd_upsample_add_amax_nhwc_f32),
  PD_E(pd_upsample_add_nhwc_f32),
  PD_E(pd_wgrad_bf16),
  PD_E(pd_wgrad_bf16_seq),
  PD_E(pd_window_attn_bwd_w12),
  PD_E(pd_window_attn_fwd_w12),
};
constexpr int kCount = (int)(sizeof(kTable) / sizeof(kTable[0]));
}  // namespace

extern "C" int pd_memset_async(void *dst, int value, int64_t bytes, void *stream)
{
  if (bytes <= 0) return PD_OK;
  return hipMemsetAsync(dst, value, (size_t)bytes, (hipStream_t)stream) == hipSuccess ? PD_OK : pd_set_error(PD_ERR_LAUNCH, "pd_memset_async failed");
}

extern "C" int pd_memcpy_d2d_async(void *dst, const void *src, int64_t bytes, void *stream)
{
  if (bytes <= 0) return PD_OK;
  return hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? PD_OK
                                                                                                         : pd_set_error(PD_ERR_LAUNCH, "pd_memcpy_d2d_async failed");
}

extern "C" int pd_cmd_fn_index(const char *name)
{
  if (!name) return -1;
  for (int i = 0; i < kCount; ++i)
    if (!strcmp(kTable[i].name, name)) return i;
  return -1;
}

extern "C" int pd_cmd_fn_nargs(int fn) { return fn >= 0 && fn < kCount ? kTable[fn].nargs : -1; }

extern "C" int pd_cmd_replay(const PdCmd *cmds, int count, const uint64_t *slot_bases, int nslots, void *stream)
{
  uint64_t w[PD_CMD_MAX_ARGS];
  for (int c = 0; c < count; ++c) {
    const PdCmd &k = cmds[c];
    if (k.fn < 0 || k.fn >= kCount || k.nargs != kTable[k.fn].nargs || k.nargs > PD_CMD_MAX_ARGS)
      return pd_set_error(PD_ERR_INVALID_ARG, "pd_cmd_replay: command %d is malformed", c);
    for (int i = 0; i < k.nargs; ++i) {
      const int s = k.kind[i];
      if (s == PD_CMD_LITERAL) w[i] = k.a[i];
      else if (s == PD_CMD_STREAM) w[i] = (uint64_t)(uintptr_t)stream;
      else if (s >= 0 && s < nslots) w[i] = slot_bases[s] + k.a[i];
      else return pd_set_error(PD_ERR_INVALID_ARG, "pd_cmd_replay: command %d argument %d names slot %d of %d", c, i, s, nslots);
    }
    const int rc = kTable[k.fn].call(w);
    if (rc != PD_OK) return rc;
  }
  return PD_OK;
}
