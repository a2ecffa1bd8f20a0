/*
 * pd_window_attention.h — C-ABI of the Swin (shifted-)window attention kernels of libpd_hip.so: window 12 x 12 = 144
 * tokens, head_dim 32, bf16 operands on the matrix cores, fp32 scores / softmax.
 *
 * Replaces, per Swin block, the body of the reference's WindowAttention.forward
 *   modeling/backbone/swin.py:135-175
 *     q*scale, q @ k^T                                   :146-150
 *     + relative_position_bias_table[relative_position_index]  [heads,144,144]   :152-160
 *     + the SW-MSA mask (0 / -100) of the window           :162-167 (mask built in BasicLayer.forward :417-444)
 *     softmax, attn @ v, transpose back to [B_, N, C]      :168-173
 * i.e. everything between the qkv Linear and the proj Linear.  The reference materialises the [B_, heads, 144, 144]
 * score tensor ~10 times per block forward+backward (363 MB each at Swin-L stage 1, 1280^2, batch 2); these kernels
 * keep one 16 x 144 score strip in registers, read q/k/v straight out of the qkv Linear's [B_, 144, 3C] output and
 * write the [B_, 144, C] layout the proj Linear consumes — no permutes, no expanded bias, no score tensor.
 *
 * The bias is looked up in the [529, heads] TABLE itself (staged in LDS): for a 12 x 12 window
 *   index(q, key) = A(q) - A(key) + 264,  A(t) = t + 11 * (t / 12)
 * (equal to the reference's relative_position_index :110-125), and the backward accumulates the table gradient the same
 * way, so neither the [heads,144,144] bias nor its gradient exists in memory.
 * The SW-MSA mask is passed as REGION IDS: region[w][t] = the label the reference paints into img_mask (:425-433) for
 * token t of window w; mask(w, i, j) = (region[w][i] != region[w][j]) ? -100 : 0 exactly as :438-441.
 *
 * Layouts (elements):
 *   qkv, dqkv : bf16 [B_, 144, 3*C], C = heads*32; q of head h at column h*32, k at C + h*32, v at 2C + h*32
 *   out, d_out: bf16 [B_, 144, C]
 *   table     : fp32 [529, heads]; dtable: fp32 [529, heads], ACCUMULATED into (atomics) — the caller zero-fills
 *   region    : uint8 [nW, 144] or NULL (no mask); window of row b_ is b_ % nW (reference :163 view(B_/nW, nW, ...))
 *   win_flags : uint8 [nW], nonzero = this window has more than one region (others skip the mask arithmetic)
 *   lse       : fp32 [B_, heads, 144], BASE-2 log-sum-exp of the scores (forward output, backward input)
 * `stream` = hipStream_t.  Returns 0 or PD_ERR_* (pd_msda.h); message via pd_last_error().
 */
#ifndef PD_WINDOW_ATTENTION_H
#define PD_WINDOW_ATTENTION_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* out_q / out_s (nullable, together): `out` again as an MX-fp8 operand (include/pd_mx8.h: [B_ * 144][C] fp8 of q_format, [B_ * 144][C / 32]
 * E8M0 bytes) for the proj Linear — a (token, head) piece is one 32-element block */
int pd_window_attn_fwd_w12(const void *qkv, const float *table, const uint8_t *region, const uint8_t *win_flags, void *out,
                           float *lse, int B_, int nW, int heads, float scale, void *out_q, void *out_s, int q_format, void *stream);

/* dqkv is written completely; dtable is accumulated into.  dqkv_q / dqkv_s (nullable, together): dqkv again as an MX-fp8 operand
 * ([B_ * 144][3 C], [B_ * 144][3 C / 32]) for the input gradient of the qkv Linear. */
int pd_window_attn_bwd_w12(const void *qkv, const float *table, const uint8_t *region, const uint8_t *win_flags,
                           const void *out, const void *d_out, const float *lse, void *dqkv, float *dtable, int B_, int nW,
                           int heads, float scale, void *dqkv_q, void *dqkv_s, int q_format, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PD_WINDOW_ATTENTION_H */
