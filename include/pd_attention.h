/*
 * pd_attention.h — C-ABI of the masked multi-head attention kernels of libpd_hip.so (head_dim 32).
 *
 * Replaces the nn.MultiheadAttention calls of the reference's masked-attention decoder
 *   transformer_decoder/mask2former_transformer_decoder.py:49-50 (self-attention over the Q queries)
 *   transformer_decoder/mask2former_transformer_decoder.py:107-110 (cross-attention, Q=100..200 queries against
 *     1024..25600 memory tokens with a per-(image, query, key) boolean mask, True = may not attend)
 * after the input/output projections (which stay library GEMMs).  The query count is tiny and the key count large,
 * so parallelism comes from splitting the KEYS across workgroups: each workgroup produces a partial
 * (max, sum, weighted values) per query and a second kernel merges the partials (log-sum-exp algebra).
 *
 * Layouts (elements; T = bf16 or fp32 according to `dtype`, PD_BF16 / PD_F32 of pd_msda.h):
 *   q, o, do, dq : [Lq, B, H*32]   element (l, b, h, d) at (l*B + b)*H*32 + h*32 + d     (nn.MultiheadAttention's seq-first)
 *   k, v, dk, dv : [Lk, B, H*32]
 *   mask         : uint8/bool [B, Lq, Lk] or NULL (shared by the heads); nonzero = blocked (-inf)
 *   lse          : fp32 [B, H, Lq]   log-sum-exp of the scaled, masked scores (forward output, backward input)
 *   workspace    : fp32, at least pd_attn_workspace_floats(...) elements
 * Scores are q.k * scale.  A row whose keys are all blocked yields o = 0 and lse = -inf (the decoder un-blocks such
 * rows beforehand, reference :405).
 */
#ifndef PD_ATTENTION_H
#define PD_ATTENTION_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int64_t pd_attn_workspace_floats(int B, int H, int Lq, int Lk);

int pd_attn_fwd_d32(const void *q, const void *k, const void *v, const uint8_t *mask, void *o, float *lse,
                    float *workspace, int B, int H, int Lq, int Lk, float scale, int dtype, void *stream);

/* The same with k and v as COLUMN SLICES of a wider row-major matrix: ld_kv elements between consecutive (key, image) rows (>= H * 32, a multiple
 * of 8).  bf16, <= 128 queries (the matrix-core kernels) only.  The decoder's key / value projections of the layers that share a memory level
 * are one product [rows, layers * 256]; each layer's attention reads its 256 columns in place. */
int pd_attn_fwd_d32_ld(const void *q, const void *k, const void *v, const uint8_t *mask, void *o, float *lse,
                       float *workspace, int B, int H, int Lq, int Lk, float scale, int dtype, int ld_kv, void *stream);

/* dq/dk/dv are written completely (no accumulation into their previous contents). */
int pd_attn_bwd_d32(const void *q, const void *k, const void *v, const uint8_t *mask, const void *o, const void *d_o,
                    const float *lse, void *dq, void *dk, void *dv, float *workspace, int B, int H, int Lq, int Lk,
                    float scale, int dtype, void *stream);

/* k / v strided as in pd_attn_fwd_d32_ld; dq / dk / dv dense. */
int pd_attn_bwd_d32_ld(const void *q, const void *k, const void *v, const uint8_t *mask, const void *o, const void *d_o,
                       const float *lse, void *dq, void *dk, void *dv, float *workspace, int B, int H, int Lq, int Lk,
                       float scale, int dtype, int ld_kv, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PD_ATTENTION_H */
