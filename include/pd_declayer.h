/*
 * pd_declayer.h — C-ABI of the fused query-side decoder layer kernels of libpd_hip.so (csrc/declayer.hip).
 *
 * Reference: transformer_decoder/mask2former_transformer_decoder.py — per layer (:395-439) the masked cross-attention's output
 * projection + residual + LayerNorm (:102-114), self-attention (:44-54), FFN (:167-171) and, in front of every layer, the prediction
 * head forward_prediction_heads (:449-459: decoder_norm + the 3-layer mask-embedding MLP :198-204).  Through rounds 1-5 every Linear,
 * LayerNorm and head of the 200-row (Q x B) query tensor was a launch of its own: ~17 launches of 5-24 us per layer and direction on
 * 7-128 workgroups of a 256-CU part, 289 launches / 2.25 ms per step with nothing to overlap them with.
 *
 * Here a workgroup OWNS 16 rows of the [R = Q*B, 256] query tensor for a whole chain of row-local operators: the rows stay in LDS /
 * registers, every weight matrix streams through the workgroup exactly once as MFMA operands loaded straight from global memory
 * (v_mfma_f32_16x16x32_bf16, the next 32 x 256 weight block always in flight), residual + LayerNorm are the epilogue of the product
 * that feeds them.  Only the attention products themselves (which mix rows) stay separate launches, so a layer is
 *     forward:   [cross-attention]  pd_dec_fwd_a  [self-attention]  pd_dec_fwd_b
 *     backward:  pd_dec_bwd_b  [self-attention backward]  pd_dec_bwd_a  [cross-attention backward]
 * Arithmetic (bf16 operands, fp32 accumulation, results rounded to bf16 where the unfused path stored bf16, LayerNorm in fp32 over
 * the same fp32 sums) follows the unfused kernels of csrc/smallgemm.hip / csrc/rowwise.hip step by step.
 *
 * Conventions: C = 256 channels, feed-forward width 2048; rows r = query * B + image (seq-first like the reference); all activations
 * row-major and contiguous ([R, C] / [R, 2048]); biases bf16; LayerNorm weights fp32; `stats` = [2, R] fp32 (mean row, then rstd row).
 * WEIGHTS are passed PACKED (pd_dec_pack_grouped, once per optimisation step): a workgroup can pull ~57 GB/s from memory when every wave
 * instruction reads 1 KB contiguous and ~30 GB/s in the 16-rows-x-64-bytes pattern an MFMA operand has in a row-major matrix
 * (tools/probes/stream_probe.hip), and that stream is what bounds these kernels.  Forward kernels take pack(W) of the [out, in] weight,
 * backward kernels pack(W^T) ("...T" arguments; transpose = 1 packs straight from the [out, in] weight).
 * `stream` = hipStream_t.  Return 0 or PD_ERR_*.
 */
#ifndef PD_DECLAYER_H
#define PD_DECLAYER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* W_eff = src ([rows, cols] bf16 row-major) or its transpose -> dst in block order: 32 output rows x 256 contraction elements per 16 KB
 * block, block (nb, kc) at element offset (nb * (K / 256) + kc) * 8192, 16-byte piece (i = 8 t + s, lane) of a block =
 * W_eff[32 nb + 16 t + (lane & 15)][256 kc + 32 s + 8 (lane >> 4) .. + 7].  Output rows % 32 == 0, contraction % 256 == 0; dst holds
 * rows * cols elements.  All problems in ONE launch; table_host_pinned / table_device: pd_dec_pack_table_bytes(count) bytes each. */
typedef struct PdDecPack {
  const void *src;
  void *dst;
  int32_t rows, cols, transpose;
} PdDecPack;
int64_t pd_dec_pack_table_bytes(int count);
int pd_dec_pack_grouped(const PdDecPack *descs, int count, void *table_host_pinned, void *table_device, void *stream);

/* pd_dec_fwd_b / pd_dec_bwd_b with a workspace run the FFN part as TWO launches: pd_dec_split() workgroups per row block (PD_DEC_SPLIT = 2 | 4 | 8,
 * default 4; 1: one launch) each take 1 / split of the hidden columns and leave fp32 partial rows in the workspace, the second launch (one
 * workgroup per row block) sums them in slab order and finishes the chain.  workspace: pd_dec_workspace_bytes(R) bytes, no initial state, not
 * shared by launches that can overlap; NULL: one launch. */
int pd_dec_split(void);
int64_t pd_dec_workspace_bytes(int R);

/* x = bf16(o W_o^T + b_o);  z = x + res;  y = LayerNorm(z) * ln_w + ln_b;  y_c = bf16(y);  ypos_c = bf16(y + qpos[r / pos_div]);
 * q = bf16(ypos_c W_q^T + b_q), k = bf16(ypos_c W_k^T + b_k), v = bf16(y_c W_v^T + b_v)   (w_qkv = [W_q; W_k; W_v], [3C, C]) */
int pd_dec_fwd_a(const void *o, const float *res, const float *qpos, int pos_div, const void *w_o, const void *b_o, const float *ln_w,
                 const float *ln_b, float eps, const void *w_qkv, const void *b_qkv, float *z, float *stats, float *y, void *y_c,
                 void *ypos_c, void *q, void *k, void *v, int R, void *stream);

/* flags bit 0: run the layer part (o != NULL): x = bf16(o W_o^T + b_o); z2 = x + res; y2 = LN(z2); y2_c = bf16(y2);
 *                  h = bf16(relu(y2_c W_1^T + b_1)); x3 = bf16(h W_2^T + b_2); z3 = x3 + y2; y3 = LN(z3)
 *              (without bit 0: y3 = res — the head in front of the first layer);
 *       bit 1: the prediction head's mask-embedding MLP and the next layer's cross-attention query projection follow:
 *                  e = bf16(M_2 relu(M_1 relu(M_0 d_c + ..) ..) ..) -> ef[b][q][:] (batch-major), qc = bf16(ypos_c W_qn^T + b_qn)
 * always: d = LayerNorm(y3) * dn_w + dn_b -> dec_out (fp32), hstats;  ypos_c = bf16(y3 + qpos[r / pos_div]) */
int pd_dec_fwd_b(const void *o, const float *res, const float *qpos, int pos_div, const void *w_o, const void *b_o, const float *ln2_w,
                 const float *ln2_b, const void *w_1, const void *b_1, const void *w_2, const void *b_2, const float *ln3_w,
                 const float *ln3_b, const float *dn_w, const float *dn_b, const void *m0_w, const void *m0_b, const void *m1_w,
                 const void *m1_b, const void *m2_w, const void *m2_b, const void *wq_next, const void *bq_next, float eps, float *z2,
                 float *stats2, void *y2_c, void *h, float *z3, float *stats3, float *y3, void *ypos_c, float *dec_out, float *hstats,
                 void *ef, void *qc_next, void *workspace, int R, int flags, void *stream);

/* backward of pd_dec_fwd_b's layer part (no gradient flows through the mask-embedding MLP inside the loop: the reference detaches
 * the per-layer mask prediction, :457; the gradient-carrying heads run outside on the stack of decoder outputs):
 *   d_pos = bf16(dqc_next W_qn)                                   (wqT_next = W_qn^T; NULL: no next layer)
 *   dzh   = LNbwd(y3, hstats, dn_w; d_out)                        dgb_dn[2C] += [dgamma | dbeta]
 *   dz3   = LNbwd(z3, stats3, ln3_w; dzh + d_res + d_pos)         dgb3[2C] += [dgamma | dbeta], db3[C] += colsum(dz3);  pos_acc[r / pos_div] += d_pos
 *   dh    = bf16((dz3_c W_2) * (h > 0)),  dx = bf16(dh W_1)       (w2T = W_2^T [2048, C], w1T = W_1^T [C, 2048])
 *   dz2   = LNbwd(z2, stats2, ln2_w; dz3 + dx)                    dgb2 / db2 likewise
 *   d_o   = bf16(dz2_c W_o)                                       (woT = W_o^T)
 * outputs: dz3_c, dh (weight-gradient operands), dz2 (fp32), dz2_c, d_o.  d_res may be NULL (last layer). */
int pd_dec_bwd_b(const void *dqc_next, const void *wqT_next, const float *d_out, const float *d_res, const float *y3, const float *hstats,
                 const float *dn_w, float *dgb_dn, const float *z3, const float *stats3, const float *ln3_w, float *dgb3, float *db3,
                 float *pos_acc, int pos_div, const void *w2T, const void *h, const void *w1T, const float *z2, const float *stats2,
                 const float *ln2_w, float *dgb2, float *db2, const void *woT, void *dz3_c, void *dh, float *dz2, void *dz2_c, void *d_o,
                 void *workspace, int R, void *stream);

/* backward of pd_dec_fwd_a:
 *   d_tp = bf16(dq W_q + dk W_k), d_tc = bf16(dv W_v)             (wqkvT = [W_q; W_k; W_v]^T, [C, 3C])
 *   dz1  = LNbwd(z, stats, ln_w; dz_in + d_tc + d_tp)             dgb[2C] += [dgamma | dbeta], db[C] += colsum(dz1);  pos_acc[r / pos_div] += d_tp
 *   d_o  = bf16(dz1_c W_o)
 * outputs: dz1 (fp32: the gradient w.r.t. the residual stream entering the layer), dz1_c, d_o. */
int pd_dec_bwd_a(const void *dq, const void *dk, const void *dv, const void *wqkvT, const float *dz_in, const float *z, const float *stats,
                 const float *ln_w, float *dgb, float *db, float *pos_acc, int pos_div, const void *woT, float *dz1, void *dz1_c, void *d_o,
                 int R, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PD_DECLAYER_H */
