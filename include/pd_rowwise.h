/*
 * pd_rowwise.h — C-ABI of the token-wise ("one wavefront per row") kernels of libpd_hip.so used by the transformer
 * layers of the training step.  Each replaces a chain of eager PyTorch launches of the reference:
 *
 *   pd_add_layernorm_{fwd,bwd}   `tgt = norm(tgt + dropout(sublayer(tgt)))` of every post-norm layer —
 *                                transformer_decoder/mask2former_transformer_decoder.py:44-54 (self-attention), :102-114
 *                                (cross-attention), :167-171 (FFN), decoder_norm :438; pixel_decoder/msdeformattn.py
 *                                MSDeformAttnTransformerEncoderLayer.forward (norm1/norm2) — plus the `with_pos_embed`
 *                                add (:41-42, :80-81) and the half-precision casts autocast inserts in front of the next
 *                                GEMM, produced as extra outputs of the same pass.  Row statistics are 64-lane
 *                                wavefront reductions (DPP), one row per wavefront, 16-byte lanes.
 *   pd_colsum_acc / pd_relu_bwd_colsum   bias gradients (column sums) of a GEMM, optionally fused with the ReLU
 *                                backward of the FFN / MLP hidden layer (:167-171, MLP :198-204).
 *   pd_mem_prep_{fwd,bwd}        decoder memory of one feature level (:392-401): tokens [B,HW,C] + level_embed ->
 *                                seq-first `memory` and `memory + pos` in the GEMM dtype, one pass.
 *   pd_attn_mask_u8              `(sigmoid(mask logits) < 0.5)` with fully-blocked rows released (:405, :455-459).
 *   pd_point_sample_nhwc_f32     point_sample of a channels-last map at points shared by all channels (matcher).
 *   pd_msda_prep_{fwd,bwd}       MSDeformAttn.forward between the projections and the sampling core
 *                                (pixel_decoder/ops/modules/ms_deform_attn.py:108-117): softmax of the attention logits
 *                                and `reference_points + offsets / (W_l, H_l)` in one pass, and their backward.
 *
 * Device pointers; dtypes are PD_F32 / PD_BF16 (pd_msda.h); `stream` = hipStream_t; returns 0 or a negative PD_ERR_*.
 * C (the normalised / channel dimension) must be a multiple of 256 and <= 1024.
 */
#ifndef PD_ROWWISE_H
#define PD_ROWWISE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * z = x + res (either may be NULL, not both);  y = LayerNorm(z) * gamma + beta  (biased variance, eps inside the sqrt).
 *   x       [rows, C] of x_dtype (the sublayer's GEMM output)      res   fp32 [rows, C]
 * Outputs (each nullable): z fp32 (what the backward needs), y fp32, y_c = y in c_dtype, ypos_c = y + pos[row / pos_div]
 * in c_dtype (pos fp32 [ceil(rows / pos_div), C]); mean / rstd fp32 [rows] (required).
 */
int pd_add_layernorm_fwd(const void *x, int x_dtype, const float *res, const float *gamma, const float *beta, float eps,
                         float *z, float *y, void *y_c, const float *pos, int pos_div, void *ypos_c, int c_dtype,
                         float *mean, float *rstd, int rows, int C, void *stream);

/*
 * g = dy + dy2 + dy_c + dypos_c (each nullable, at least one given; dy / dy2 fp32, dy_c / dypos_c of c_dtype);
 * dz = LayerNorm backward of g through (z, mean, rstd, gamma):  written to dz (fp32, may alias dy) and, when given, to
 * dz_c in dzc_dtype.  Accumulated (+=, the caller zeroes them; each nullable):
 *   dgamma[C] += sum_rows g * xhat      dbeta[C] += sum_rows g      dbias[C] += sum_rows dz
 *   dpos_acc[row / pos_div, C] += dypos_c[row]   (gradient of the positional table added in the forward)
 */
int pd_add_layernorm_bwd(const float *dy, const float *dy2, const void *dy_c, const void *dypos_c, int c_dtype,
                         const float *z, const float *mean, const float *rstd, const float *gamma, float *dz, void *dz_c,
                         int dzc_dtype, float *dgamma, float *dbeta, float *dbias, float *dpos_acc, int pos_div,
                         int rows, int C, void *stream);

/* The two calls above with the absolute row maxima of their outputs written next to them (each nullable): y_amax[rows] = max |y|,
 * ypos_amax[rows] = max |y + pos| (needs ypos_c), dz_amax[rows] = max |dz| — the row-scaling input of pd_gemm_tn_f16x2
 * (include/pd_gemm.h) for the GEMMs that consume these tensors, produced where the row is in registers anyway. */
int pd_add_layernorm_fwd_amax(const void *x, int x_dtype, const float *res, const float *gamma, const float *beta, float eps,
                              float *z, float *y, void *y_c, const float *pos, int pos_div, void *ypos_c, int c_dtype,
                              float *mean, float *rstd, float *y_amax, float *ypos_amax, int rows, int C, void *stream);
int pd_add_layernorm_bwd_amax(const float *dy, const float *dy2, const void *dy_c, const void *dypos_c, int c_dtype,
                              const float *z, const float *mean, const float *rstd, const float *gamma, float *dz, void *dz_c,
                              int dzc_dtype, float *dgamma, float *dbeta, float *dbias, float *dpos_acc, int pos_div,
                              float *dz_amax, int rows, int C, void *stream);

/* acc[N] (fp32) += column sums of x [rows, N] (dtype).  N % 128 == 0. */
int pd_colsum_acc(const void *x, int dtype, int rows, int N, float *acc, void *stream);

/* dh [rows, N] *= (h > 0) in place, then acc[N] += column sums of the result (acc nullable). */
int pd_relu_bwd_colsum(void *dh, const void *h, int dtype, int rows, int N, float *acc, void *stream);

/*
 * tok: fp32 tokens of one level, element (b, p, c) at tok[b * tok_batch_stride + p * C + c]; level_embed fp32 [C]
 * (nullable); pos fp32 [HW, C].  Writes seq-first [HW, B, C] (row p * B + b) in c_dtype:
 *   mem_c = tok + level_embed        mempos_c = tok + level_embed + pos[p]
 */
int pd_mem_prep_fwd(const float *tok, int64_t tok_batch_stride, const float *level_embed, const float *pos, void *mem_c,
                    void *mempos_c, int c_dtype, int B, int HW, int C, void *stream);

/* dtok[b * dtok_batch_stride + p * C + c] = dmem_c[p * B + b, c] + dmempos_c[p * B + b, c]  (fp32 out; either input nullable) */
int pd_mem_prep_bwd(const void *dmem_c, const void *dmempos_c, int c_dtype, float *dtok, int64_t dtok_batch_stride, int B,
                    int HW, int C, void *stream);

/* mask[r, k] = logits[r, k] < 0, except rows where that holds for every k, which become all 0.  logits [rows, n] dtype. */
int pd_attn_mask_u8(const void *logits, int dtype, int rows, int n, uint8_t *mask, void *stream);

/* The Hungarian matcher's per-point terms in one pass (reference modeling/matcher.py:108-158: batch_sigmoid_ce_loss_jit and
 * batch_dice_loss_jit on the sampled points; replaces .float(), F.softplus, .sigmoid() and two .sum(-1) over [B * heads * Q, points]):
 * x [rows, n] fp32 / bf16 -> x_f32 [rows, n] (nullable), sigmoid_x [rows, n], softplus_sum [rows] = sum_k softplus(x[r, k]) (beta 1,
 * threshold 20), sigmoid_sum [rows] = sum_k sigmoid(x[r, k]). */
int pd_matcher_point_terms(const void *x, int dtype, int rows, int n, float *x_f32, float *sigmoid_x, float *softplus_sum,
                           float *sigmoid_sum, void *stream);

/*
 * offs fp32 [tokens, M, L, P, 2] and logits fp32 [tokens, M, L*P] with row strides ld_offs / ld_logits in elements (so both can
 * be column ranges of ONE projection output), ref fp32 [tokens, L, 2] (x, y in [0,1]), spatial_shapes int64 [L, 2] (H_l, W_l)
 * on the device ->
 *   loc [tokens, M, L, P, 2] = ref[token, l] + offs / (W_l, H_l)        attn [tokens, M, L*P] = softmax(logits)
 * exactly the two roundings (divide, add) of the reference expression.
 */
int pd_msda_prep_fwd(const float *offs, const float *logits, const float *ref, const int64_t *spatial_shapes, float *loc,
                     float *attn, int64_t tokens, int M, int L, int P, int ld_offs, int ld_logits, void *stream);

/* d_offs = gloc / (W_l, H_l);  d_logits = attn * (gattn - sum_j attn_j * gattn_j) */
int pd_msda_prep_bwd(const float *gloc, const float *gattn, const float *attn, const int64_t *spatial_shapes, float *d_offs,
                     float *d_logits, int64_t tokens, int M, int L, int P, int ld_offs, int ld_logits, void *stream);
/* the same with row_amax[tokens] = max |.| over each token's d_offs and d_logits (8 heads, L P in {8, 12, 16}) for pd_gemm_tn_f16x2 */
int pd_msda_prep_bwd_amax(const float *gloc, const float *gattn, const float *attn, const int64_t *spatial_shapes, float *d_offs, float *d_logits,
                          float *row_amax, int64_t tokens, int M, int L, int P, int ld_offs, int ld_logits, void *stream);

/*
 * out[b, p, :] = bilinear sample of in[b] (fp32, channels-last [B, H, W, C], C % 4 == 0) at coords[b, p] = (x, y) in
 * [0, 1]: F.grid_sample(mode="bilinear", padding_mode="zeros", align_corners=False) with grid = 2*coords - 1, for
 * points shared by all channels (detectron2 point_sample as the matcher uses it, reference matcher.py:128-139, applied
 * to the mask FEATURES: bilinear sampling commutes with the mask_embed . mask_features product, so the Q masks of a head
 * need not exist to be sampled).  Same operation order as the torch kernel.
 */
int pd_point_sample_nhwc_f32(const float *in, const float *coords, float *out, int B, int H, int W, int C, int P, void *stream);
/* the same with the samples rounded to bf16 on the way out (the matcher multiplies them by bf16 mask embeddings under autocast) */
int pd_point_sample_nhwc_f32_bf16(const float *in, const float *coords, void *out, int B, int H, int W, int C, int P, void *stream);

/*
 * The same sampling of PLANAR maps: out[n, c, p] = bilinear sample of in[n, c] ([N, C, H, W] fp32 contiguous) at coords[n, p] — points
 * per map, shared by its C channels (detectron2 point_sample as the criterion uses it on the matched mask logits [N, 1, h, w] and on
 * the target masks, reference criterion.py:165-197), and its gradient with respect to `in`: grad_in (ZERO-FILLED by the caller)
 * += the bilinear weights times grad_out[n, c, p] (fp32 atomics; single-channel maps of <= 16 LDS tiles are instead WRITTEN in full by a tiled kernel without global atomics — pd_point_sample_planar_bwd_needs_zero() tells which).  Same operation order as torch's grid_sampler in the forward.
 */
int pd_point_sample_planar_f32(const float *in, const float *coords, float *out, int N, int C, int H, int W, int P, void *stream);
int pd_point_sample_planar_bwd_f32(const float *grad_out, const float *coords, float *grad_in, int N, int C, int H, int W, int P, void *stream);
int pd_point_sample_planar_bwd_needs_zero(int C, int H, int W);
/* the same question for N maps (more than 65 535 maps always take the accumulating kernel): what callers should use */
int pd_point_sample_planar_bwd_needs_zero_n(int N, int C, int H, int W);

/*
 * FPN top-down step of the pixel decoder (reference msdeformattn.py:356-358:
 * `y = cur_fpn + F.interpolate(out[-1], size=cur_fpn.shape[-2:], mode="bilinear", align_corners=False)`), channels-last
 * fp32 [B, H, W, C] / [B, h, w, C]; same source-index arithmetic as torch's upsample_bilinear2d.
 *   pd_upsample_add_nhwc_f32     y = cur + upsample(lo), any (h, w) -> (H, W); lo_batch_stride: floats between the images of lo (0 = h w C;
 *                                larger when lo is one level of a [B, tokens, C] tensor, read in place)
 *   pd_upsample2x_bwd_nhwc_f32   dlo = upsample^T(dy) for H = 2h, W = 2w, in gather form (no atomics); d(cur) = dy
 */
int pd_upsample_add_nhwc_f32(const float *lo, int64_t lo_batch_stride, const float *cur, float *y, int B, int h, int w, int H, int W, int C, void *stream);
/* the same with amax[B * H * W] = absolute maximum over the channels of every output pixel (C == 256 only) */
int pd_upsample_add_amax_nhwc_f32(const float *lo, int64_t lo_batch_stride, const float *cur, float *y, float *amax, int B, int h, int w, int H, int W, int C,
                                  void *stream);
int pd_upsample2x_bwd_nhwc_f32(const float *dy, float *dlo, int B, int h, int w, int C, void *stream);

/*
 * F.interpolate(x, size=(heights[i], widths[i]), mode="bilinear", align_corners=False) of a channels-last fp32 map x [B, H, W, C] for up to
 * PD_RESIZE_MAX sizes in one launch; outs[i] [B, heights[i] * widths[i], C] in out_dtype (PD_F32 / PD_BF16: rounded on the way out).  The
 * decoder's pooled mask features (reference mask2former_transformer_decoder.py:452, one resize per level).  heights / widths / outs: HOST arrays.
 */
#define PD_RESIZE_MAX 4
int pd_resize_bilinear_nhwc_f32(const float *x, int B, int H, int W, int C, const int *heights, const int *widths, void *const *outs, int count,
                                int out_dtype, void *stream);


/* q = a + b (fp32 [rows, cols], cols % 4 == 0), optionally a_copy = a, and the absolute maxima of every row of a and of q (the row scales of
 * pd_gemm_tn_f16x2, pd_gemm.h) in ONE pass: the entry of the pixel decoder's encoder (reference msdeformattn.py:120 `with_pos_embed(src, pos)`:
 * query = src + pos feeds the offset / weight projections, src the value projection). */
int pd_add_rows_amax_f32(const float *a, const float *b, float *q, float *a_copy, float *a_amax, float *q_amax, int rows, int cols, void *stream);

/* out1 = (a + b) + c, out2 = d + c over n fp32 elements (n % 4 == 0, 16-byte aligned): the three elementwise sums that end the encoder's backward
 * (d(src) = its three terms, d(pos) = accumulator + last term; reference msdeformattn.py:41-42 `with_pos_embed` backward) as one launch */
int pd_sum3_sum2_f32(const float *a, const float *b, const float *c, const float *d, float *out1, float *out2, int64_t n, void *stream);

/*
 * Linear layer with few output columns (K <= 8): the decoder's class head over all prediction heads' queries (reference
 * mask2former_transformer_decoder.py:223 class_embed = nn.Linear(hidden_dim, num_classes + 1), applied at :446).
 *   forward   y [R, K] fp32 = x [R, C] (bf16) w [K, C]^T + b [K]      (w, b in wb_dtype: PD_F32 / PD_BF16; b nullable; C % 4 == 0)
 *   backward  dx_out [R, C] (dx_dtype; nullable) = dx_in (bf16, nullable: the gradient another consumer of x already produced) + dy [R, K] w;
 *             dw [K, C] (w_dtype) = dy^T x;  db [K] (b_dtype; nullable) = column sums of dy.  `partial`: pd_skinny_linear_partial_floats(R, C, K)
 *             floats — per-workgroup partial sums added in workgroup order (deterministic).  fp32 products and sums.
 */
int pd_skinny_linear_fwd(const void *x_bf16, const void *w, const void *b, int wb_dtype, float *y, int R, int C, int K, void *stream);
int64_t pd_skinny_linear_partial_floats(int R, int C, int K);
int pd_skinny_linear_bwd(const void *x_bf16, const void *w, int w_dtype, const float *dy, const void *dx_in_bf16, void *dx_out, int dx_dtype,
                         float *partial, void *dw, void *db, int b_dtype, int R, int C, int K, void *stream);

/*
 * dst_i [batch, cols, rows] (contiguous) = src_i [batch, rows, cols]^T, fp32, for up to PD_TRANSPOSE_MAX problems in one launch.  src_i may
 * be strided: src_batch_stride / src_row_stride in floats (unit column stride) — the layers' weights where they lie in the flat parameter
 * buffer.  The transposed weight stacks of the fp32 encoder's input-gradient GEMMs (dX = dY W needs W^T as the [N, K] operand).
 */
#define PD_TRANSPOSE_MAX 8
typedef struct {
  const void *src;
  void *dst;
  int64_t src_batch_stride, src_row_stride;
  int32_t batch, rows, cols, reserved;
} PdTransposeProblem;
int pd_transpose_batched_f32(const PdTransposeProblem *problems, int count, void *stream);

/* `count` (<= PD_COPY_MAX_SEGS) dense byte ranges dst_i <- src_i in ONE launch (concatenations of slices of several tensors). */
#define PD_COPY_MAX_SEGS 48
typedef struct PdCopySeg {
  const void *src;
  void *dst;
  int64_t bytes;
} PdCopySeg;
int pd_copy_segments(const PdCopySeg *segs, int count, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PD_ROWWISE_H */
