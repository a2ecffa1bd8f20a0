/*
 * pd_swin.h — C-ABI of the token-row kernels of the fused Swin block of libpd_hip.so.
 *
 * A Swin block is  x' = x + DropPath(proj(W-MSA(window_partition(shift(pad(LN1(x)))))));  out = x' + DropPath(MLP(LN2(x')))
 *   reference modeling/backbone/swin.py:239-299 (SwinTransformerBlock.forward), window_partition / reverse :73-100,
 *   F.pad :254-258, torch.roll :261-266, 276-289, DropPath :35-51.
 * pad + roll + window_partition (and their inverses) are a fixed permutation of token rows plus zero rows, and the
 * residual adds / DropPath scales are row-wise, so the whole glue between the GEMMs of a block is two calls of ONE
 * kernel each way:
 *
 *   pd_swin_ln_fwd    s[i] = x[i] + rscale[img] * r[rrow(i)]                (skipped when r == NULL: s = x)
 *                     y[yrow(i)] = bf16(LayerNorm(s[i]) * gamma + beta),  mean / rstd[i] saved;  y[zero rows] = 0
 *   pd_swin_ln_bwd    ds[i] = dsup[i] + LayerNorm'(dy[yrow(i)]);  dr[rrow(i)] = bf16(rscale[img] * ds[i]);  dr[zero rows] = 0
 *                     dgamma / dbeta += column sums (atomics into fp32 [C]: the caller zero-fills)
 *
 * Rows i = img * L + t (t < L tokens of an image).  A row map (int32 [L], shared by the images) sends token t to a row
 * of a tensor with `rows_per_image` rows per image: yrow(i) = img * y_rows + ymap[t]; NULL = identity (then *_rows = L).
 * LN1 of a block: y window-major (ymap = token -> window slot, zero rows = the padded slots), r = the previous block's
 * MLP output (identity map).  LN2: r = the proj output, window-major (rmap = token -> slot), y token-major.
 * x, s, dsup, ds: fp32 [R, C] (16-byte aligned);  r, y, dy, dr: bf16 (8-byte aligned);  C % 64 == 0, C <= 1536 (every Swin-T/S/B/L stage width);
 * rscale: fp32 [images] (DropPath keep mask / keep_prob) or NULL = 1.  `stream` = hipStream_t; returns 0 or PD_ERR_*.
 */
#ifndef PD_SWIN_H
#define PD_SWIN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* y_q / y_s (nullable, together): y again as an MX-fp8 operand (include/pd_mx8.h: [rows][C] fp8 of q_format, [rows][C / 32] E8M0 bytes,
 * rows as y) — quantised from the bf16 values y holds, zero rows included */
int pd_swin_ln_fwd(const float *x, const void *r, const int32_t *rmap, int r_rows, const float *rscale, const float *gamma,
                   const float *beta, float eps, float *s, void *y, const int32_t *ymap, int y_rows, const int32_t *zero_rows,
                   int n_zero, float *mean, float *rstd, int images, int L, int C, void *y_q, void *y_s, int q_format, void *stream);

/* dr_q / dr_s (nullable, together; need dr): dr again as an MX-fp8 operand, rows as dr.
 * n_rep / rep_stride: the column sums go to copy (workgroup % n_rep) of dgamma / dbeta, copies rep_stride floats apart (all zero-filled by
 * the caller, who sums them): ~500 workgroups adding into the same 2 C addresses serialise; n_rep = 1 is the single accumulator */
int pd_swin_ln_bwd(const void *dy, const int32_t *ymap, int y_rows, const float *dsup, const float *s, const float *mean,
                   const float *rstd, const float *gamma, float *ds, void *dr, const int32_t *rmap, int r_rows,
                   const float *rscale, const int32_t *zero_rows, int n_zero, float *dgamma, float *dbeta, int images, int L,
                   int C, void *dr_q, void *dr_s, int q_format, int n_rep, int64_t rep_stride, void *stream);

/* Plain fp32 LayerNorm over `rows` token rows of C channels (C % 4 == 0, C <= 3072), forward and backward: the Swin backbone's per-stage
 * OUTPUT norms (reference modeling/backbone/swin.py:675-680), PatchMerging's norm over 4 C channels (:339) and the patch embedding's (:565),
 * which stay fp32 under autocast.  mean / rstd [rows] are saved by the forward;
 * the backward ACCUMULATES into dgamma / dbeta (fp32 [C], zero-filled by the caller) and overwrites dx. */
int pd_layernorm_rows_f32_fwd(const float *x, const float *gamma, const float *beta, float eps, float *y, float *mean, float *rstd, int64_t rows,
                              int C, void *stream);
int pd_layernorm_rows_f32_bwd(const float *dy, const float *x, const float *mean, const float *rstd, const float *gamma, float *dx, float *dgamma,
                              float *dbeta, int64_t rows, int C, void *stream);

/* Patch merging's gather + LayerNorm (reference modeling/backbone/swin.py:325-339, `x = cat([x0, x1, x2, x3], -1); x = self.norm(x)`): x fp32 [B, H, W, C]
 * (H, W even) -> y bf16 [B (H/2) (W/2), 4 C], channel block 2 cp + rp of a merged row = pixel (2 i + rp, 2 j + cp) — the order of
 * modeling/backbone/swin.py's permuted view — normalised over the 4 C channels in fp32 (mean / rstd [rows] saved).  The backward overwrites dx fp32
 * [B, H, W, C] (every pixel belongs to one merged row) and ACCUMULATES into dgamma / dbeta (fp32 [4 C], zero-filled by the caller).  C % 4 == 0, 4 C <= 3072. */
int pd_swin_merge_ln_fwd(const float *x, const float *gamma, const float *beta, float eps, void *y, float *mean, float *rstd, int B, int H, int W, int C,
                         void *stream);
int pd_swin_merge_ln_bwd(const void *dy, const float *x, const float *mean, const float *rstd, const float *gamma, float *dx, float *dgamma, float *dbeta,
                         int B, int H, int W, int C, void *stream);

/* The end of a fused Swin stage with its output norm (reference swin.py:234-236 + :675-680): s = cur + rscale[row / L] * r (cur fp32, r bf16 — the last block's MLP
 * output —, rscale fp32 [images] or NULL = 1: DropPath), y = LayerNorm(s) * gamma + beta, both fp32 [rows, C]; mean / rstd [rows] saved.  Backward: dy = gradient of
 * y, dsum = gradient of s from the rest of the network (or NULL); dsup = dsum + LayerNorm'(dy) (fp32: the stream's gradient), df = bf16(rscale * dsup) (the MLP
 * output's gradient); ACCUMULATES into dgamma / dbeta (zero-filled by the caller).  C % 4 == 0, C <= 3072. */
int pd_swin_tail_ln_fwd(const float *cur, const void *r, const float *rscale, int L, const float *gamma, const float *beta, float eps, float *s, float *y,
                        float *mean, float *rstd, int64_t rows, int C, void *stream);
int pd_swin_tail_ln_bwd(const float *dy, const float *dsum, const float *s, const float *mean, const float *rstd, const float *gamma, const float *rscale, int L,
                        float *dsup, void *df, float *dgamma, float *dbeta, int64_t rows, int C, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PD_SWIN_H */
