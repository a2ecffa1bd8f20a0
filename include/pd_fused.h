/*
 * pd_fused.h — C-ABI of the fused bandwidth-bound kernels of libpd_hip.so that replace chains of elementwise
 * PyTorch launches on the training step (each a full HBM round trip in the reference's eager execution):
 *
 *   pd_affine_act_{fwd,bwd}_bf16   frozen-BN affine (+ residual add) (+ ReLU) after every backbone convolution:
 *                                  detectron2 Conv2d(norm=FrozenBN) -> F.relu_ / `out += shortcut; relu_` of
 *                                  BottleneckBlock (the reference takes them from detectron2 0.6).
 *   pd_multi_gather_sumsq          collects every parameter gradient autograd produced into the flat fp32 gradient
 *                                  buffer (bf16 -> fp32 where needed) and accumulates the global sum of squares for
 *                                  clip_grad_norm_ in the same pass (reference base_trainer.py:127-131).
 * Device pointers; `stream` = hipStream_t; 0 or negative PD_ERR_* (pd_msda.h).
 */
#ifndef PD_FUSED_H
#define PD_FUSED_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 x 3, stride 2, pad 1 max pooling of the R50 stem on bf16 NHWC maps [B,H,W,C], C % 8 == 0 (detectron2 BasicStem: F.max_pool2d):
 * y [B,OH,OW,C] with OH = (H - 1) / 2 + 1; argmax: one byte per output element (8 per 64-bit word, same element order as y) = the
 * window position 0..8 of the maximum, the FIRST one in row-major window order (ATen's tie rule).  The backward gathers: every
 * input pixel adds the gradients of the <= 4 windows whose recorded maximum it is. */
int pd_maxpool3s2_fwd_bf16(const void *x, void *y, void *argmax, int B, int H, int W, int C, void *stream);
int pd_maxpool3s2_bwd_bf16(const void *dy, const void *argmax, void *dx, int B, int H, int W, int C, void *stream);

/* y = act(x * scale[c] + bias[c] (+ residual)); x, residual (nullable), y: bf16, channels-last (c fastest,
 * `channels` % 8 == 0, n % channels == 0); scale, bias: fp32 [channels]; relu != 0 applies max(.,0). y may alias x. */
int pd_affine_act_fwd_bf16(const void *x, const void *residual, const float *scale, const float *bias, void *y,
                           int64_t n, int channels, int relu, void *stream);

/* gx = gy * [y > 0 if relu] * scale[c];  gres (nullable) = gy * [y > 0 if relu].  gx may alias gy. */
int pd_affine_act_bwd_bf16(const void *gy, const void *y, const float *scale, void *gx, void *gres, int64_t n,
                           int channels, int relu, void *stream);
/* The same with the incoming gradient given as the SUM of two tensors (gy2 nullable): an activation that feeds two consumers (a
 * bottleneck block's output: the next block's first convolution and its shortcut) gets one gradient from each, and the add that
 * autograd would launch between them and this kernel is done here instead. */
int pd_affine_act_bwd2_bf16(const void *gy, const void *gy2, const void *y, const float *scale, void *gx, void *gres, int64_t n,
                            int channels, int relu, void *stream);

/*
 * For block b in [block_begin, block_end): copy `blk_len[b]` elements from tensor blk_tensor[b] starting at element
 * blk_start[b] into dst[blk_dst[b] ...] as fp32.  src_ptrs[t] is the device address of tensor t (0 => zeros),
 * src_is_bf16[t] its element type.  *sumsq (float64, nullable) += sum of squares of everything copied.
 */
int pd_multi_gather_sumsq(const int64_t *src_ptrs, const int32_t *src_is_bf16, const int32_t *blk_tensor,
                          const int64_t *blk_start, const int64_t *blk_dst, const int32_t *blk_len, float *dst,
                          double *sumsq, int block_begin, int block_end, void *stream);

/*
 * Channels-last fp32 GroupNorm (+ReLU) building blocks for the pixel decoder's Conv2d -> GroupNorm(32) (-> ReLU) layers
 * (reference pixel_decoder/msdeformattn.py:236-239, 270-285: nn.GroupNorm(32, conv_dim) / get_norm("GN")).  Maps are
 * [N, P pixels, C channels] with C the fastest dimension (C = 4*2^k <= 256).  The caller combines the per-(n, c)
 * sums into group statistics / coefficients (O(N*C) work) between the calls:
 *   pd_nc_sums_f32   mode 0: out[n,c] = {sum_p x, sum_p x^2};  mode 1: {sum_p g*(x*a[n,c]+b[n,c]), sum_p g} with
 *                    g = dy, or dy*[y > 0] when relu; `out` (float64 [N, C, 2]) is zeroed by the library
 *   pd_nc_affine_f32   y = x*a[n,c] + b[n,c], then ReLU if relu
 *   pd_nc_affine2_f32  dx = g*a[n,c] + x*p[n,c] + r[n,c]
 */
int pd_nc_sums_f32(const float *x, const float *dy, const float *y, const float *a, const float *b, double *out, int N, int P,
                   int C, int mode, int relu, void *stream);
int pd_nc_affine_f32(const float *x, const float *a, const float *b, float *y, int N, int P, int C, int relu, void *stream);
/* pd_nc_affine_f32 / pd_nc_affine2_f32 that also write the absolute maximum over the channels of every output pixel, amax[N * P]
 * (C == 256 only: one wavefront = one pixel) — the row-scaling input of pd_gemm_tn_f16x2 / pd_conv3x3_nhwc_f16x2 (pd_gemm.h) for the
 * convolution that consumes the map, produced where the pixel is in registers anyway. */
int pd_nc_affine_amax_f32(const float *x, const float *a, const float *b, float *y, float *amax, int N, int P, int C, int relu, void *stream);
int pd_nc_affine2_amax_f32(const float *dy, const float *x, const float *y, const float *a, const float *p, const float *r, float *dx,
                           float *amax, int N, int P, int C, int relu, void *stream);
int pd_nc_affine2_f32(const float *dy, const float *x, const float *y, const float *a, const float *p, const float *r, float *dx,
                      int N, int P, int C, int relu, void *stream);

/*
 * The O(N*C) algebra between those passes as one launch each (instead of ~15 / ~20 tiny elementwise launches):
 *   pd_gn_coeffs_fwd   sums (float64 [N,C,2] from pd_nc_sums mode 0) -> group mean / rstd (biased variance over the
 *                      P*(C/G) elements of a group, clamped at 0) as per-(n,c) arrays, a = rstd*weight,
 *                      b = bias - mean*a, xb = -mean*rstd
 *   pd_gn_coeffs_bwd   sums (mode 1) -> a, p, r of pd_nc_affine2 and the weight / bias gradients gw[C], gb[C]
 */
int pd_gn_coeffs_fwd(const double *sums, const float *weight, const float *bias, int N, int C, int G, int P, float eps, float *a,
                     float *b, float *mean_c, float *rstd_c, float *xb, void *stream);
int pd_gn_coeffs_bwd(const double *sums, const float *weight, const float *mean_c, const float *rstd_c, int N, int C, int G, int P,
                     float *a, float *pcoef, float *rcoef, float *gw, float *gb, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PD_FUSED_H */
