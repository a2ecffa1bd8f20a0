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

/* y = act(x * scale[c] + bias[c] (+ residual)); x, residual (nullable), y: bf16, channels-last (c fastest,
 * `channels` % 8 == 0, n % channels == 0); scale, bias: fp32 [channels]; relu != 0 applies max(.,0). y may alias x. */
int pd_affine_act_fwd_bf16(const void *x, const void *residual, const float *scale, const float *bias, void *y,
                           int64_t n, int channels, int relu, void *stream);

/* gx = gy * [y > 0 if relu] * scale[c];  gres (nullable) = gy * [y > 0 if relu].  gx may alias gy. */
int pd_affine_act_bwd_bf16(const void *gy, const void *y, const float *scale, void *gx, void *gres, int64_t n,
                           int channels, int relu, void *stream);

/*
 * For block b in [block_begin, block_end): copy `blk_len[b]` elements from tensor blk_tensor[b] starting at element
 * blk_start[b] into dst[blk_dst[b] ...] as fp32.  src_ptrs[t] is the device address of tensor t (0 => zeros),
 * src_is_bf16[t] its element type.  *sumsq (float64, nullable) += sum of squares of everything copied.
 */
int pd_multi_gather_sumsq(const int64_t *src_ptrs, const int32_t *src_is_bf16, const int32_t *blk_tensor,
                          const int64_t *blk_start, const int64_t *blk_dst, const int32_t *blk_len, float *dst,
                          double *sumsq, int block_begin, int block_end, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PD_FUSED_H */
