/*
 * pd_stem.h — the ResNet stem convolution (7 x 7, stride 2, padding 3, 3 -> 64 channels) of libpd_hip.so, forward and filter gradient.
 *
 * Replaces, on the hot path, what the reference gets from detectron2 0.6 `BasicStem` (un-vendored; selected by
 * the reference's configs/mask2former/coco/instance-segmentation/Base-COCO-InstanceSegmentation.yaml:2-15 `build_resnet_backbone`,
 * STEM_OUT_CHANNELS 64, NORM FrozenBN):  conv1 = Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False, norm=FrozenBN) followed
 * by ReLU (and a 3 x 3 / 2 max pooling, pd_maxpool3s2_*_bf16 in pd_fused.h).  Rounds 1-4 ran this layer on MIOpen (forward + filter
 * gradient) with a separate frozen-BN / ReLU pass; it was the last library convolution of BASELINE config 2.
 *
 * Layouts (all device pointers, plain sizes; `stream` is a hipStream_t):
 *   x       [B, H, W, 3]   NHWC image batch, fp32 (x_is_f32 = 1: rounded to bf16 as it is staged — what autocast's cast does) or bf16
 *   w       [64, 7, 7, 3]  bf16 filter, channels-last (the layout of a torch [64, 3, 7, 7] channels_last tensor)
 *   scale, bias [64]       fp32 frozen-BatchNorm affine (weight / sqrt(var + eps), bias - mean * scale)
 *   y       [B, Ho, Wo, 64] bf16, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1:  y = act(conv(x, w) * scale + bias), act = ReLU if relu
 * Arithmetic: bf16 products, fp32 accumulation on v_mfma_f32_32x32x16_bf16 (K = 147 padded to 176), fp32 affine, one rounding to bf16.
 *
 * Filter gradient (the image needs no gradient, so there is no input gradient):
 *   gy      [B, Ho, Wo, 64] bf16 gradient of y;  y as written by the forward (NULL when relu = 0)
 *   dw      [64, 7, 7, 3]  bf16 (dw_is_f32 = 0) or fp32:  dw[o] = scale[o] * sum_{pixels} (y > 0 ? gy : 0)[.., o] * x[window]
 *   workspace: pd_stem_wgrad_workspace_floats() floats (partial sums per workgroup, summed in workgroup order: deterministic)
 *
 * Errors: 0 on success, negative PD_ERR_* with pd_last_error(); nothing is launched on error.
 */
#ifndef PD_STEM_H
#define PD_STEM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int pd_stem7x7_fwd(const void *x, int x_is_f32, const void *w_bf16, const float *scale, const float *bias, void *y_bf16, int B, int H, int W, int relu,
                   void *stream);
int64_t pd_stem_wgrad_workspace_floats(void);
int pd_stem7x7_wgrad(const void *x, int x_is_f32, const void *gy_bf16, const void *y_bf16, const float *scale, void *dw, int dw_is_f32,
                     float *workspace, int B, int H, int W, int relu, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PD_STEM_H */
