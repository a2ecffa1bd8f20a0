/*
 * pd_conv.h — C-ABI of the bf16 NHWC convolutions of the ResNet backbone in libpd_hip.so (csrc/conv_bf16.hip).
 *
 * What they replace.  The reference's R50 backbone is detectron2 0.6's `build_resnet_backbone` (un-vendored dependency;
 * instantiated by the reference through part_distillation/config.py + the r50 YAMLs under configs/ `MODEL.BACKBONE.NAME`, SURVEY
 * Appendix D): every `BottleneckBlock` runs  Conv2d(1x1) -> FrozenBN -> ReLU,  Conv2d(3x3, stride s) -> FrozenBN -> ReLU,
 * Conv2d(1x1) -> FrozenBN, `out += shortcut`, ReLU  (detectron2/modeling/backbone/resnet.py BottleneckBlock.forward), each
 * Conv2d a cuDNN call under fp16/bf16 autocast followed by separate normalisation / add / activation kernels.  Here one
 * launch is the convolution AND its frozen-BN affine, residual add and ReLU; the input-gradient launch is the transposed
 * convolution AND the sum with the gradient arriving over the block's shortcut.
 *
 * Layouts (device pointers, 16-byte aligned; bf16 = uint16 storage; `stream` = hipStream_t):
 *   activations  NHWC  [batch][h][w][channels]                 (torch channels_last storage of an NCHW-shaped tensor)
 *   filter       [co][k][k][ci]                                (torch channels_last storage of the [co][ci][k][k] weight)
 *   filter^T     [ci][k][k][co]                                (same taps, channel roles swapped: pd_conv_bf16_dgrad's operand)
 * Supported: ci % 64 == 0, co % 64 == 0, k in {1, 3}, stride in {1, 2}, pad == k / 2, groups = dilation = 1 — every
 * convolution of R50/R101 except the 7x7 stem (ci = 3).  pd_conv_bf16_supported() says so; the functions return
 * PD_ERR_INVALID_ARG (pd_msda.h) for anything else.  fp32 accumulation over k*k*ci on v_mfma_f32_32x32x16_bf16, ONE rounding
 * to bf16 at the end (the unfused chain rounds the convolution result before the affine).
 */
#ifndef PD_CONV_H
#define PD_CONV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int pd_conv_bf16_supported(int ci, int co, int k, int stride, int pad);

/* y[b,oy,ox,:] = act( conv(x, w)[b,oy,ox,:] * scale[:] + bias[:] (+ residual[b,oy,ox,:]) ),  act = ReLU when relu != 0.
 * scale, bias: fp32 [co], nullable (identity / zero); residual: bf16 like y, nullable.  ho = (hi + 2 pad - k) / stride + 1. */
int pd_conv_bf16_fwd(const void *x, const void *w, const float *scale, const float *bias, const void *residual, void *y, int batch,
                     int hi, int wi, int ci, int ho, int wo, int co, int k, int stride, int pad, int relu, void *stream);

/* dx[b,iy,ix,:] = sum over taps and co of dz[b,(iy+pad-dy)/stride,(ix+pad-dx)/stride,:] . wt[:,dy,dx,:]  (+ addend[b,iy,ix,:])
 * — the gradient of pd_conv_bf16_fwd's convolution with respect to x, given dz = the gradient at the convolution's output.
 * wt = the [ci][k][k][co] transpose of the filter; addend: bf16 like dx, nullable; dx may alias addend. */
int pd_conv_bf16_dgrad(const void *dz, const void *wt, const void *addend, void *dx, int batch, int hi, int wi, int ci, int ho, int wo,
                       int co, int k, int stride, int pad, void *stream);

/* dw[co][dy][dx][ci] = sum over b, oy, ox of dz[b,oy,ox,co] * x[b, oy*stride + dy - pad, ox*stride + dx - pad, ci]   (bf16 result,
 * fp32 accumulation): the gradient of the convolution with respect to its filter, in the filter's own [co][k][k][ci] layout.
 * Any k, stride, pad (the 7x7 stem excepted only by ci % 8 == 0); ci % 8 == 0, co % 8 == 0.  workspace: fp32, at least
 * pd_conv_bf16_wgrad_workspace_floats(...) elements, 16-byte aligned, reusable by the next call on the same stream. */
int64_t pd_conv_bf16_wgrad_workspace_floats(int batch, int ho, int wo, int ci, int co, int k);
int pd_conv_bf16_wgrad(const void *dz, const void *x, void *dw, float *workspace, int64_t workspace_floats, int batch, int hi, int wi,
                       int ci, int ho, int wo, int co, int k, int stride, int pad, void *stream);

/* The filter gradients of MANY convolutions in one go (a backbone's, deferred to the end of its backward pass: a single layer's
 * is a 20-40 us launch that cannot fill the chip).  descs: host array; table_host_pinned / table_device: caller-provided staging
 * of pd_conv_bf16_wgrad_grouped_table_bytes(count) bytes each (the function fills the pinned one, copies it with an asynchronous
 * memcpy on `stream` and launches at most four kernel pairs, one per output-tile shape); the pinned buffer must stay untouched
 * until that copy has executed.  workspace >= pd_conv_bf16_wgrad_grouped_workspace_floats(descs, count).  count <= 256. */
typedef struct PdConvWgradDesc {
  const void *dz, *x;
  void *dw;
  float *db;     /* nullable: fp32 [co], += sum over pixels of dz (the bias gradient of a convolution / Linear with bias) */
  int32_t batch, hi, wi, ci, ho, wo, co, k, stride, pad;
  const float *scale;   /* nullable: fp32 [co], dw[co][...] is multiplied by scale[co] (the frozen-BN scale when dz is the gradient
                           BEHIND the affine: pd_igemm.h's fused backbone) */
} PdConvWgradDesc;
int64_t pd_conv_bf16_wgrad_grouped_table_bytes(int count);
int64_t pd_conv_bf16_wgrad_grouped_workspace_floats(const PdConvWgradDesc *descs, int count);
int pd_conv_bf16_wgrad_grouped(const PdConvWgradDesc *descs, int count, void *table_host_pinned, void *table_device, float *workspace,
                               int64_t workspace_floats, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PD_CONV_H */
