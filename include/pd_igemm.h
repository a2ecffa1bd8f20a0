/*
 * pd_igemm.h — C-ABI of the bf16 implicit-GEMM family of libpd_hip.so (csrc/igemm_bf16.hip): ONE kernel skeleton for
 *
 *   (a) the ResNet-50 bottleneck convolutions forward, with the frozen-BN affine, the residual add and the ReLU in the epilogue
 *       (detectron2 0.6 BottleneckBlock.forward, selected by configs/mask2former/coco/instance-segmentation/
 *       Base-COCO-InstanceSegmentation.yaml:2-15) — what the reference gets from cuDNN + separate
 *       normalisation / add / activation kernels;
 *   (b) their input gradients (transposed convolution), with the gradient arriving over the other branch added and the ReLU mask
 *       of the layer below applied in the epilogue (autograd's ConvolutionBackward + ThresholdBackward + AddBackward);
 *   (c) every nn.Linear of the Swin backbones over the stage's tokens, forward and input gradient: qkv / proj
 *       (part_distillation/modeling/backbone/swin.py:127-129), Mlp fc1 + GELU / fc2 (:34-36), PatchMerging.reduction (:312), with
 *       bias, exact-erf GELU and GELU' in the epilogue — a Linear is the 1 x 1 case of (a) / (b);
 *   (d) the decoder's key / value projections over the memory tokens (mask2former_transformer_decoder.py:102-114).
 *
 * GEMM view:  out[m][n] = epilogue( sum_{tap, c} src[pixel(m, tap)][c] * w[n][tap][c] ),  m over the result grid's pixels (all
 * images), n over output channels.  src rows are gathered (im2col is never formed): forward  pixel = (oy*stride + dy - pad, ...),
 * input gradient  pixel = ((iy + pad - dy) / stride, ...) where divisible.  bf16 operands, fp32 accumulation on
 * v_mfma_f32_32x32x16_bf16, ONE rounding to bf16 at the end.
 *
 * Layouts (device pointers, 16-byte aligned; bf16 = uint16 storage):
 *   src / out / res / gate   NHWC rows  [batch][h][w][channels]  (a token matrix [tokens][channels] is the 1 x 1 case: h = tokens, w = 1)
 *   w                        [n][taps][src channels]   (torch channels_last storage of a conv filter; nn.Linear.weight as it lies);
 *                            for an input gradient pass the [ci][taps][co] transpose
 * Restrictions: src channels % 64 == 0, n % 64 == 0, k in {1, 3}, stride in {1, 2}, pad == k / 2; PD_ERR_INVALID_ARG otherwise.
 */
#ifndef PD_IGEMM_H
#define PD_IGEMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PD_IG_ACT_NONE = 0, PD_IG_ACT_RELU = 1, PD_IG_ACT_GELU = 2 };
enum { PD_IG_GATE_NONE = 0, PD_IG_GATE_RELU = 1, PD_IG_GATE_GELU = 2 };    /* out *= (gate > 0)  |  out *= gelu'(gate) */
enum { PD_IG_RES_DENSE = 0, PD_IG_RES_UP2 = 1 };   /* res indexed like out | res is [batch][ho/2][wo/2][n]: added at even (y, x) only */

typedef struct PdIgemm {
  const void *src, *w;
  const float *scale, *bias;          /* fp32 [n], nullable: acc * scale + bias */
  const void *res;                    /* bf16, nullable: + res (before act / gate), indexed per res_mode */
  const void *res2;                   /* bf16 [m][n], nullable: a second, dense addend (a gradient arriving from outside the backbone) */
  const void *gate;                   /* bf16 [m][n], nullable: see gate_mode (applied last) */
  void *out;                          /* bf16 [m][n] */
  void *out_pre;                      /* bf16 [m][n], nullable: the value BEFORE act (fc1's pre-activation for GELU') */
  int32_t batch, hs, ws, cs;          /* source grid and channels */
  int32_t ho, wo, n;                  /* result grid and channels */
  int32_t k, stride, pad;
  int32_t dgrad;                      /* 0: forward gather, 1: input-gradient gather */
  int32_t act, gate_mode, res_mode;
  int32_t bias_bf16;                  /* != 0: `bias` points to bf16 values (an nn.Linear bias kept in 16 bits), not fp32 */
  int32_t out_col_slab;               /* S != 0 (a multiple of 128 dividing n): `out` is [n / S][m][S] — columns [j S, (j + 1) S) of the product form their own
                                         dense [m][S] matrix (several Linears over the same rows as ONE product, each result usable on its own);
                                         res / res2 / gate / out_pre must be null */
} PdIgemm;

/* workspace the split-K schedule of this problem needs: fp32 partial tiles + one ticket word per tile (0 when it runs unsplit).
 * The tickets must be ZERO before the first launch that uses them; the kernel leaves them zero. */
int64_t pd_igemm_bf16_workspace_bytes(const PdIgemm *p);
/* workspace: nullable when pd_igemm_bf16_workspace_bytes(p) == 0; layout [tickets (4 KB)][fp32 slabs] */
int pd_igemm_bf16(const PdIgemm *p, void *workspace, int64_t workspace_bytes, void *stream);
int pd_igemm_bf16_supported(int cs, int n, int k, int stride, int pad);

/* `count` problems enqueued back to back on `stream` by ONE call (they run in order and share the workspace, which must hold the
 * largest need: pd_igemm_bf16_seq_workspace_bytes).  The list is host memory and is read before the call returns. */
int64_t pd_igemm_bf16_seq_workspace_bytes(const PdIgemm *list, int count);
int pd_igemm_bf16_seq(const PdIgemm *list, int count, void *workspace, int64_t workspace_bytes, void *stream);

/* dst[ci][tap][co] = src[co][tap][ci] (* scale[co]) for `count` filters in ONE launch: the [ci][taps][co] operands of the input gradients, rebuilt
 * from the updated weights once per step.  co % 64 == ci % 64 == 0.  table_host_pinned / table_device: caller-provided staging of
 * pd_filter_transpose_table_bytes(count) bytes each; the pinned one must stay untouched until the asynchronous copy has executed. */
typedef struct PdFilterTranspose {
  const void *src;
  void *dst;
  const float *scale;                 /* nullable: fp32 [co] folded into the transposed copy (frozen-BN scale) */
  int32_t co, taps, ci;
} PdFilterTranspose;
int64_t pd_filter_transpose_table_bytes(int count);
int pd_filter_transpose_grouped(const PdFilterTranspose *descs, int count, void *table_host_pinned, void *table_device, void *stream);

/* ---- weight gradient of a Linear / 1 x 1 convolution over m rows (csrc/wgrad_bf16.hip):
 *   dw[n][k] = row_scale[n] * sum_m dy[m][n] * x[m][k]   (bf16 operands, fp32 accumulation; result bf16, or fp32 with dw_f32; row stride ldw elements)
 *   db[n]   += sum_m dy[m][n]                            (fp32, nullable; the caller zero-fills or accumulates)
 * What autograd's MmBackward / ConvolutionBackward(weight) computes for nn.Linear (swin.py:34-36, 127-129, 312) under bf16 autocast.
 * n, k, ldy, ldx multiples of 8; operands 16-byte aligned.  Two launches when the rows are cut into slices (fp32 partial tiles, then their sum in
 * slice order: deterministic).  workspace: pd_wgrad_bf16_workspace_bytes(p) bytes, no initial state; its first 16 KB are never touched, so the
 * buffer that holds pd_igemm_bf16's tickets can be passed. */
typedef struct PdWgrad {
  const void *dy, *x;
  void *dw;
  float *db;                          /* nullable */
  const float *row_scale;             /* nullable */
  int32_t m, n, k, ldy, ldx, ldw;
  int32_t dw_f32;                     /* != 0: dw points to fp32 (the gradient of an fp32 master weight: no 16-bit rounding, no cast pass) */
} PdWgrad;
int64_t pd_wgrad_bf16_workspace_bytes(const PdWgrad *p);
int pd_wgrad_bf16(const PdWgrad *p, void *workspace, int64_t workspace_bytes, void *stream);
/* `count` <= PD_WGRAD_SEQ_MAX problems by one call (the four Linears of a Swin block): their main launches back to back, then ONE launch for
 * all their slice sums.  The workspace holds every problem's slabs side by side: pd_wgrad_bf16_seq_workspace_bytes.  The list is host memory,
 * read before the call returns. */
#define PD_WGRAD_SEQ_MAX 8
int64_t pd_wgrad_bf16_seq_workspace_bytes(const PdWgrad *list, int count);
int pd_wgrad_bf16_seq(const PdWgrad *list, int count, void *workspace, int64_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PD_IGEMM_H */
