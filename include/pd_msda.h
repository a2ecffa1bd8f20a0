/*
 * pd_msda.h — C-ABI of the multi-scale deformable attention operator
 * (libpd_hip.so, built from partdistillation_amd/csrc/ for gfx950).
 *
 * These two entry points are exactly what the reference's pybind module
 * `MultiScaleDeformableAttention` binds for this path:
 *   reference part_distillation/modeling/pixel_decoder/ops/src/vision.cpp:19-22
 *     m.def("ms_deform_attn_forward",  &ms_deform_attn_forward, ...)
 *     m.def("ms_deform_attn_backward", &ms_deform_attn_backward, ...)
 *   dispatched by ops/src/ms_deform_attn.h:26-67 to
 *   ops/src/cuda/ms_deform_attn_cuda.cu:26-86 (forward) / :89-159 (backward).
 *
 * Plain pointers and sizes only — no torch types.  All pointers are DEVICE
 * pointers (including spatial_shapes / level_start_index, which the reference
 * also keeps on the device, ms_deform_attn_cuda.cu:41-42).  Tensors are
 * contiguous row-major:
 *   value              [batch, spatial_size, num_heads, channels]
 *   spatial_shapes     int64 [num_levels, 2]   (H_l, W_l)
 *   level_start_index  int64 [num_levels]
 *   sampling_loc       [batch, num_query, num_heads, num_levels, num_point, 2]  (x, y) in [0,1]
 *   attn_weight        [batch, num_query, num_heads, num_levels, num_point]
 *   output/grad_output [batch, num_query, num_heads*channels]
 * dtype: PD_F32 or PD_F64 (the reference dispatches float/double only,
 * ms_deform_attn_cuda.cu:70).  `stream` is a hipStream_t (the reference
 * launches on the current stream, ms_deform_attn_cuda.cu:71).
 *
 * Ownership: inputs are borrowed and never written.  The caller allocates the
 * outputs; the library overwrites `output` completely and zero-fills
 * grad_value / grad_sampling_loc / grad_attn_weight itself before accumulating
 * (the reference returns at::zeros-initialised tensors, .cu:60,127-129).
 *
 * Errors: return 0 on success; a negative PD_ERR_* otherwise, with a message
 * retrievable through pd_last_error() (the reference raises through
 * AT_ASSERTM, e.g. `batch % im2col_step_ == 0`, .cu:58).  Nothing is launched
 * when an error is returned.
 */
#ifndef PD_MSDA_H
#define PD_MSDA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PD_F32 = 0, PD_F64 = 1, PD_BF16 = 2 };

enum {
  PD_OK = 0,
  PD_ERR_INVALID_ARG = -1, /* null pointer, non-positive size, unknown dtype */
  PD_ERR_IM2COL_STEP = -2, /* batch % min(batch, im2col_step) != 0 (reference .cu:58) */
  PD_ERR_LAUNCH = -3       /* hipGetLastError() after launch was not hipSuccess */
};

/* replaces MSDA.ms_deform_attn_forward (vision.cpp:20, ms_deform_attn_cuda.cu:26-86) */
int pd_msda_forward(const void *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                    const void *sampling_loc, const void *attn_weight, void *output,
                    int batch, int spatial_size, int num_heads, int channels, int num_levels,
                    int num_query, int num_point, int im2col_step, int dtype, void *stream);

/* pd_msda_forward that also leaves the absolute maximum of every output row (all heads of a query) in row_amax[batch * num_query]
 * (ZERO-FILLED by the caller; atomic max) — the row-scaling input of pd_gemm_tn_f16x2 (pd_gemm.h) for the output projection that
 * reads the result.  fp32, channels = 32, 3 levels, 4 points only (the kernel of the hot path); not part of the reference's operator. */
int pd_msda_forward_amax(const void *value, const int64_t *spatial_shapes, const int64_t *level_start_index, const void *sampling_loc,
                         const void *attn_weight, void *output, float *row_amax, int batch, int spatial_size, int num_heads, int channels,
                         int num_levels, int num_query, int num_point, int im2col_step, int dtype, void *stream);

/* replaces MSDA.ms_deform_attn_backward (vision.cpp:21, ms_deform_attn_cuda.cu:89-159) */
int pd_msda_backward(const void *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                     const void *sampling_loc, const void *attn_weight, const void *grad_output,
                     void *grad_value, void *grad_sampling_loc, void *grad_attn_weight,
                     int batch, int spatial_size, int num_heads, int channels, int num_levels,
                     int num_query, int num_point, int im2col_step, int dtype, void *stream);

/* ---- the module path's fused form (round 5; NOT part of the reference's operator ABI, which is the two functions above) ----
 * MSDeformAttn.forward (reference ops/modules/ms_deform_attn.py:86-131) feeds the operator
 *   attention_weights = softmax(Linear_aw(query)) over the L P samples of a head      (:110-111)
 *   sampling_locations = reference_points + Linear_so(query) / (W_l, H_l)              (:108, :114-117)
 * The fused pair takes the RAW output of those two projections — one matrix `oa` [batch * num_query, ld_oa] fp32 with the
 * sampling offsets in columns [0, 2 M L P) ordered (head, level, point, xy) and the attention logits in columns [2 M L P, 3 M L P)
 * ordered (head, level, point), i.e. exactly the two Linear outputs side by side — and the 2-d reference points
 * `ref` [batch * num_query, L, 2], and performs both steps in the kernels' registers: no sampling_loc / attn_weight tensors, no
 * separate softmax / location kernels in either direction.
 *   forward : output [batch, num_query, M * 32]; stats [batch * num_query * M, 2] <- {max logit, 1 / sum exp} of every (query, head)
 *             (the backward re-forms a probability with one exponential); row_amax (optional, ZERO-FILLED by the caller) as
 *             pd_msda_forward_amax.
 *   backward: grad_value [batch, S, M, 32] (zero-filled by the library, as pd_msda_backward) and d_oa [batch * num_query, ld_doa] =
 *             the gradient of `oa` in the same column layout (every element written once): d offset = grad_sampling_loc / (W, H),
 *             d logit = a (g - sum_j a_j g_j) (sum_j a_j g_j = <grad_output, fwd_output>; the kernel sums its own per-level partials and does not read fwd_output);
 *             d_oa_amax (optional; the library zero-fills it) receives max |.| of every row of d_oa; scratch = batch * num_query * M * L
 *             floats of working memory (contents undefined afterwards).
 * Served geometry (pd_msda_fused_supported): fp32, 32 channels per head, 3 levels, 4 points, num_query == spatial_size (the
 * pixel decoder's encoder self-attention).  Other geometries: form loc / attn (pd_msda_prep_fwd, pd_rowwise.h) and call the
 * operator above.  Semantics of the sampling itself: ms_deform_im2col_cuda.cuh:38-164, 242-304, unchanged. */
int pd_msda_fused_supported(int batch, int spatial_size, int num_heads, int channels, int num_levels, int num_query, int num_point);
int pd_msda_fused_forward(const float *value, const int64_t *spatial_shapes, const int64_t *level_start_index, const float *oa, int ld_oa,
                          const float *ref, float *output, float *stats, float *row_amax, int batch, int spatial_size, int num_heads,
                          int channels, int num_levels, int num_query, int num_point, void *stream);
int pd_msda_fused_backward(const float *value, const int64_t *spatial_shapes, const int64_t *level_start_index, const float *oa, int ld_oa,
                           const float *ref, const float *stats, const float *fwd_output, const float *grad_output, float *grad_value,
                           float *d_oa, int ld_doa, float *d_oa_amax, float *scratch, int batch, int spatial_size, int num_heads, int channels,
                           int num_levels, int num_query, int num_point, void *stream);

/* message of the last error on this thread ("" if none) */
const char *pd_last_error(void);

/* library / ABI version, bumped when a signature changes */
int pd_abi_version(void);

/* tools / bench.py only.  For the Mask2Former geometry (3 levels, 4 points, 32 channels, num_query == spatial_size) pd_msda_backward
 * has two LDS-window kernels: windows with a 5-cell halo (all 32 channels of a head per workgroup) and with a 9-cell halo (16
 * channels per workgroup; ~1.7x the cost at small offsets, 3.5x faster at offsets of ~4 cells sigma).  Every launch counts the
 * sample points that leave the 5-cell windows and publishes {missed, looked-at} to host-mapped memory when it finishes; a launch
 * takes the 9-cell kernel when the most recent results that have arrived show more than 2 % misses (pd_debug_set("msda_gate_pct",
 * per_mille); measured break-even: 0.9 % -> 0.28 vs 0.39 ms, 3.9 % -> 0.63 vs 0.40 ms per launch at BASELINE config-2 geometry).  out3 <- {missed, looked-at} of the most recent launch whose result has arrived, and the
 * variant the most recent launch took (2 = 9-cell halo, 3 = 5-cell halo).  Reads host memory only, never synchronises. */
int pd_msda_backward_last_gate(unsigned *out3);

#ifdef __cplusplus
}
#endif
#endif /* PD_MSDA_H */
