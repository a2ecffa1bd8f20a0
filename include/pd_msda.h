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

/* replaces MSDA.ms_deform_attn_backward (vision.cpp:21, ms_deform_attn_cuda.cu:89-159) */
/* pd_msda_forward that also leaves the absolute maximum of every output row (all heads of a query) in row_amax[batch * num_query]
 * (ZERO-FILLED by the caller; atomic max) — the row-scaling input of pd_gemm_tn_f16x2 (pd_gemm.h) for the output projection that
 * reads the result.  fp32, channels = 32, 3 levels, 4 points only (the kernel of the hot path); not part of the reference's operator. */
int pd_msda_forward_amax(const void *value, const int64_t *spatial_shapes, const int64_t *level_start_index, const void *sampling_loc,
                         const void *attn_weight, void *output, float *row_amax, int batch, int spatial_size, int num_heads, int channels,
                         int num_levels, int num_query, int num_point, int im2col_step, int dtype, void *stream);

int pd_msda_backward(const void *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                     const void *sampling_loc, const void *attn_weight, const void *grad_output,
                     void *grad_value, void *grad_sampling_loc, void *grad_attn_weight,
                     int batch, int spatial_size, int num_heads, int channels, int num_levels,
                     int num_query, int num_point, int im2col_step, int dtype, void *stream);

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
