// fp8 operand preparation for the library fp8 GEMMs of BASELINE config 5 (include/pd_fp8.h): amax + saturating
// quantisation with the gfx950 hardware converters.  Both kernels are one streaming pass (HBM-bound): 8 elements per
// lane per iteration, 16-byte loads, 8-byte stores.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pd_common.h"
#include "pd_fp8.h"
#include "pd_msda.h"

namespace {
constexpr float E4M3_MAX = 448.f, E5M2_MAX = 57344.f;

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

template <typename T> __device__ __forceinline__ void load8(const T *p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float *p, float (&v)[8])
{
  const float4 a = reinterpret_cast<const float4 *>(p)[0], b = reinterpret_cast<const float4 *>(p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<unsigned short>(const unsigned short *p, float (&v)[8])
{
  const uint4 a = *reinterpret_cast<const uint4 *>(p);
  v[0] = bf_lo(a.x); v[1] = bf_hi(a.x); v[2] = bf_lo(a.y); v[3] = bf_hi(a.y);
  v[4] = bf_lo(a.z); v[5] = bf_hi(a.z); v[6] = bf_lo(a.w); v[7] = bf_hi(a.w);
}

template <typename T>
__global__ __launch_bounds__(256) void amax_kernel(const T *__restrict__ x, int64_t n8, float *amax)
{
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float v[8];
    load8<T>(x + i * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[j]));               // fmaxf drops NaN: a NaN input does not poison the scale
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned *>(amax), __float_as_uint(m));   // m >= 0: uint order
}

template <typename T, int FMT>
__global__ __launch_bounds__(256) void quantize_kernel(const T *__restrict__ x, int64_t n8, const float *__restrict__ amax,
                                                       uint2 *__restrict__ out, float *scale_inv)
{
  constexpr float FMAX = FMT == PD_FP8_E4M3 ? E4M3_MAX : E5M2_MAX;
  const float a = amax[0];
  const float scale = a > 0.f ? __fdiv_rn(FMAX, a) : 1.f;                  // correctly rounded: the tests re-derive it
  if (blockIdx.x == 0 && threadIdx.x == 0) scale_inv[0] = a > 0.f ? __fdiv_rn(a, FMAX) : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float v[8];
    load8<T>(x + i * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fminf(fmaxf(v[j] * scale, -FMAX), FMAX);
    uint2 w = {0u, 0u};
    if (FMT == PD_FP8_E4M3) {
      w.x = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w.x, false); w.x = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w.x, true);
      w.y = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], w.y, false); w.y = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], w.y, true);
    } else {
      w.x = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], w.x, false); w.x = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], w.x, true);
      w.y = __builtin_amdgcn_cvt_pk_bf8_f32(v[4], v[5], w.y, false); w.y = __builtin_amdgcn_cvt_pk_bf8_f32(v[6], v[7], w.y, true);
    }
    out[i] = w;
  }
}

int check(const void *x, int64_t n, int dtype, const char *who)
{
  if (n < 0 || (n & 7) != 0) return pd_set_error(PD_ERR_INVALID_ARG, "%s: n = %lld must be a non-negative multiple of 8", who, (long long)n);
  if (dtype != PD_F32 && dtype != PD_BF16) return pd_set_error(PD_ERR_INVALID_ARG, "%s: dtype must be PD_F32 or PD_BF16", who);
  if (n && (!x || ((uintptr_t)x & 15) != 0)) return pd_set_error(PD_ERR_INVALID_ARG, "%s: x must be non-null and 16-byte aligned", who);
  return PD_OK;
}
int blocks_for(int64_t n8) { const int64_t b = (n8 + 255) / 256; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }
}  // namespace

extern "C" int pd_fp8_amax(const void *x, int64_t n, int dtype, float *amax, void *stream_)
{
  int rc = check(x, n, dtype, "pd_fp8_amax");
  if (rc) return rc;
  if (!amax) return pd_set_error(PD_ERR_INVALID_ARG, "pd_fp8_amax: null amax");
  if (n == 0) return PD_OK;
  hipStream_t s = (hipStream_t)stream_;
  if (dtype == PD_F32) hipLaunchKernelGGL(amax_kernel<float>, dim3(blocks_for(n / 8)), dim3(256), 0, s, (const float *)x, n / 8, amax);
  else hipLaunchKernelGGL(amax_kernel<unsigned short>, dim3(blocks_for(n / 8)), dim3(256), 0, s, (const unsigned short *)x, n / 8, amax);
  return pd_check_launch("pd_fp8_amax");
}

extern "C" int pd_fp8_quantize(const void *x, int64_t n, int dtype, const float *amax, int format, uint8_t *out, float *scale_inv,
                               void *stream_)
{
  int rc = check(x, n, dtype, "pd_fp8_quantize");
  if (rc) return rc;
  if (format != PD_FP8_E4M3 && format != PD_FP8_E5M2) return pd_set_error(PD_ERR_INVALID_ARG, "pd_fp8_quantize: unknown format %d", format);
  if (!amax || !scale_inv || (n && (!out || ((uintptr_t)out & 7) != 0)))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_fp8_quantize: null / misaligned pointer");
  hipStream_t s = (hipStream_t)stream_;
  const int64_t n8 = n / 8;
  const dim3 g(blocks_for(n8)), b(256);
#define LAUNCH(T, F) hipLaunchKernelGGL((quantize_kernel<T, F>), g, b, 0, s, (const T *)x, n8, amax, (uint2 *)out, scale_inv)
  if (dtype == PD_F32) { if (format == PD_FP8_E4M3) LAUNCH(float, PD_FP8_E4M3); else LAUNCH(float, PD_FP8_E5M2); }
  else { if (format == PD_FP8_E4M3) LAUNCH(unsigned short, PD_FP8_E4M3); else LAUNCH(unsigned short, PD_FP8_E5M2); }
#undef LAUNCH
  return pd_check_launch("pd_fp8_quantize");
}
