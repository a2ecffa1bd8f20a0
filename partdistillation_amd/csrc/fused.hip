// Fused bandwidth-bound kernels (gfx950): affine+residual+ReLU epilogue of the frozen-BN backbone convolutions
// (bf16 NHWC, 16-byte lanes) and the multi-tensor gradient gather with fused sum of squares.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "pd_common.h"
#include "pd_fused.h"
#include "pd_msda.h"

namespace {

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f2bf(float f)
{
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                                  // round to nearest even
  return (unsigned short)(u >> 16);
}

template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void affine_act_fwd(const u16x8 *__restrict__ x, const u16x8 *__restrict__ res,
                                                       const float *__restrict__ scale, const float *__restrict__ bias,
                                                       u16x8 *__restrict__ y, int64_t n8, int c8)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const int c = (int)(i % c8) * 8;
    const u16x8 v = x[i];
    u16x8 r;
    if (RES) r = res[i];
    const float4 s0 = *reinterpret_cast<const float4 *>(scale + c), s1 = *reinterpret_cast<const float4 *>(scale + c + 4);
    const float4 b0 = *reinterpret_cast<const float4 *>(bias + c), b1 = *reinterpret_cast<const float4 *>(bias + c + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float bi[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    u16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      // fp32 inside, one rounding at the end (the unfused half-precision chain rounds after every op)
      float f = bf2f(v[k]) * sc[k] + bi[k];
      if (RES) f += bf2f(r[k]);
      if (RELU) f = fmaxf(f, 0.f);
      o[k] = f2bf(f);
    }
    y[i] = o;
  }
}

template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void affine_act_bwd(const u16x8 *__restrict__ gy, const u16x8 *__restrict__ y,
                                                       const float *__restrict__ scale, u16x8 *__restrict__ gx,
                                                       u16x8 *__restrict__ gres, int64_t n8, int c8)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const int c = (int)(i % c8) * 8;
    const u16x8 g = gy[i];
    u16x8 yy;
    if (RELU) yy = y[i];
    const float4 s0 = *reinterpret_cast<const float4 *>(scale + c), s1 = *reinterpret_cast<const float4 *>(scale + c + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    u16x8 o, orr;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool on = !RELU || ((yy[k] & 0x7fff) != 0 && !(yy[k] & 0x8000));   // y > 0
      const unsigned short gk = on ? g[k] : (unsigned short)0;
      if (RES) orr[k] = gk;
      o[k] = f2bf(bf2f(gk) * sc[k]);
    }
    gx[i] = o;
    if (RES) gres[i] = orr;
  }
}

__global__ __launch_bounds__(256) void multi_gather_sumsq(const int64_t *__restrict__ src_ptrs, const int32_t *__restrict__ is_bf16,
                                                           const int32_t *__restrict__ blk_tensor, const int64_t *__restrict__ blk_start,
                                                           const int64_t *__restrict__ blk_dst, const int32_t *__restrict__ blk_len,
                                                           float *__restrict__ dst, double *__restrict__ sumsq, int block_begin)
{
  const int b = block_begin + blockIdx.x;
  const int t = blk_tensor[b];
  const int64_t start = blk_start[b];
  const int len = blk_len[b];
  float *d = dst + blk_dst[b];
  const int64_t base = src_ptrs[t];
  float acc = 0.f;
  if (base == 0) {
    for (int i = threadIdx.x; i < len; i += 256) d[i] = 0.f;
  } else if (is_bf16[t]) {
    const unsigned short *s = reinterpret_cast<const unsigned short *>(base) + start;
    for (int i = threadIdx.x; i < len; i += 256) { const float v = bf2f(s[i]); d[i] = v; acc += v * v; }
  } else {
    const float *s = reinterpret_cast<const float *>(base) + start;
    if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
      const int l4 = len >> 2;
      for (int i = threadIdx.x; i < l4; i += 256) {
        const float4 v = reinterpret_cast<const float4 *>(s)[i];
        reinterpret_cast<float4 *>(d)[i] = v;
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      }
      for (int i = (l4 << 2) + threadIdx.x; i < len; i += 256) { const float v = s[i]; d[i] = v; acc += v * v; }
    } else {
      for (int i = threadIdx.x; i < len; i += 256) { const float v = s[i]; d[i] = v; acc += v * v; }
    }
  }
  if (sumsq) {
    double a = (double)acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(sumsq, part[0] + part[1] + part[2] + part[3]);
  }
}

inline int grid_for(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }

}  // namespace

extern "C" int pd_affine_act_fwd_bf16(const void *x, const void *residual, const float *scale, const float *bias, void *y,
                                      int64_t n, int channels, int relu, void *stream_)
{
  if (n < 0 || channels <= 0 || (channels & 7) || (n % channels)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_affine_act_fwd_bf16: n=%lld channels=%d", (long long)n, channels);
  if (n == 0) return PD_OK;
  if (!x || !scale || !bias || !y) return pd_set_error(PD_ERR_INVALID_ARG, "pd_affine_act_fwd_bf16: null pointer");
  const int64_t n8 = n / 8;
  const int c8 = channels / 8;
  hipStream_t s = (hipStream_t)stream_;
  dim3 g(grid_for(n8)), b(256);
#define LAUNCH(R, A) hipLaunchKernelGGL((affine_act_fwd<R, A>), g, b, 0, s, (const u16x8 *)x, (const u16x8 *)residual, scale, bias, (u16x8 *)y, n8, c8)
  if (residual) { if (relu) LAUNCH(true, true); else LAUNCH(true, false); }
  else { if (relu) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
  return pd_check_launch("pd_affine_act_fwd_bf16");
}

extern "C" int pd_affine_act_bwd_bf16(const void *gy, const void *y, const float *scale, void *gx, void *gres, int64_t n,
                                      int channels, int relu, void *stream_)
{
  if (n < 0 || channels <= 0 || (channels & 7) || (n % channels)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_affine_act_bwd_bf16: n=%lld channels=%d", (long long)n, channels);
  if (n == 0) return PD_OK;
  if (!gy || !scale || !gx || (relu && !y)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_affine_act_bwd_bf16: null pointer");
  const int64_t n8 = n / 8;
  const int c8 = channels / 8;
  hipStream_t s = (hipStream_t)stream_;
  dim3 g(grid_for(n8)), b(256);
#define LAUNCH(R, A) hipLaunchKernelGGL((affine_act_bwd<R, A>), g, b, 0, s, (const u16x8 *)gy, (const u16x8 *)y, scale, (u16x8 *)gx, (u16x8 *)gres, n8, c8)
  if (gres) { if (relu) LAUNCH(true, true); else LAUNCH(true, false); }
  else { if (relu) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
  return pd_check_launch("pd_affine_act_bwd_bf16");
}

extern "C" int pd_multi_gather_sumsq(const int64_t *src_ptrs, const int32_t *src_is_bf16, const int32_t *blk_tensor,
                                     const int64_t *blk_start, const int64_t *blk_dst, const int32_t *blk_len, float *dst,
                                     double *sumsq, int block_begin, int block_end, void *stream_)
{
  if (block_end < block_begin) return pd_set_error(PD_ERR_INVALID_ARG, "pd_multi_gather_sumsq: bad block range");
  if (block_end == block_begin) return PD_OK;
  if (!src_ptrs || !src_is_bf16 || !blk_tensor || !blk_start || !blk_dst || !blk_len || !dst)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_multi_gather_sumsq: null pointer");
  hipLaunchKernelGGL(multi_gather_sumsq, dim3(block_end - block_begin), dim3(256), 0, (hipStream_t)stream_, src_ptrs, src_is_bf16,
                     blk_tensor, blk_start, blk_dst, blk_len, dst, sumsq, block_begin);
  return pd_check_launch("pd_multi_gather_sumsq");
}
