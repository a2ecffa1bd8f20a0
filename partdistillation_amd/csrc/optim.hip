// Fused clipped AdamW over flat buffers (gfx950).  HBM-bound: 28 B / fp32
// parameter (read p,g,m,v; write p,m,v) + 4 B / parameter for the norm pass.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pd_common.h"
#include "pd_msda.h"
#include "pd_optim.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void sumsq_kernel(const T *__restrict__ x, int64_t n, double *__restrict__ accum)
{
  double acc = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (sizeof(T) == 4) {
    // 16-byte lanes over the aligned body
    const int64_t n4 = n >> 2;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    float a = 0.f;
    for (int64_t k = i; k < n4; k += stride) {
      const float4 v = x4[k];
      a += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      if ((k / stride & 63) == 63) { acc += a; a = 0.f; }   // fold the fp32 partial into fp64 now and then
    }
    acc += a;
    for (int64_t k = (n4 << 2) + i; k < n; k += stride) { const double v = (double)x[k]; acc += v * v; }
  } else {
    for (int64_t k = i; k < n; k += stride) { const double v = (double)x[k]; acc += v * v; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(accum, part[0] + part[1] + part[2] + part[3]);
}

template <typename T>
__global__ __launch_bounds__(256) void adamw_kernel(T *__restrict__ p, const T *__restrict__ g, T *__restrict__ m,
                                                     T *__restrict__ v, int64_t n, T lr, T b1, T b2, T eps, T wd,
                                                     T bc1, T sqrt_bc2, const double *__restrict__ sumsq, T max_norm,
                                                     const float *__restrict__ dyn)
{
  if (dyn) { lr = (T)dyn[0]; bc1 = (T)dyn[1]; sqrt_bc2 = (T)dyn[2]; }
  T coef = 1;
  if (max_norm > 0) {
    const T total = (T)sqrt(*sumsq);
    coef = max_norm / (total + (T)1e-6);
    coef = coef < (T)1 ? coef : (T)1;
  }
  const T step_size = lr / bc1, decay = (T)1 - lr * wd;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const T gi = g[i] * coef;
    T pi = p[i] * decay;
    const T mi = b1 * m[i] + ((T)1 - b1) * gi;
    const T vi = b2 * v[i] + ((T)1 - b2) * gi * gi;
    const T denom = (T)sqrt(vi) / sqrt_bc2 + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// fp32 body vectorised 4-wide
__global__ __launch_bounds__(256) void adamw_kernel_f32x4(float4 *__restrict__ p, const float4 *__restrict__ g,
                                                           float4 *__restrict__ m, float4 *__restrict__ v, int64_t n4,
                                                           float lr, float b1, float b2, float eps, float wd, float bc1,
                                                           float sqrt_bc2, const double *__restrict__ sumsq, float max_norm,
                                                           const float *__restrict__ dyn)
{
  if (dyn) { lr = dyn[0]; bc1 = dyn[1]; sqrt_bc2 = dyn[2]; }
  float coef = 1.f;
  if (max_norm > 0.f) {
    const float total = (float)sqrt(*sumsq);
    coef = fminf(max_norm / (total + 1e-6f), 1.f);
  }
  const float step_size = lr / bc1, decay = 1.f - lr * wd;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 P = p[i], G = g[i], M = m[i], V = v[i];
    float *pp = &P.x, *gg = &G.x, *mm = &M.x, *vv = &V.x;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float gi = gg[c] * coef;
      const float mi = b1 * mm[c] + (1.f - b1) * gi;
      const float vi = b2 * vv[c] + (1.f - b2) * gi * gi;
      pp[c] = pp[c] * decay - step_size * (mi / (sqrtf(vi) / sqrt_bc2 + eps));
      mm[c] = mi; vv[c] = vi;
    }
    p[i] = P; m[i] = M; v[i] = V;
  }
}

// as adamw_kernel_f32x4, and additionally refreshes the bf16 copy of the parameters that the autocast modules read
__global__ __launch_bounds__(256) void adamw_kernel_f32x4_shadow(float4 *__restrict__ p, const float4 *__restrict__ g,
                                                                  float4 *__restrict__ m, float4 *__restrict__ v,
                                                                  ushort4 *__restrict__ shadow, int64_t n4, float lr, float b1,
                                                                  float b2, float eps, float wd, float bc1, float sqrt_bc2,
                                                                  const double *__restrict__ sumsq, float max_norm,
                                                                  const float *__restrict__ dyn)
{
  if (dyn) { lr = dyn[0]; bc1 = dyn[1]; sqrt_bc2 = dyn[2]; }
  float coef = 1.f;
  if (max_norm > 0.f) {
    const float total = (float)sqrt(*sumsq);
    coef = fminf(max_norm / (total + 1e-6f), 1.f);
  }
  const float step_size = lr / bc1, decay = 1.f - lr * wd;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 P = p[i], G = g[i], M = m[i], V = v[i];
    float *pp = &P.x, *gg = &G.x, *mm = &M.x, *vv = &V.x;
    unsigned short sh[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float gi = gg[c] * coef;
      const float mi = b1 * mm[c] + (1.f - b1) * gi;
      const float vi = b2 * vv[c] + (1.f - b2) * gi * gi;
      pp[c] = pp[c] * decay - step_size * (mi / (sqrtf(vi) / sqrt_bc2 + eps));
      mm[c] = mi; vv[c] = vi;
      unsigned u = __float_as_uint(pp[c]);
      u += 0x7fffu + ((u >> 16) & 1u);
      sh[c] = (unsigned short)(u >> 16);
    }
    p[i] = P; m[i] = M; v[i] = V;
    shadow[i] = make_ushort4(sh[0], sh[1], sh[2], sh[3]);
  }
}

inline int grid_for(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }

}  // namespace

extern "C" int pd_sumsq_accumulate(const void *x, int64_t n, int dtype, double *accum, void *stream_)
{
  if (n < 0 || !accum || (n > 0 && !x)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_sumsq_accumulate: bad argument");
  if (dtype != PD_F32 && dtype != PD_F64) return pd_set_error(PD_ERR_INVALID_ARG, "pd_sumsq_accumulate: dtype %d", dtype);
  if (n == 0) return PD_OK;
  hipStream_t s = (hipStream_t)stream_;
  if (dtype == PD_F32) {
    if (((uintptr_t)x & 15) != 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_sumsq_accumulate: buffer must be 16-byte aligned");
    // <= 1 024 workgroups: each ends in ONE fp64 atomic on the same address, and 3 000 of those took longer than the 44 MB they summed
    const int g4 = grid_for(n / 4 + 1);
    hipLaunchKernelGGL(sumsq_kernel<float>, dim3(g4 > 1024 ? 1024 : g4), dim3(256), 0, s, (const float *)x, n, accum);
  } else {
    const int g8 = grid_for(n);
    hipLaunchKernelGGL(sumsq_kernel<double>, dim3(g8 > 1024 ? 1024 : g8), dim3(256), 0, s, (const double *)x, n, accum);
  }
  return pd_check_launch("pd_sumsq_accumulate");
}

extern "C" int pd_adamw_clipped_shadow(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, void *shadow_bf16,
                                       int64_t n, double lr, double beta1, double beta2, double eps, double weight_decay,
                                       int step, const double *grad_sumsq, double max_norm, const float *dyn, void *stream_)
{
  if (n < 0 || step < 1 || (n & 3)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_adamw_clipped_shadow: n=%lld (must be a multiple of 4) step=%d", (long long)n, step);
  if (n == 0) return PD_OK;
  if (!param || !grad || !exp_avg || !exp_avg_sq || !shadow_bf16 || (max_norm > 0 && !grad_sumsq))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_adamw_clipped_shadow: null pointer");
  const double bc1 = 1.0 - pow(beta1, step), sqrt_bc2 = sqrt(1.0 - pow(beta2, step));
  hipLaunchKernelGGL(adamw_kernel_f32x4_shadow, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream_, (float4 *)param,
                     (const float4 *)grad, (float4 *)exp_avg, (float4 *)exp_avg_sq, (ushort4 *)shadow_bf16, n / 4, (float)lr,
                     (float)beta1, (float)beta2, (float)eps, (float)weight_decay, (float)bc1, (float)sqrt_bc2, grad_sumsq,
                     (float)max_norm, dyn);
  return pd_check_launch("pd_adamw_clipped_shadow");
}

extern "C" int pd_adamw_clipped(void *param, const void *grad, void *exp_avg, void *exp_avg_sq, int64_t n, int dtype,
                                double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                                const double *grad_sumsq, double max_norm, const float *dyn, void *stream_)
{
  if (n < 0 || step < 1) return pd_set_error(PD_ERR_INVALID_ARG, "pd_adamw_clipped: n=%lld step=%d", (long long)n, step);
  if (dtype != PD_F32 && dtype != PD_F64) return pd_set_error(PD_ERR_INVALID_ARG, "pd_adamw_clipped: dtype %d", dtype);
  if (n == 0) return PD_OK;
  if (!param || !grad || !exp_avg || !exp_avg_sq || (max_norm > 0 && !grad_sumsq))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_adamw_clipped: null pointer");
  hipStream_t s = (hipStream_t)stream_;
  const double bc1 = 1.0 - pow(beta1, step), sqrt_bc2 = sqrt(1.0 - pow(beta2, step));
  if (dtype == PD_F32) {
    const bool aligned = (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0;
    const int64_t n4 = aligned ? n / 4 : 0;
    if (n4 > 0)
      hipLaunchKernelGGL(adamw_kernel_f32x4, dim3(grid_for(n4)), dim3(256), 0, s, (float4 *)param, (const float4 *)grad,
                         (float4 *)exp_avg, (float4 *)exp_avg_sq, n4, (float)lr, (float)beta1, (float)beta2, (float)eps,
                         (float)weight_decay, (float)bc1, (float)sqrt_bc2, grad_sumsq, (float)max_norm, dyn);
    const int64_t done = n4 * 4;
    if (done < n)
      hipLaunchKernelGGL(adamw_kernel<float>, dim3(grid_for(n - done)), dim3(256), 0, s, (float *)param + done,
                         (const float *)grad + done, (float *)exp_avg + done, (float *)exp_avg_sq + done, n - done,
                         (float)lr, (float)beta1, (float)beta2, (float)eps, (float)weight_decay, (float)bc1,
                         (float)sqrt_bc2, grad_sumsq, (float)max_norm, dyn);
  } else {
    hipLaunchKernelGGL(adamw_kernel<double>, dim3(grid_for(n)), dim3(256), 0, s, (double *)param, (const double *)grad,
                       (double *)exp_avg, (double *)exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, sqrt_bc2,
                       grad_sumsq, max_norm, dyn);
  }
  return pd_check_launch("pd_adamw_clipped");
}
