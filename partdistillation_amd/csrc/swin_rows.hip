// Token-row kernels of the fused Swin block (include/pd_swin.h): residual add (+ DropPath scale) + LayerNorm with row
// maps, forward and backward.  One wavefront per token row, lane l holds channels l, l + 64, ... (E = C / 64 registers),
// so every load / store instruction of a wave covers 256 contiguous bytes (fp32) or 128 (bf16); HBM-bound, one pass.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pd_common.h"
#include "pd_msda.h"
#include "pd_swin.h"

namespace {
typedef unsigned short bf16_t;

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ bf16_t f2bf(float f)
{
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

constexpr int WAVES = 4;   // rows per workgroup pass (forward)
// waves per workgroup in the backward (their column sums meet in LDS before the atomics): 8, or 4 where 8 copies of the
// two fp32 [C] sums would not fit the 64 KB of static LDS
template <int E> struct BwdWaves { static constexpr int value = E >= 12 ? 4 : 8; };

template <int E>
__global__ __launch_bounds__(64 * WAVES) void ln_fwd(const float *__restrict__ x, const bf16_t *__restrict__ r,
                                                     const int32_t *__restrict__ rmap, int r_rows, const float *__restrict__ rscale,
                                                     const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                     float *__restrict__ s, bf16_t *__restrict__ y, const int32_t *__restrict__ ymap,
                                                     int y_rows, const int32_t *__restrict__ zero_rows, int n_zero,
                                                     float *__restrict__ mean, float *__restrict__ rstd, int images, int L)
{
  constexpr int C = 64 * E;
  const int lane = threadIdx.x & 63;
  const int64_t R = (int64_t)images * L;
  const int64_t i = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (i >= R) {                                                         // trailing waves zero the padded rows of y
    const int64_t z = i - R;
    if (z < (int64_t)images * n_zero) {
      bf16_t *yr = y + ((z / n_zero) * y_rows + zero_rows[z % n_zero]) * C;
#pragma unroll
      for (int e = 0; e < E; ++e) yr[e * 64 + lane] = 0;
    }
    return;
  }
  const int img = (int)(i / L), t = (int)(i - (int64_t)img * L);
  float v[E];
  const float *xr = x + i * C;
#pragma unroll
  for (int e = 0; e < E; ++e) v[e] = xr[e * 64 + lane];
  if (r) {
    const bf16_t *rr = r + ((int64_t)img * r_rows + (rmap ? rmap[t] : t)) * C;
    const float sc = rscale ? rscale[img] : 1.f;
    float *sr = s + i * C;
#pragma unroll
    for (int e = 0; e < E; ++e) { v[e] = fmaf(sc, bf2f(rr[e * 64 + lane]), v[e]); sr[e * 64 + lane] = v[e]; }
  }
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) sum += v[e];
  const float mu = wave_sum(sum) * (1.f / C);
  float sq = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) { const float d = v[e] - mu; sq = fmaf(d, d, sq); }
  const float rs = rsqrtf(wave_sum(sq) * (1.f / C) + eps);
  if (lane == 0) { mean[i] = mu; rstd[i] = rs; }
  bf16_t *yr = y + ((int64_t)img * y_rows + (ymap ? ymap[t] : t)) * C;
#pragma unroll
  for (int e = 0; e < E; ++e) yr[e * 64 + lane] = f2bf(fmaf((v[e] - mu) * rs, gamma[e * 64 + lane], beta[e * 64 + lane]));
}

template <int E>
__global__ __launch_bounds__(64 * BwdWaves<E>::value) void ln_bwd(const bf16_t *__restrict__ dy, const int32_t *__restrict__ ymap, int y_rows,
                                                     const float *__restrict__ dsup, const float *__restrict__ s,
                                                     const float *__restrict__ mean, const float *__restrict__ rstd,
                                                     const float *__restrict__ gamma, float *__restrict__ ds, bf16_t *__restrict__ dr,
                                                     const int32_t *__restrict__ rmap, int r_rows, const float *__restrict__ rscale,
                                                     const int32_t *__restrict__ zero_rows, int n_zero, float *__restrict__ dgamma,
                                                     float *__restrict__ dbeta, int images, int L, int rows_per_wave)
{
  constexpr int C = 64 * E, BWAVES = BwdWaves<E>::value;
  __shared__ float red[2][BWAVES][C];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t R = (int64_t)images * L;
  const int64_t first = ((int64_t)blockIdx.x * BWAVES + wv) * rows_per_wave;
  float gm[E], ag[E], ab[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { gm[e] = gamma[e * 64 + lane]; ag[e] = 0.f; ab[e] = 0.f; }
  for (int64_t i = first; i < first + rows_per_wave; ++i) {
    if (i >= R) {
      const int64_t z = i - R;
      if (dr && z < (int64_t)images * n_zero) {
        bf16_t *rr = dr + ((z / n_zero) * r_rows + zero_rows[z % n_zero]) * C;
#pragma unroll
        for (int e = 0; e < E; ++e) rr[e * 64 + lane] = 0;
      }
      continue;
    }
    const int img = (int)(i / L), t = (int)(i - (int64_t)img * L);
    const bf16_t *dyr = dy + ((int64_t)img * y_rows + (ymap ? ymap[t] : t)) * C;
    const float *sr = s + i * C;
    const float mu = mean[i], rs = rstd[i];
    float g[E], xh[E];
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float d = bf2f(dyr[e * 64 + lane]);
      xh[e] = (sr[e * 64 + lane] - mu) * rs;
      ag[e] = fmaf(d, xh[e], ag[e]);
      ab[e] += d;
      g[e] = d * gm[e];
      a += g[e];
      b = fmaf(g[e], xh[e], b);
    }
    a = wave_sum(a) * (1.f / C);
    b = wave_sum(b) * (1.f / C);
    float *dsr = ds + i * C;
    const float sc = rscale ? rscale[img] : 1.f;
    bf16_t *rr = dr ? dr + ((int64_t)img * r_rows + (rmap ? rmap[t] : t)) * C : nullptr;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float d = rs * (g[e] - a - xh[e] * b);
      if (dsup) d += dsup[i * C + e * 64 + lane];
      dsr[e * 64 + lane] = d;
      if (rr) rr[e * 64 + lane] = f2bf(sc * d);
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) { red[0][wv][e * 64 + lane] = ag[e]; red[1][wv][e * 64 + lane] = ab[e]; }
  __syncthreads();
  for (int cidx = threadIdx.x; cidx < C; cidx += 64 * BWAVES) {
    float ga = 0.f, ba = 0.f;
#pragma unroll
    for (int w = 0; w < BWAVES; ++w) { ga += red[0][w][cidx]; ba += red[1][w][cidx]; }
    atomicAdd(dgamma + cidx, ga);
    atomicAdd(dbeta + cidx, ba);
  }
}

template <int E, typename... Args>
void launch_bwd(int64_t rows, hipStream_t st, Args... args)
{
  // rows per wave: ~512 workgroups; every workgroup ends with 2*C atomics, so few, fat workgroups
  constexpr int BW = BwdWaves<E>::value;
  int rpw = (int)((rows + 512 * BW - 1) / (512 * BW));
  rpw = rpw < 2 ? 2 : (rpw > 32 ? 32 : rpw);
  const dim3 g((unsigned)((rows + (int64_t)BW * rpw - 1) / ((int64_t)BW * rpw))), b(64 * BW);
  hipLaunchKernelGGL((ln_bwd<E>), g, b, 0, st, args..., rpw);
}

int check(int images, int L, int C, const char *who)
{
  if (images < 0 || L < 0) return pd_set_error(PD_ERR_INVALID_ARG, "%s: negative sizes", who);
  if (C <= 0 || C % 64 != 0 || C > 1536) return pd_set_error(PD_ERR_INVALID_ARG, "%s: C = %d must be a multiple of 64 up to 1536", who, C);
  return PD_OK;
}

#define E_SWITCH(C, BODY)                                             \
  switch ((C) / 64) {                                                 \
    case 1: { constexpr int E = 1; BODY; } break;                     \
    case 2: { constexpr int E = 2; BODY; } break;                     \
    case 3: { constexpr int E = 3; BODY; } break;                     \
    case 4: { constexpr int E = 4; BODY; } break;                     \
    case 6: { constexpr int E = 6; BODY; } break;                     \
    case 8: { constexpr int E = 8; BODY; } break;                     \
    case 12: { constexpr int E = 12; BODY; } break;                   \
    case 16: { constexpr int E = 16; BODY; } break;                   \
    case 24: { constexpr int E = 24; BODY; } break;                   \
    default: return pd_set_error(PD_ERR_INVALID_ARG, "pd_swin_ln: width %d has no instantiation (64 x {1,2,3,4,6,8,12,16,24})", (C)); \
  }
}  // namespace

extern "C" int pd_swin_ln_fwd(const float *x, const void *r, const int32_t *rmap, int r_rows, const float *rscale,
                              const float *gamma, const float *beta, float eps, float *s, void *y, const int32_t *ymap,
                              int y_rows, const int32_t *zero_rows, int n_zero, float *mean, float *rstd, int images, int L, int C,
                              void *stream_)
{
  int rc = check(images, L, C, "pd_swin_ln_fwd");
  if (rc) return rc;
  const int64_t R = (int64_t)images * L;
  if (R == 0) return PD_OK;
  if (!x || !gamma || !beta || !y || !mean || !rstd || (r && !s) || (n_zero > 0 && !zero_rows) || n_zero < 0)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_swin_ln_fwd: null pointer");
  const int64_t rows = R + (int64_t)images * n_zero;
  const dim3 g((unsigned)((rows + WAVES - 1) / WAVES)), b(64 * WAVES);
  hipStream_t st = (hipStream_t)stream_;
  E_SWITCH(C, hipLaunchKernelGGL((ln_fwd<E>), g, b, 0, st, x, (const bf16_t *)r, rmap, r_rows, rscale, gamma, beta, eps, s,
                                 (bf16_t *)y, ymap, y_rows, zero_rows, n_zero, mean, rstd, images, L));
  return pd_check_launch("pd_swin_ln_fwd");
}

extern "C" int pd_swin_ln_bwd(const void *dy, const int32_t *ymap, int y_rows, const float *dsup, const float *s, const float *mean,
                              const float *rstd, const float *gamma, float *ds, void *dr, const int32_t *rmap, int r_rows,
                              const float *rscale, const int32_t *zero_rows, int n_zero, float *dgamma, float *dbeta, int images,
                              int L, int C, void *stream_)
{
  int rc = check(images, L, C, "pd_swin_ln_bwd");
  if (rc) return rc;
  const int64_t R = (int64_t)images * L;
  if (R == 0) return PD_OK;
  if (!dy || !s || !mean || !rstd || !gamma || !ds || !dgamma || !dbeta || (n_zero > 0 && !zero_rows) || n_zero < 0)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_swin_ln_bwd: null pointer");
  const int64_t rows = R + (dr ? (int64_t)images * n_zero : 0);
  E_SWITCH(C, (launch_bwd<E>(rows, (hipStream_t)stream_, (const bf16_t *)dy, ymap, y_rows, dsup, s, mean, rstd, gamma, ds,
                             (bf16_t *)dr, rmap, r_rows, rscale, zero_rows, n_zero, dgamma, dbeta, images, L)));
  return pd_check_launch("pd_swin_ln_bwd");
}
