// The set criterion's arithmetic between sampled point logits and losses (C-ABI: include/pd_criterion.h; the assignment solver is lsa.hip).
//
// Everything here is small (<= 50 MB of bf16 logits at BASELINE config 2) and was, in the reference and in rounds 1-3 of this repo, a
// chain of eager launches — ~150 of the step's ~1 000 — each paying a dependent-launch boundary for a few microseconds of work, plus
// two 100 MB fp32 intermediates written only to be multiplied by an n_targets-column matrix.  Each kernel is one pass:
//   matcher_costs       a workgroup owns QR query rows of one (image, head) problem: per point it loads the target samples once, the QR
//                       logits, and accumulates softplus / sigmoid sums and the 2 n_targets dot products in registers
//                       (reference matcher.py:13-62, 108-158)
//   mask_point_losses   a workgroup per matched mask: BCE-with-logits mean and dice of its sampled points; the backward recomputes the
//                       sigmoid instead of storing it (criterion.py:25-69)
//   uncertain_points    a workgroup per matched mask: radix select (4 x 8 bits, keys in registers) of the k-th smallest |logit| among the
//                       K oversampled points, then an index-ordered compaction of the chosen coordinates (criterion.py:72-88, 181-189)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pd_common.h"
#include "pd_criterion.h"
#include "pd_msda.h"

namespace {
typedef unsigned short bf16_t;
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float softplus_t(float v, float e) { return v > 20.f ? v : fmaxf(v, 0.f) + log1pf(e); }   // e = exp(-|v|)

// ------------------------------------------------------------------------------------------------ matcher costs
constexpr int QR = 4, JT = 8;       // query rows per workgroup, target columns per pass
constexpr int NACC = QR * (2 + 2 * JT) + JT;

template <typename T>
__global__ __launch_bounds__(256) void matcher_costs(const T *__restrict__ x, const float *__restrict__ t, int64_t t_sb, int64_t t_sd,
                                                     int64_t t_sj, const float *__restrict__ prob, const int64_t *__restrict__ labels,
                                                     float *__restrict__ cost, int heads, int Q, int n, int nt, int classes, float w_mask,
                                                     float w_class, float w_dice)
{
  __shared__ float red[4][NACC];
  const int qtiles = (Q + QR - 1) / QR;
  const int p = blockIdx.x / qtiles, q0 = (blockIdx.x - p * qtiles) * QR;
  const int b = p / heads, d = p - b * heads;
  const float *tp = t + (int64_t)b * t_sb + (int64_t)d * t_sd;
  const T *xp = x + ((int64_t)p * Q + q0) * n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j0 = 0; j0 < nt; j0 += JT) {
    float sp[QR], sg[QR], xt[QR][JT], st[QR][JT], ts[JT];
#pragma unroll
    for (int r = 0; r < QR; ++r) {
      sp[r] = sg[r] = 0.f;
#pragma unroll
      for (int j = 0; j < JT; ++j) xt[r][j] = st[r][j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < JT; ++j) ts[j] = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
      float tv[JT];
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        tv[j] = j0 + j < nt ? tp[(int64_t)(j0 + j) * t_sj + i] : 0.f;
        ts[j] += tv[j];
      }
#pragma unroll
      for (int r = 0; r < QR; ++r) {
        if (q0 + r >= Q) continue;
        float v;
        if constexpr (sizeof(T) == 2) v = bf2f(xp[(int64_t)r * n + i]); else v = xp[(int64_t)r * n + i];
        const float e = expf(-fabsf(v));
        const float s = 1.f / (1.f + expf(-v));
        sp[r] += softplus_t(v, e);
        sg[r] += s;
#pragma unroll
        for (int j = 0; j < JT; ++j) { xt[r][j] = fmaf(v, tv[j], xt[r][j]); st[r][j] = fmaf(s, tv[j], st[r][j]); }
      }
    }
    // lanes -> wavefront -> workgroup
    float *mine = red[wave];
#pragma unroll
    for (int r = 0; r < QR; ++r) {
      const float a = wave_sum(sp[r]), c = wave_sum(sg[r]);
      if (lane == 0) { mine[r * (2 + 2 * JT)] = a; mine[r * (2 + 2 * JT) + 1] = c; }
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const float u = wave_sum(xt[r][j]), w = wave_sum(st[r][j]);
        if (lane == 0) { mine[r * (2 + 2 * JT) + 2 + j] = u; mine[r * (2 + 2 * JT) + 2 + JT + j] = w; }
      }
    }
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const float u = wave_sum(ts[j]);
      if (lane == 0) mine[QR * (2 + 2 * JT) + j] = u;
    }
    __syncthreads();
    if (threadIdx.x < QR * JT) {
      const int r = threadIdx.x / JT, j = threadIdx.x - r * JT;
      if (q0 + r < Q && j0 + j < nt) {
        auto tot = [&](int k) { return (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]); };
        const float spr = tot(r * (2 + 2 * JT)), sgr = tot(r * (2 + 2 * JT) + 1);
        const float xtj = tot(r * (2 + 2 * JT) + 2 + j), stj = tot(r * (2 + 2 * JT) + 2 + JT + j), tsj = tot(QR * (2 + 2 * JT) + j);
        const float cm = (spr - xtj) / (float)n;
        const float cd = 1.f - (2.f * stj + 1.f) / (sgr + tsj + 1.f);
        const int64_t lab = labels[(int64_t)b * nt + j0 + j];
        const float cc = -prob[((int64_t)p * Q + q0 + r) * classes + lab];
        cost[((int64_t)p * Q + q0 + r) * nt + j0 + j] = w_mask * cm + w_class * cc + w_dice * cd;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ mask point losses
__global__ __launch_bounds__(256) void mask_point_losses_fwd(const float *__restrict__ x, const float *__restrict__ y,
                                                             float *__restrict__ bce, float *__restrict__ dice, float *__restrict__ stats, int n)
{
  __shared__ float red[4][4];
  const float *xr = x + (int64_t)blockIdx.x * n, *yr = y + (int64_t)blockIdx.x * n;
  float a = 0.f, sy = 0.f, s1 = 0.f, y1 = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float v = xr[i], t = yr[i];
    const float e = expf(-fabsf(v));
    a += fmaxf(v, 0.f) - v * t + log1pf(e);
    const float s = 1.f / (1.f + expf(-v));
    sy = fmaf(s, t, sy); s1 += s; y1 += t;
  }
  a = wave_sum(a); sy = wave_sum(sy); s1 = wave_sum(s1); y1 = wave_sum(y1);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wave][0] = a; red[wave][1] = sy; red[wave][2] = s1; red[wave][3] = y1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    auto tot = [&](int k) { return (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]); };
    const float A = tot(0), SY = tot(1), S1 = tot(2), Y1 = tot(3);
    bce[blockIdx.x] = A / (float)n;
    dice[blockIdx.x] = 1.f - (2.f * SY + 1.f) / (S1 + Y1 + 1.f);
    stats[blockIdx.x * 3] = SY; stats[blockIdx.x * 3 + 1] = S1; stats[blockIdx.x * 3 + 2] = Y1;
  }
}

__global__ __launch_bounds__(256) void mask_point_losses_bwd(const float *__restrict__ x, const float *__restrict__ y,
                                                             const float *__restrict__ stats, const float *__restrict__ d_bce,
                                                             const float *__restrict__ d_dice, float *__restrict__ dx, int n, int chunks)
{
  const int row = blockIdx.x / chunks, c = blockIdx.x - row * chunks;
  const float num = 2.f * stats[row * 3] + 1.f, den = stats[row * 3 + 1] + stats[row * 3 + 2] + 1.f;
  const float gb = (d_bce ? d_bce[row] : 0.f) / (float)n, gd = (d_dice ? d_dice[row] : 0.f) / (den * den);
  const int per = (n + chunks - 1) / chunks, lo = c * per, hi = min(n, lo + per);
  const float *xr = x + (int64_t)row * n, *yr = y + (int64_t)row * n;
  float *o = dx + (int64_t)row * n;
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const float v = xr[i], t = yr[i];
    const float s = 1.f / (1.f + expf(-v));
    o[i] = gb * (s - t) - gd * (2.f * t * den - num) * s * (1.f - s);
  }
}

// ------------------------------------------------------------------------------------------------ uncertain points
constexpr int UT = 1024, UV = PD_UNCERTAIN_MAX_K / UT;            // threads per row, keys per thread (a contiguous run of the row)

__device__ __forceinline__ int block_excl_scan(int v, int *wsum, int &total)
{
  // inclusive scan inside the wavefront, then over the 16 wavefront totals
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(inc, o, 64);
    if (lane >= o) inc += u;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int before = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < UT / 64; ++w) {
    const int s = wsum[w];
    before += w < wave ? s : 0;
    tot += s;
  }
  __syncthreads();
  total = tot;
  return before + inc - v;
}

__global__ __launch_bounds__(UT) void uncertain_points(const float *__restrict__ logits, const float2 *__restrict__ coords,
                                                       const float2 *__restrict__ rnd, float2 *__restrict__ out, int K, int k, int n_random)
{
  __shared__ int hist[256];
  __shared__ int wsum[UT / 64];
  __shared__ unsigned s_prefix;
  __shared__ int s_remaining;
  const int row = blockIdx.x, tid = threadIdx.x;
  const float *lr = logits + (int64_t)row * K;
  const int per = (K + UT - 1) / UT;                               // <= UV
  const int lo = tid * per;
  unsigned key[UV];
#pragma unroll
  for (int m = 0; m < UV; ++m) {
    const int i = lo + m;
    key[m] = (m < per && i < K) ? (__float_as_uint(lr[i]) & 0x7fffffffu) : 0xffffffffu;     // |logit| as an ordered integer; padding sorts last
  }
  unsigned prefix = 0, mask = 0;
  int remaining = k;
#pragma unroll 1
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < UV; ++m)
      if (key[m] != 0xffffffffu && (key[m] & mask) == prefix) atomicAdd(&hist[(key[m] >> shift) & 255], 1);
    __syncthreads();
    if (tid < 64) {                                                // one wavefront: 4 bins per lane, scan, find the bin of the k-th key
      int c[4], s = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) { c[u] = hist[tid * 4 + u]; s += c[u]; }
      int inc = s;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if (tid >= o) inc += v;
      }
      int before = inc - s;
      if (before < remaining && remaining <= inc) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (before < remaining && remaining <= before + c[u]) {
            s_prefix = prefix | ((unsigned)(tid * 4 + u) << shift);
            s_remaining = remaining - before;
          }
          before += c[u];
        }
      }
    }
    __syncthreads();
    prefix = s_prefix; remaining = s_remaining;
    mask |= 255u << shift;
    __syncthreads();
  }
  // prefix = the k-th smallest key T; `remaining` of the keys equal to T are taken (lowest indices), every smaller key is
  int nless = 0, neq = 0;
#pragma unroll
  for (int m = 0; m < UV; ++m) { nless += key[m] < prefix; neq += key[m] == prefix; }
  int tot_less, tot_eq;
  const int pos_less = block_excl_scan(nless, wsum, tot_less);
  const int pos_eq = block_excl_scan(neq, wsum, tot_eq);
  float2 *orow = out + (int64_t)row * (k + n_random);
  const float2 *crow = coords + (int64_t)row * K;
  int a = pos_less, e = pos_eq;
#pragma unroll
  for (int m = 0; m < UV; ++m) {
    if (key[m] < prefix) orow[a++] = crow[lo + m];
    else if (key[m] == prefix) { if (e < remaining) orow[tot_less + e] = crow[lo + m]; ++e; }
  }
  for (int i = tid; i < n_random; i += UT) orow[k + i] = rnd[(int64_t)row * n_random + i];
}
}  // namespace

extern "C" int pd_matcher_costs(const void *x, int dtype, const float *t, int64_t t_image_stride, int64_t t_head_stride,
                                int64_t t_target_stride, const float *prob, const int64_t *labels, float *cost, int problems, int heads, int Q,
                                int n, int n_targets, int classes, float w_mask, float w_class, float w_dice, void *stream_)
{
  if (problems < 0 || heads <= 0 || Q < 0 || n <= 0 || n_targets < 0 || classes <= 0 || (dtype != PD_F32 && dtype != PD_BF16) || (problems % heads))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_matcher_costs: problems=%d heads=%d Q=%d n=%d n_targets=%d classes=%d dtype=%d", problems, heads, Q, n,
                        n_targets, classes, dtype);
  if (problems == 0 || Q == 0 || n_targets == 0) return PD_OK;
  if (!x || !t || !prob || !labels || !cost) return pd_set_error(PD_ERR_INVALID_ARG, "pd_matcher_costs: null pointer");
  const dim3 g((unsigned)(problems * ((Q + QR - 1) / QR))), b(256);
  if (dtype == PD_BF16)
    hipLaunchKernelGGL((matcher_costs<bf16_t>), g, b, 0, (hipStream_t)stream_, (const bf16_t *)x, t, t_image_stride, t_head_stride, t_target_stride,
                       prob, labels, cost, heads, Q, n, n_targets, classes, w_mask, w_class, w_dice);
  else
    hipLaunchKernelGGL((matcher_costs<float>), g, b, 0, (hipStream_t)stream_, (const float *)x, t, t_image_stride, t_head_stride, t_target_stride,
                       prob, labels, cost, heads, Q, n, n_targets, classes, w_mask, w_class, w_dice);
  return pd_check_launch("pd_matcher_costs");
}

extern "C" int pd_mask_point_losses_fwd(const float *x, const float *y, float *bce, float *dice, float *stats, int rows, int n, void *stream_)
{
  if (rows < 0 || n <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_mask_point_losses_fwd: rows=%d n=%d", rows, n);
  if (rows == 0) return PD_OK;
  if (!x || !y || !bce || !dice || !stats) return pd_set_error(PD_ERR_INVALID_ARG, "pd_mask_point_losses_fwd: null pointer");
  hipLaunchKernelGGL(mask_point_losses_fwd, dim3(rows), dim3(256), 0, (hipStream_t)stream_, x, y, bce, dice, stats, n);
  return pd_check_launch("pd_mask_point_losses_fwd");
}

extern "C" int pd_mask_point_losses_bwd(const float *x, const float *y, const float *stats, const float *d_bce, const float *d_dice, float *dx,
                                        int rows, int n, void *stream_)
{
  if (rows < 0 || n <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_mask_point_losses_bwd: rows=%d n=%d", rows, n);
  if (rows == 0) return PD_OK;
  if (!x || !y || !stats || !dx) return pd_set_error(PD_ERR_INVALID_ARG, "pd_mask_point_losses_bwd: null pointer");
  const int chunks = n >= 4096 ? 4 : 1;
  hipLaunchKernelGGL(mask_point_losses_bwd, dim3(rows * chunks), dim3(256), 0, (hipStream_t)stream_, x, y, stats, d_bce, d_dice, dx, n, chunks);
  return pd_check_launch("pd_mask_point_losses_bwd");
}

extern "C" int pd_uncertain_points(const float *logits, const float *coords, const float *random_coords, float *out, int rows, int K, int k,
                                   int n_random, void *stream_)
{
  if (rows < 0 || K <= 0 || k <= 0 || k > K || K > PD_UNCERTAIN_MAX_K || n_random < 0)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_uncertain_points: rows=%d K=%d k=%d n_random=%d (1 <= k <= K <= %d)", rows, K, k, n_random, PD_UNCERTAIN_MAX_K);
  if (rows == 0) return PD_OK;
  if (!logits || !coords || !out || (n_random && !random_coords)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_uncertain_points: null pointer");
  hipLaunchKernelGGL(uncertain_points, dim3(rows), dim3(UT), 0, (hipStream_t)stream_, logits, (const float2 *)coords, (const float2 *)random_coords,
                     (float2 *)out, K, k, n_random);
  return pd_check_launch("pd_uncertain_points");
}
