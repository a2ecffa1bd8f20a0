// Token-row kernels of the fused Swin block (include/pd_swin.h): residual add (+ DropPath scale) + LayerNorm with row
// maps, forward and backward.  One wavefront per token row; HBM-bound, one pass.
// Lane map (round 4): lane l holds the FOUR consecutive channels 4 (l + 64 j) .. + 3 of chunk j (C / 4 float4 per row, ceil(C / 256)
// chunks; widths that do not fill the last chunk leave its upper lanes idle), so a load / store instruction moves 16 bytes per lane
// (fp32) or 8 (bf16) — a quarter of the memory instructions of the one-channel-per-lane map it replaces (lane l: channels l, l + 64, ..)
// — and a 32-channel block of the row is EIGHT ADJACENT LANES: the rows can leave as MX-fp8 operands of the next GEMM
// (include/pd_mx8.h; block maximum = three DPP steps) next to their bf16 copy, without a pass of their own.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mx8_quant.h"
#include "pd_common.h"
#include "pd_msda.h"
#include "pd_swin.h"

int g_swin_ln_abl = 0;     // pd_debug_set "swin_ln_abl" (tools/ only): 1 = skip the column-sum atomics of the backward

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

// the float lane k (uniform) holds, in a scalar register
__device__ __forceinline__ float lane_f(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }

constexpr int WAVES = 4;   // rows per workgroup pass (forward)
// waves per workgroup in the backward (their column sums meet in LDS before the atomics): 8, or 4 where 8 copies of the
// two fp32 [C] sums would not fit the 64 KB of static LDS
template <int E> struct BwdWaves { static constexpr int value = E >= 12 ? 4 : 8; };

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// maximum over the 8 lanes of an aligned group (a 32-channel block of the row), in every lane
__device__ __forceinline__ float group8_max(float v)
{
  v = fmaxf(v, dpp_f<0xB1>(v));      // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_f<0x4E>(v));      // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_f<0x141>(v));     // row_half_mirror
  return v;
}
__device__ __forceinline__ uint2 pack_bf16x4(float a, float b, float c, float d)
{
  return make_uint2((unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16), (unsigned)f2bf(c) | ((unsigned)f2bf(d) << 16));
}
__device__ __forceinline__ float4 unpack_bf16x4(uint2 u)
{
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
// the four values a lane just rounded to bf16, again as MX fp8: block = the lane's aligned group of 8
template <int FMT>
__device__ __forceinline__ void emit_mx(uint2 bf, uint8_t *qrow, uint8_t *srow, int v4)
{
  const float4 o = unpack_bf16x4(bf);                                   // quantise what the bf16 copy holds
  const float m = group8_max(fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
  float mult;
  const unsigned byte = pdmx::mx_exponent<FMT>(m, mult);
  *reinterpret_cast<unsigned *>(qrow + 4 * v4) = pdmx::mx_pack4<FMT>(o.x, o.y, o.z, o.w, mult);
  if ((v4 & 7) == 0) srow[v4 >> 3] = (uint8_t)byte;
}

// One channel per lane and register (lane l: channels l, l + 64, ..): the map of rounds 1-3, kept for the narrow stages (C <= 192: a row
// is at most 48 float4 — the four-channel map leaves a quarter to three quarters of the lanes idle and measured 15-25 % slower there)
template <int E>
__global__ __launch_bounds__(64 * WAVES) void ln_fwd_narrow(const float *__restrict__ x, const bf16_t *__restrict__ r,
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
__global__ __launch_bounds__(64 * BwdWaves<E>::value) void ln_bwd_narrow(const bf16_t *__restrict__ dy, const int32_t *__restrict__ ymap, int y_rows,
                                                     const float *__restrict__ dsup, const float *__restrict__ s,
                                                     const float *__restrict__ mean, const float *__restrict__ rstd,
                                                     const float *__restrict__ gamma, float *__restrict__ ds, bf16_t *__restrict__ dr,
                                                     const int32_t *__restrict__ rmap, int r_rows, const float *__restrict__ rscale,
                                                     const int32_t *__restrict__ zero_rows, int n_zero, float *__restrict__ dgamma,
                                                     float *__restrict__ dbeta, int images, int L, int rows_per_wave, int n_rep, int64_t rep_stride)
{
  constexpr int C = 64 * E, BWAVES = BwdWaves<E>::value;
  dgamma += (int64_t)(blockIdx.x % n_rep) * rep_stride; dbeta += (int64_t)(blockIdx.x % n_rep) * rep_stride;
  __shared__ float red[2][BWAVES][C];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t R = (int64_t)images * L;
  const int64_t first = ((int64_t)blockIdx.x * BWAVES + wv) * rows_per_wave;
  float gm[E], ag[E], ab[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { gm[e] = gamma[e * 64 + lane]; ag[e] = 0.f; ab[e] = 0.f; }
  // row bookkeeping of all this wave's rows in one go + the next row's inputs in flight under the current one: see ln_bwd below
  const int nrows = (int)(first >= R ? 0 : (R - first < rows_per_wave ? R - first : rows_per_wave));
  int yrow_l = 0, rrow_l = 0;
  float mu_l = 0.f, rs_l = 0.f, sc_l = 1.f;
  if (lane < nrows) {
    const int64_t il = first + lane;
    const int img = (int)(il / L), t = (int)(il - (int64_t)img * L);
    yrow_l = img * y_rows + (ymap ? ymap[t] : t);
    rrow_l = img * r_rows + (rmap ? rmap[t] : t);
    mu_l = mean[il]; rs_l = rstd[il];
    if (rscale) sc_l = rscale[img];
  }
  struct RowIn { bf16_t d[E]; float sv[E], up[E]; };
  auto load_row = [&](int k, RowIn &w) {
    const int64_t i = first + k;
    const bf16_t *dyr = dy + (int64_t)__builtin_amdgcn_readlane(yrow_l, k) * C;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      w.d[e] = dyr[e * 64 + lane];
      w.sv[e] = s[i * C + e * 64 + lane];
      w.up[e] = dsup ? dsup[i * C + e * 64 + lane] : 0.f;
    }
  };
  RowIn cur, nxt;
  if (nrows > 0) load_row(0, cur);
  for (int k = 0; k < nrows; ++k) {
    if (k + 1 < nrows) load_row(k + 1, nxt);
    const int64_t i = first + k;
    const float mu = lane_f(mu_l, k), rs = lane_f(rs_l, k), sc = lane_f(sc_l, k);
    float g[E], xh[E];
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float d = bf2f(cur.d[e]);
      xh[e] = (cur.sv[e] - mu) * rs;
      ag[e] = fmaf(d, xh[e], ag[e]);
      ab[e] += d;
      g[e] = d * gm[e];
      a += g[e];
      b = fmaf(g[e], xh[e], b);
    }
    a = wave_sum(a) * (1.f / C);
    b = wave_sum(b) * (1.f / C);
    float *dsr = ds + i * C;
    bf16_t *rr = dr ? dr + (int64_t)__builtin_amdgcn_readlane(rrow_l, k) * C : nullptr;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float d = rs * (g[e] - a - xh[e] * b) + cur.up[e];
      dsr[e * 64 + lane] = d;
      if (rr) rr[e * 64 + lane] = f2bf(sc * d);
    }
    cur = nxt;
  }
  if (dr)
    for (int64_t i = first + nrows; i < first + rows_per_wave; ++i) {     // trailing rows of the launch zero the padded rows of dr
      const int64_t z = i - R;
      if (z < 0 || z >= (int64_t)images * n_zero) continue;
      bf16_t *rr = dr + ((z / n_zero) * r_rows + zero_rows[z % n_zero]) * C;
#pragma unroll
      for (int e = 0; e < E; ++e) rr[e * 64 + lane] = 0;
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

// MXF: -1 no MX copy, else the fp8 format of y_q
template <int E, int MXF>
__global__ __launch_bounds__(64 * WAVES) void ln_fwd(const float *__restrict__ x, const bf16_t *__restrict__ r,
                                                     const int32_t *__restrict__ rmap, int r_rows, const float *__restrict__ rscale,
                                                     const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                     float *__restrict__ s, bf16_t *__restrict__ y, const int32_t *__restrict__ ymap,
                                                     int y_rows, const int32_t *__restrict__ zero_rows, int n_zero,
                                                     float *__restrict__ mean, float *__restrict__ rstd, int images, int L,
                                                     uint8_t *__restrict__ y_q, uint8_t *__restrict__ y_s)
{
  constexpr int C = 64 * E, V4 = 16 * E, NJ = (V4 + 63) / 64;
  const int lane = threadIdx.x & 63;
  const int64_t R = (int64_t)images * L;
  const int64_t i = (int64_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (i >= R) {                                                         // trailing waves zero the padded rows of y
    const int64_t z = i - R;
    if (z < (int64_t)images * n_zero) {
      const int64_t row = (z / n_zero) * y_rows + zero_rows[z % n_zero];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int v4 = lane + 64 * j;
        if (v4 < V4) {
          *reinterpret_cast<uint2 *>(y + row * C + 4 * v4) = make_uint2(0u, 0u);
          if (MXF >= 0) { *reinterpret_cast<unsigned *>(y_q + row * C + 4 * v4) = 0u; if ((v4 & 7) == 0) y_s[row * (C / 32) + (v4 >> 3)] = 0; }
        }
      }
    }
    return;
  }
  const int img = (int)(i / L), t = (int)(i - (int64_t)img * L);
  float4 v[NJ];
  const float *xr = x + i * C;
#pragma unroll
  for (int j = 0; j < NJ; ++j) v[j] = (lane + 64 * j < V4) ? *reinterpret_cast<const float4 *>(xr + 4 * (lane + 64 * j)) : make_float4(0.f, 0.f, 0.f, 0.f);
  if (r) {
    const bf16_t *rr = r + ((int64_t)img * r_rows + (rmap ? rmap[t] : t)) * C;
    const float sc = rscale ? rscale[img] : 1.f;
    float *sr = s + i * C;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (lane + 64 * j < V4) {
        const float4 q = unpack_bf16x4(*reinterpret_cast<const uint2 *>(rr + 4 * (lane + 64 * j)));
        v[j].x = fmaf(sc, q.x, v[j].x); v[j].y = fmaf(sc, q.y, v[j].y); v[j].z = fmaf(sc, q.z, v[j].z); v[j].w = fmaf(sc, q.w, v[j].w);
        *reinterpret_cast<float4 *>(sr + 4 * (lane + 64 * j)) = v[j];
      }
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  const float mu = wave_sum(sum) * (1.f / C);
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    if (lane + 64 * j < V4) {
      const float a = v[j].x - mu, b = v[j].y - mu, c = v[j].z - mu, d = v[j].w - mu;
      sq += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rs = rsqrtf(wave_sum(sq) * (1.f / C) + eps);
  if (lane == 0) { mean[i] = mu; rstd[i] = rs; }
  const int64_t yrow = (int64_t)img * y_rows + (ymap ? ymap[t] : t);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int v4 = lane + 64 * j;
    if (v4 < V4) {                                                      // (whole groups of 8 lanes: V4 is a multiple of 8)
      const float4 g = *reinterpret_cast<const float4 *>(gamma + 4 * v4), bt = *reinterpret_cast<const float4 *>(beta + 4 * v4);
      const uint2 o = pack_bf16x4(fmaf((v[j].x - mu) * rs, g.x, bt.x), fmaf((v[j].y - mu) * rs, g.y, bt.y),
                                  fmaf((v[j].z - mu) * rs, g.z, bt.z), fmaf((v[j].w - mu) * rs, g.w, bt.w));
      *reinterpret_cast<uint2 *>(y + yrow * C + 4 * v4) = o;
      if (MXF >= 0) emit_mx<(MXF >= 0 ? MXF : 0)>(o, y_q + yrow * C, y_s + yrow * (C / 32), v4);
    }
  }
}

template <int E, int MXF>
__global__ __launch_bounds__(64 * BwdWaves<E>::value) void ln_bwd(const bf16_t *__restrict__ dy, const int32_t *__restrict__ ymap, int y_rows,
                                                     const float *__restrict__ dsup, const float *__restrict__ s,
                                                     const float *__restrict__ mean, const float *__restrict__ rstd,
                                                     const float *__restrict__ gamma, float *__restrict__ ds, bf16_t *__restrict__ dr,
                                                     const int32_t *__restrict__ rmap, int r_rows, const float *__restrict__ rscale,
                                                     const int32_t *__restrict__ zero_rows, int n_zero, float *__restrict__ dgamma,
                                                     float *__restrict__ dbeta, int images, int L, int rows_per_wave,
                                                     uint8_t *__restrict__ dr_q, uint8_t *__restrict__ dr_s, int abl, int n_rep, int64_t rep_stride)
{
  constexpr int C = 64 * E, BWAVES = BwdWaves<E>::value, V4 = 16 * E, NJ = (V4 + 63) / 64;
  // column sums: every workgroup ends with 2 C atomic adds; ~500 workgroups on the same 2 C addresses serialise (19 of the 47 us of a
  // 14 112 x 768 launch) — the caller may hand n_rep zero-filled copies rep_stride floats apart and sum them once per stage
  dgamma += (int64_t)(blockIdx.x % n_rep) * rep_stride; dbeta += (int64_t)(blockIdx.x % n_rep) * rep_stride;
  __shared__ float red[2][BWAVES][C];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t R = (int64_t)images * L;
  const int64_t first = ((int64_t)blockIdx.x * BWAVES + wv) * rows_per_wave;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 gm[NJ], ag[NJ], ab[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) { gm[j] = (lane + 64 * j < V4) ? *reinterpret_cast<const float4 *>(gamma + 4 * (lane + 64 * j)) : zero4; ag[j] = zero4; ab[j] = zero4; }
  // Row bookkeeping of ALL this wave's rows at once, lane k for row first + k (rows_per_wave <= 32): the row-map entries, mean / rstd and the
  // DropPath scale are one coalesced load each instead of a dependent load in front of every row's data, and the loop below reads them
  // with v_readlane.  Round 6: a wave walked its rows one after the other — map entry, then the row, then the reductions, then the next map
  // entry — 19.7 us for 85 MB at Swin-B's third stage (three rows per wave), 106 us for 200 MB at the first (32 rows per wave).
  const int nrows = (int)(first >= R ? 0 : (R - first < rows_per_wave ? R - first : rows_per_wave));
  int yrow_l = 0, rrow_l = 0;
  float mu_l = 0.f, rs_l = 0.f, sc_l = 1.f;
  if (lane < nrows) {
    const int64_t il = first + lane;
    const int img = (int)(il / L), t = (int)(il - (int64_t)img * L);
    yrow_l = img * y_rows + (ymap ? ymap[t] : t);
    rrow_l = img * r_rows + (rmap ? rmap[t] : t);
    mu_l = mean[il]; rs_l = rstd[il];
    if (rscale) sc_l = rscale[img];
  }
  struct RowIn { uint2 d[NJ]; float4 sv[NJ], up[NJ]; };
  auto load_row = [&](int k, RowIn &w) {                                 // the three input rows of row first + k: issued one row ahead of their use
    const int64_t i = first + k;
    const bf16_t *dyr = dy + (int64_t)__builtin_amdgcn_readlane(yrow_l, k) * C;
    const float *sr = s + i * C;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int v4 = lane + 64 * j;
      if (NJ * 64 == V4 || v4 < V4) {
        w.d[j] = *reinterpret_cast<const uint2 *>(dyr + 4 * v4);
        w.sv[j] = *reinterpret_cast<const float4 *>(sr + 4 * v4);
        w.up[j] = dsup ? *reinterpret_cast<const float4 *>(dsup + i * C + 4 * v4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  RowIn cur, nxt;
  if (nrows > 0) load_row(0, cur);
  for (int k = 0; k < nrows; ++k) {
    if (k + 1 < nrows) load_row(k + 1, nxt);
    const int64_t i = first + k;
    const float mu = lane_f(mu_l, k), rs = lane_f(rs_l, k), sc = lane_f(sc_l, k);
    float4 g[NJ], xh[NJ];
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int v4 = lane + 64 * j;
      const bool act = NJ * 64 == V4 || v4 < V4;
      const float4 d = act ? unpack_bf16x4(cur.d[j]) : zero4;
      const float4 sv = act ? cur.sv[j] : make_float4(mu, mu, mu, mu);
      xh[j] = make_float4((sv.x - mu) * rs, (sv.y - mu) * rs, (sv.z - mu) * rs, (sv.w - mu) * rs);
      ag[j].x = fmaf(d.x, xh[j].x, ag[j].x); ag[j].y = fmaf(d.y, xh[j].y, ag[j].y); ag[j].z = fmaf(d.z, xh[j].z, ag[j].z); ag[j].w = fmaf(d.w, xh[j].w, ag[j].w);
      ab[j].x += d.x; ab[j].y += d.y; ab[j].z += d.z; ab[j].w += d.w;
      g[j] = make_float4(d.x * gm[j].x, d.y * gm[j].y, d.z * gm[j].z, d.w * gm[j].w);
      a += (g[j].x + g[j].y) + (g[j].z + g[j].w);
      b += (g[j].x * xh[j].x + g[j].y * xh[j].y) + (g[j].z * xh[j].z + g[j].w * xh[j].w);
    }
    a = wave_sum(a) * (1.f / C);
    b = wave_sum(b) * (1.f / C);
    float *dsr = ds + i * C;
    const int64_t rrow = __builtin_amdgcn_readlane(rrow_l, k);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int v4 = lane + 64 * j;
      if (NJ * 64 == V4 || v4 < V4) {
        const float4 up = cur.up[j];
        float4 d;
        d.x = rs * (g[j].x - a - xh[j].x * b) + up.x; d.y = rs * (g[j].y - a - xh[j].y * b) + up.y;
        d.z = rs * (g[j].z - a - xh[j].z * b) + up.z; d.w = rs * (g[j].w - a - xh[j].w * b) + up.w;
        *reinterpret_cast<float4 *>(dsr + 4 * v4) = d;
        if (dr) {
          const uint2 o = pack_bf16x4(sc * d.x, sc * d.y, sc * d.z, sc * d.w);
          *reinterpret_cast<uint2 *>(dr + rrow * C + 4 * v4) = o;
          if (MXF >= 0) emit_mx<(MXF >= 0 ? MXF : 0)>(o, dr_q + rrow * C, dr_s + rrow * (C / 32), v4);
        }
      }
    }
    cur = nxt;
  }
  if (dr)                                                                 // trailing rows of the launch zero the padded rows of dr
    for (int64_t i = first + nrows; i < first + rows_per_wave; ++i) {
      const int64_t z = i - R;
      if (z < 0 || z >= (int64_t)images * n_zero) continue;
      const int64_t row = (z / n_zero) * r_rows + zero_rows[z % n_zero];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int v4 = lane + 64 * j;
        if (v4 < V4) {
          *reinterpret_cast<uint2 *>(dr + row * C + 4 * v4) = make_uint2(0u, 0u);
          if (MXF >= 0) { *reinterpret_cast<unsigned *>(dr_q + row * C + 4 * v4) = 0u; if ((v4 & 7) == 0) dr_s[row * (C / 32) + (v4 >> 3)] = 0; }
        }
      }
    }
  if (abl & 1) return;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int v4 = lane + 64 * j;
    if (v4 < V4) { *reinterpret_cast<float4 *>(&red[0][wv][4 * v4]) = ag[j]; *reinterpret_cast<float4 *>(&red[1][wv][4 * v4]) = ab[j]; }
  }
  __syncthreads();
  for (int cidx = threadIdx.x; cidx < C; cidx += 64 * BWAVES) {
    float ga = 0.f, ba = 0.f;
#pragma unroll
    for (int w = 0; w < BWAVES; ++w) { ga += red[0][w][cidx]; ba += red[1][w][cidx]; }
    atomicAdd(dgamma + cidx, ga);
    atomicAdd(dbeta + cidx, ba);
  }
}

template <int E>
void bwd_geometry(int64_t rows, dim3 &g, dim3 &b, int &rpw)
{
  // rows per wave: ~512 workgroups; every workgroup ends with 2*C atomics, so few, fat workgroups (1024 / 2048 workgroups of 16 / 8 rows per
  // wave measured the same with the sums in 8 copies and worse without)
  constexpr int BW = BwdWaves<E>::value;
  rpw = (int)((rows + 512 * BW - 1) / (512 * BW));
  rpw = rpw < 2 ? 2 : (rpw > 32 ? 32 : rpw);
  g = dim3((unsigned)((rows + (int64_t)BW * rpw - 1) / ((int64_t)BW * rpw)));
  b = dim3(64 * BW);
}

int check(int images, int L, int C, const char *who)
{
  if (images < 0 || L < 0) return pd_set_error(PD_ERR_INVALID_ARG, "%s: negative sizes", who);
  if (C <= 0 || C % 64 != 0 || C > 1536) return pd_set_error(PD_ERR_INVALID_ARG, "%s: C = %d must be a multiple of 64 up to 1536", who, C);
  return PD_OK;
}

int check_mx(const void *q, const void *sc, int fmt, const char *who)
{
  if ((q != nullptr) != (sc != nullptr)) return pd_set_error(PD_ERR_INVALID_ARG, "%s: the MX element and scale pointers come together", who);
  if (q && fmt != PD_MX8_E4M3 && fmt != PD_MX8_E5M2) return pd_set_error(PD_ERR_INVALID_ARG, "%s: unknown fp8 format %d", who, fmt);
  if ((uintptr_t)q & 3) return pd_set_error(PD_ERR_INVALID_ARG, "%s: misaligned MX element pointer", who);
  return PD_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Plain fp32 LayerNorm over token rows, forward and backward: the per-stage OUTPUT norms of the Swin backbone (reference
// modeling/backbone/swin.py:675-680 `norm_layer = getattr(self, f"norm{i}"); x_out = norm_layer(x_out)`), which autocast keeps in fp32 — input
// the stage's fp32 residual stream, output the fp32 map the pixel decoder reads.  Rounds 1-4 left them to ATen (0.97 ms per Swin-B step,
// 1.93 ms per Swin-L step).  One wavefront per row, 16 bytes per lane and chunk, C % 4 == 0, C <= 3072 (NQ = ceil(C / 256) chunks; the widths above 1536 are PatchMerging's
// LayerNorm over 4 C channels, reference :339, and the patch embedding's norm :565 — both ATen through round 5: 0.6 ms per Swin-B step).
template <int NQ>
__global__ __launch_bounds__(256) void ln_rows_f32_fwd(const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       float eps, float *__restrict__ y, float *__restrict__ mean, float *__restrict__ rstd,
                                                       int64_t rows, int C)
{
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float *xr = x + row * C;
  float4 v[NQ];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int c = 4 * (lane + 64 * j);
    v[j] = c < C ? *reinterpret_cast<const float4 *>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mu = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    if (4 * (lane + 64 * j) < C) {
      const float a = v[j].x - mu, b = v[j].y - mu, c2 = v[j].z - mu, d = v[j].w - mu;
      sq += (a * a + b * b) + (c2 * c2 + d * d);
    }
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float rs = rsqrtf(sq / (float)C + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  float *yr = y + row * C;
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int c = 4 * (lane + 64 * j);
    if (c < C) {
      const float4 g = *reinterpret_cast<const float4 *>(gamma + c), b = *reinterpret_cast<const float4 *>(beta + c);
      *reinterpret_cast<float4 *>(yr + c) = make_float4(fmaf((v[j].x - mu) * rs, g.x, b.x), fmaf((v[j].y - mu) * rs, g.y, b.y),
                                                        fmaf((v[j].z - mu) * rs, g.z, b.z), fmaf((v[j].w - mu) * rs, g.w, b.w));
    }
  }
}

// dx = rstd (gh - mean_c(gh) - xhat mean_c(gh xhat)), gh = dy gamma;  dgamma += sum_rows dy xhat;  dbeta += sum_rows dy.  8 wavefronts per
// workgroup, each walking rows; their column sums meet in LDS and leave as 2 C atomics per workgroup (few, fat workgroups: see pd_add_layernorm_bwd)
template <int NQ> struct RowsBwdWaves { static constexpr int value = NQ >= 12 ? 2 : (NQ >= 4 ? 4 : 8); };   // (column sums of all wavefronts in <= 64 KB of LDS)
template <int NQ>
__global__ __launch_bounds__(64 * RowsBwdWaves<NQ>::value) void ln_rows_f32_bwd(const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ mean,
                                                       const float *__restrict__ rstd, const float *__restrict__ gamma, float *__restrict__ dx,
                                                       float *__restrict__ dgamma, float *__restrict__ dbeta, int64_t rows, int C)
{
  constexpr int WV = RowsBwdWaves<NQ>::value;
  __shared__ float red[WV][2][NQ * 256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4 gm[NQ], ag[NQ], ab[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int c = 4 * (lane + 64 * j);
    gm[j] = c < C ? *reinterpret_cast<const float4 *>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    ag[j] = ab[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int64_t row = (int64_t)blockIdx.x * WV + wave; row < rows; row += (int64_t)gridDim.x * WV) {
    const float mu = mean[row], rs = rstd[row];
    float4 g[NQ], xh[NQ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int c = 4 * (lane + 64 * j);
      if (c < C) {
        g[j] = *reinterpret_cast<const float4 *>(dy + row * C + c);
        const float4 xv = *reinterpret_cast<const float4 *>(x + row * C + c);
        xh[j] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
      } else {
        g[j] = xh[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      ag[j].x += g[j].x * xh[j].x; ag[j].y += g[j].y * xh[j].y; ag[j].z += g[j].z * xh[j].z; ag[j].w += g[j].w * xh[j].w;
      ab[j].x += g[j].x; ab[j].y += g[j].y; ab[j].z += g[j].z; ab[j].w += g[j].w;
      g[j].x *= gm[j].x; g[j].y *= gm[j].y; g[j].z *= gm[j].z; g[j].w *= gm[j].w;
      s1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
      s2 += (g[j].x * xh[j].x + g[j].y * xh[j].y) + (g[j].z * xh[j].z + g[j].w * xh[j].w);
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    const float m1 = s1 / (float)C, m2 = s2 / (float)C;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int c = 4 * (lane + 64 * j);
      if (c < C)
        *reinterpret_cast<float4 *>(dx + row * C + c) = make_float4(rs * (g[j].x - m1 - xh[j].x * m2), rs * (g[j].y - m1 - xh[j].y * m2),
                                                                    rs * (g[j].z - m1 - xh[j].z * m2), rs * (g[j].w - m1 - xh[j].w * m2));
    }
  }
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    *reinterpret_cast<float4 *>(&red[wave][0][4 * (lane + 64 * j)]) = ag[j];
    *reinterpret_cast<float4 *>(&red[wave][1][4 * (lane + 64 * j)]) = ab[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 64 * WV) {
    const int k = i / C, c = i - k * C;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < WV; ++w) sum += red[w][k][c];
    atomicAdd((k ? dbeta : dgamma) + c, sum);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Patch merging's gather + LayerNorm (reference modeling/backbone/swin.py:325-339): the 2 x 2 neighbours (2 i + rp, 2 j + cp) of the fp32 stage output,
// concatenated as channel block 2 cp + rp of a 4 C row, normalised in fp32 and written in bf16 — the operand of the reduction Linear.  Through round 6's
// first half this was a permuted copy (ATen), a LayerNorm over the copy (fp32 out) and a cast: 370 MB of traffic for 100 at Swin-B's first merge; the backward
// a cast, the LayerNorm' and the permuted copy back.  One wavefront per merged row, 16 bytes per lane and chunk like ln_rows_f32_*; every input pixel belongs
// to exactly one merged row, so the backward's dx is a plain store.
__device__ __forceinline__ int64_t merge_src(int64_t row, int c, int H, int W, int C)       // element offset of channel c (of 4 C) of merged row `row` in x [B, H, W, C]
{
  const int Wh = W >> 1, Hh = H >> 1;
  const int64_t b = row / ((int64_t)Hh * Wh);
  const int rem = (int)(row - b * Hh * Wh), i = rem / Wh, j = rem - i * Wh;
  const int k = c / C, cin = c - k * C;
  return ((b * H + 2 * i + (k & 1)) * W + 2 * j + (k >> 1)) * (int64_t)C + cin;
}

template <int NQ>
__global__ __launch_bounds__(256) void ln_merge_fwd(const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                    bf16_t *__restrict__ y, float *__restrict__ mean, float *__restrict__ rstd, int64_t rows, int H, int W, int C)
{
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, C4 = 4 * C;
  if (row >= rows) return;
  float4 v[NQ];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int c = 4 * (lane + 64 * j);
    v[j] = c < C4 ? *reinterpret_cast<const float4 *>(x + merge_src(row, c, H, W, C)) : make_float4(0.f, 0.f, 0.f, 0.f);
    sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mu = sum / (float)C4;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    if (4 * (lane + 64 * j) < C4) {
      const float a = v[j].x - mu, b = v[j].y - mu, c2 = v[j].z - mu, d = v[j].w - mu;
      sq += (a * a + b * b) + (c2 * c2 + d * d);
    }
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float rs = rsqrtf(sq / (float)C4 + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  bf16_t *yr = y + row * C4;
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int c = 4 * (lane + 64 * j);
    if (c < C4) {
      const float4 g = *reinterpret_cast<const float4 *>(gamma + c), b = *reinterpret_cast<const float4 *>(beta + c);
      *reinterpret_cast<uint2 *>(yr + c) = pack_bf16x4(fmaf((v[j].x - mu) * rs, g.x, b.x), fmaf((v[j].y - mu) * rs, g.y, b.y),
                                                        fmaf((v[j].z - mu) * rs, g.z, b.z), fmaf((v[j].w - mu) * rs, g.w, b.w));
    }
  }
}

template <int NQ>
__global__ __launch_bounds__(64 * RowsBwdWaves<NQ>::value) void ln_merge_bwd(const bf16_t *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ mean,
                                                       const float *__restrict__ rstd, const float *__restrict__ gamma, float *__restrict__ dx,
                                                       float *__restrict__ dgamma, float *__restrict__ dbeta, int64_t rows, int H, int W, int C)
{
  constexpr int WV = RowsBwdWaves<NQ>::value;
  __shared__ float red[WV][2][NQ * 256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, C4 = 4 * C;
  float4 gm[NQ], ag[NQ], ab[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int c = 4 * (lane + 64 * j);
    gm[j] = c < C4 ? *reinterpret_cast<const float4 *>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    ag[j] = ab[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int64_t row = (int64_t)blockIdx.x * WV + wave; row < rows; row += (int64_t)gridDim.x * WV) {
    const float mu = mean[row], rs = rstd[row];
    float4 g[NQ], xh[NQ];
    int64_t src[NQ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int c = 4 * (lane + 64 * j);
      src[j] = 0;
      if (c < C4) {
        src[j] = merge_src(row, c, H, W, C);
        g[j] = unpack_bf16x4(*reinterpret_cast<const uint2 *>(dy + row * C4 + c));
        const float4 xv = *reinterpret_cast<const float4 *>(x + src[j]);
        xh[j] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
      } else {
        g[j] = xh[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      ag[j].x += g[j].x * xh[j].x; ag[j].y += g[j].y * xh[j].y; ag[j].z += g[j].z * xh[j].z; ag[j].w += g[j].w * xh[j].w;
      ab[j].x += g[j].x; ab[j].y += g[j].y; ab[j].z += g[j].z; ab[j].w += g[j].w;
      g[j].x *= gm[j].x; g[j].y *= gm[j].y; g[j].z *= gm[j].z; g[j].w *= gm[j].w;
      s1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
      s2 += (g[j].x * xh[j].x + g[j].y * xh[j].y) + (g[j].z * xh[j].z + g[j].w * xh[j].w);
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    const float m1 = s1 / (float)C4, m2 = s2 / (float)C4;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      if (4 * (lane + 64 * j) < C4)
        *reinterpret_cast<float4 *>(dx + src[j]) = make_float4(rs * (g[j].x - m1 - xh[j].x * m2), rs * (g[j].y - m1 - xh[j].y * m2),
                                                               rs * (g[j].z - m1 - xh[j].z * m2), rs * (g[j].w - m1 - xh[j].w * m2));
    }
  }
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    *reinterpret_cast<float4 *>(&red[wave][0][4 * (lane + 64 * j)]) = ag[j];
    *reinterpret_cast<float4 *>(&red[wave][1][4 * (lane + 64 * j)]) = ab[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C4; i += 64 * WV) {
    const int k = i / C4, c = i - k * C4;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < WV; ++w) sum += red[w][k][c];
    atomicAdd((k ? dbeta : dgamma) + c, sum);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The END of a fused Swin stage: the last block's MLP output joins the residual stream (s = cur + DropPath scale * r, the stage's fp32 output, reference
// swin.py:234-236) and the stage's output norm (:675-680) reads it — one kernel instead of a cast, a product, a sum (ATen) and the row LayerNorm; the backward
// forms the stream's gradient (LayerNorm' of the norm's gradient + whatever arrives for s from the next stage) and its 16-bit DropPath-scaled copy, the two
// tensors the stage's recorded backward starts from, instead of the row LayerNorm', autograd's sum, a product and a cast.
template <int NQ>
__global__ __launch_bounds__(256) void ln_tail_fwd(const float *__restrict__ cur, const bf16_t *__restrict__ r, const float *__restrict__ rscale, int L,
                                                   const float *__restrict__ gamma, const float *__restrict__ beta, float eps, float *__restrict__ s,
                                                   float *__restrict__ y, float *__restrict__ mean, float *__restrict__ rstd, int64_t rows, int C)
{
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float sc = rscale ? rscale[row / L] : 1.f;
  float4 v[NQ];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int c = 4 * (lane + 64 * j);
    if (c < C) {
      const float4 x = *reinterpret_cast<const float4 *>(cur + row * C + c);
      const float4 rr = unpack_bf16x4(*reinterpret_cast<const uint2 *>(r + row * C + c));
      {
        // (the ATen chain rounds the product before the sum — out = r.float() * scale, then + cur —; __fmul_rn / __fadd_rn are plain operators to hipcc
        // and contract into one fma: the pragma keeps the two roundings, the stage output stays bit-identical to the unfused path)
#pragma clang fp contract(off)
        const float px = rr.x * sc, py = rr.y * sc, pz = rr.z * sc, pw = rr.w * sc;
        v[j] = make_float4(px + x.x, py + x.y, pz + x.z, pw + x.w);
      }
      *reinterpret_cast<float4 *>(s + row * C + c) = v[j];
    } else {
      v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mu = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    if (4 * (lane + 64 * j) < C) {
      const float a = v[j].x - mu, b = v[j].y - mu, c2 = v[j].z - mu, d = v[j].w - mu;
      sq += (a * a + b * b) + (c2 * c2 + d * d);
    }
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float rs = rsqrtf(sq / (float)C + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int c = 4 * (lane + 64 * j);
    if (c < C) {
      const float4 g = *reinterpret_cast<const float4 *>(gamma + c), b = *reinterpret_cast<const float4 *>(beta + c);
      *reinterpret_cast<float4 *>(y + row * C + c) = make_float4(fmaf((v[j].x - mu) * rs, g.x, b.x), fmaf((v[j].y - mu) * rs, g.y, b.y),
                                                                 fmaf((v[j].z - mu) * rs, g.z, b.z), fmaf((v[j].w - mu) * rs, g.w, b.w));
    }
  }
}

template <int NQ>
__global__ __launch_bounds__(64 * RowsBwdWaves<NQ>::value) void ln_tail_bwd(const float *__restrict__ dy, const float *__restrict__ dsum, const float *__restrict__ s,
                                                       const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                       const float *__restrict__ rscale, int L, float *__restrict__ dsup, bf16_t *__restrict__ df,
                                                       float *__restrict__ dgamma, float *__restrict__ dbeta, int64_t rows, int C)
{
  constexpr int WV = RowsBwdWaves<NQ>::value;
  __shared__ float red[WV][2][NQ * 256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4 gm[NQ], ag[NQ], ab[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int c = 4 * (lane + 64 * j);
    gm[j] = c < C ? *reinterpret_cast<const float4 *>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    ag[j] = ab[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int64_t row = (int64_t)blockIdx.x * WV + wave; row < rows; row += (int64_t)gridDim.x * WV) {
    const float mu = mean[row], rs = rstd[row], sc = rscale ? rscale[row / L] : 1.f;
    float4 g[NQ], xh[NQ], up[NQ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int c = 4 * (lane + 64 * j);
      if (c < C) {
        g[j] = *reinterpret_cast<const float4 *>(dy + row * C + c);
        const float4 xv = *reinterpret_cast<const float4 *>(s + row * C + c);
        up[j] = dsum ? *reinterpret_cast<const float4 *>(dsum + row * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        xh[j] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
      } else {
        g[j] = xh[j] = up[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      ag[j].x += g[j].x * xh[j].x; ag[j].y += g[j].y * xh[j].y; ag[j].z += g[j].z * xh[j].z; ag[j].w += g[j].w * xh[j].w;
      ab[j].x += g[j].x; ab[j].y += g[j].y; ab[j].z += g[j].z; ab[j].w += g[j].w;
      g[j].x *= gm[j].x; g[j].y *= gm[j].y; g[j].z *= gm[j].z; g[j].w *= gm[j].w;
      s1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
      s2 += (g[j].x * xh[j].x + g[j].y * xh[j].y) + (g[j].z * xh[j].z + g[j].w * xh[j].w);
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    const float m1 = s1 / (float)C, m2 = s2 / (float)C;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int c = 4 * (lane + 64 * j);
      if (c < C) {
        const float4 d = make_float4(up[j].x + rs * (g[j].x - m1 - xh[j].x * m2), up[j].y + rs * (g[j].y - m1 - xh[j].y * m2),
                                     up[j].z + rs * (g[j].z - m1 - xh[j].z * m2), up[j].w + rs * (g[j].w - m1 - xh[j].w * m2));
        *reinterpret_cast<float4 *>(dsup + row * C + c) = d;
        *reinterpret_cast<uint2 *>(df + row * C + c) = pack_bf16x4(d.x * sc, d.y * sc, d.z * sc, d.w * sc);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    *reinterpret_cast<float4 *>(&red[wave][0][4 * (lane + 64 * j)]) = ag[j];
    *reinterpret_cast<float4 *>(&red[wave][1][4 * (lane + 64 * j)]) = ab[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 64 * WV) {
    const int k = i / C, c = i - k * C;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < WV; ++w) sum += red[w][k][c];
    atomicAdd((k ? dbeta : dgamma) + c, sum);
  }
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
                              void *y_q, void *y_s, int q_format, void *stream_)
{
  int rc = check(images, L, C, "pd_swin_ln_fwd");
  if (rc) return rc;
  if ((rc = check_mx(y_q, y_s, q_format, "pd_swin_ln_fwd")) != PD_OK) return rc;
  const int64_t R = (int64_t)images * L;
  if (R == 0) return PD_OK;
  if (!x || !gamma || !beta || !y || !mean || !rstd || (r && !s) || (n_zero > 0 && !zero_rows) || n_zero < 0)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_swin_ln_fwd: null pointer");
  if (((uintptr_t)x | (uintptr_t)s | (uintptr_t)gamma | (uintptr_t)beta) & 15 || ((uintptr_t)r | (uintptr_t)y) & 7)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_swin_ln_fwd: misaligned pointer (fp32 rows 16 bytes, bf16 rows 8)");
  const int64_t rows = R + (int64_t)images * n_zero;
  const dim3 g((unsigned)((rows + WAVES - 1) / WAVES)), b(64 * WAVES);
  hipStream_t st = (hipStream_t)stream_;
#define FWD(MXF) hipLaunchKernelGGL((ln_fwd<E, MXF>), g, b, 0, st, x, (const bf16_t *)r, rmap, r_rows, rscale, gamma, beta, eps, s, (bf16_t *)y, ymap, \
                                    y_rows, zero_rows, n_zero, mean, rstd, images, L, (uint8_t *)y_q, (uint8_t *)y_s)
  E_SWITCH(C, if (!y_q && E <= 3) hipLaunchKernelGGL((ln_fwd_narrow<E>), g, b, 0, st, x, (const bf16_t *)r, rmap, r_rows, rscale, gamma, beta, eps, s,
                                                        (bf16_t *)y, ymap, y_rows, zero_rows, n_zero, mean, rstd, images, L);
              else if (!y_q) FWD(-1); else if (q_format == PD_MX8_E4M3) FWD(PD_MX8_E4M3); else FWD(PD_MX8_E5M2));
#undef FWD
  return pd_check_launch("pd_swin_ln_fwd");
}

extern "C" int pd_swin_ln_bwd(const void *dy, const int32_t *ymap, int y_rows, const float *dsup, const float *s, const float *mean,
                              const float *rstd, const float *gamma, float *ds, void *dr, const int32_t *rmap, int r_rows,
                              const float *rscale, const int32_t *zero_rows, int n_zero, float *dgamma, float *dbeta, int images,
                              int L, int C, void *dr_q, void *dr_s, int q_format, int n_rep, int64_t rep_stride, void *stream_)
{
  if (n_rep < 1 || (n_rep > 1 && rep_stride < C)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_swin_ln_bwd: n_rep >= 1 copies at least C floats apart");
  int rc = check(images, L, C, "pd_swin_ln_bwd");
  if (rc) return rc;
  if ((rc = check_mx(dr_q, dr_s, q_format, "pd_swin_ln_bwd")) != PD_OK) return rc;
  if (dr_q && !dr) return pd_set_error(PD_ERR_INVALID_ARG, "pd_swin_ln_bwd: the MX copy of dr needs dr");
  const int64_t R = (int64_t)images * L;
  if (R == 0) return PD_OK;
  if (!dy || !s || !mean || !rstd || !gamma || !ds || !dgamma || !dbeta || (n_zero > 0 && !zero_rows) || n_zero < 0)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_swin_ln_bwd: null pointer");
  if (((uintptr_t)s | (uintptr_t)dsup | (uintptr_t)ds | (uintptr_t)gamma) & 15 || ((uintptr_t)dy | (uintptr_t)dr) & 7)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_swin_ln_bwd: misaligned pointer (fp32 rows 16 bytes, bf16 rows 8)");
  const int64_t rows = R + (dr ? (int64_t)images * n_zero : 0);
  hipStream_t st = (hipStream_t)stream_;
  dim3 g, b;
  int rpw = 0;
#define BWD(MXF) hipLaunchKernelGGL((ln_bwd<E, MXF>), g, b, 0, st, (const bf16_t *)dy, ymap, y_rows, dsup, s, mean, rstd, gamma, ds, (bf16_t *)dr, rmap, \
                                    r_rows, rscale, zero_rows, n_zero, dgamma, dbeta, images, L, rpw, (uint8_t *)dr_q, (uint8_t *)dr_s, g_swin_ln_abl, n_rep, rep_stride)
  E_SWITCH(C, bwd_geometry<E>(rows, g, b, rpw);
              if (!dr_q && E <= 3) hipLaunchKernelGGL((ln_bwd_narrow<E>), g, b, 0, st, (const bf16_t *)dy, ymap, y_rows, dsup, s, mean, rstd, gamma, ds, (bf16_t *)dr,
                                                      rmap, r_rows, rscale, zero_rows, n_zero, dgamma, dbeta, images, L, rpw, n_rep, rep_stride);
              else if (!dr_q) BWD(-1); else if (q_format == PD_MX8_E4M3) BWD(PD_MX8_E4M3); else BWD(PD_MX8_E5M2));
#undef BWD
  return pd_check_launch("pd_swin_ln_bwd");
}

extern "C" int pd_layernorm_rows_f32_fwd(const float *x, const float *gamma, const float *beta, float eps, float *y, float *mean, float *rstd,
                                         int64_t rows, int C, void *stream_)
{
  if (rows < 0 || C <= 0 || (C & 3) || C > 3072) return pd_set_error(PD_ERR_INVALID_ARG, "pd_layernorm_rows_f32_fwd: rows=%lld C=%d (a multiple of 4 up to 3072)", (long long)rows, C);
  if (rows == 0) return PD_OK;
  if (!x || !gamma || !beta || !y || !mean || !rstd || (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_layernorm_rows_f32_fwd: null / misaligned pointer");
  const dim3 g((unsigned)((rows + 3) / 4)), b(256);
  hipStream_t st = (hipStream_t)stream_;
  switch ((C + 255) / 256) {
    case 1: hipLaunchKernelGGL(ln_rows_f32_fwd<1>, g, b, 0, st, x, gamma, beta, eps, y, mean, rstd, rows, C); break;
    case 2: hipLaunchKernelGGL(ln_rows_f32_fwd<2>, g, b, 0, st, x, gamma, beta, eps, y, mean, rstd, rows, C); break;
    case 3: hipLaunchKernelGGL(ln_rows_f32_fwd<3>, g, b, 0, st, x, gamma, beta, eps, y, mean, rstd, rows, C); break;
    case 4: hipLaunchKernelGGL(ln_rows_f32_fwd<4>, g, b, 0, st, x, gamma, beta, eps, y, mean, rstd, rows, C); break;
    case 5: case 6: hipLaunchKernelGGL(ln_rows_f32_fwd<6>, g, b, 0, st, x, gamma, beta, eps, y, mean, rstd, rows, C); break;
    case 7: case 8: hipLaunchKernelGGL(ln_rows_f32_fwd<8>, g, b, 0, st, x, gamma, beta, eps, y, mean, rstd, rows, C); break;
    default: hipLaunchKernelGGL(ln_rows_f32_fwd<12>, g, b, 0, st, x, gamma, beta, eps, y, mean, rstd, rows, C); break;
  }
  return pd_check_launch("pd_layernorm_rows_f32_fwd");
}

extern "C" int pd_layernorm_rows_f32_bwd(const float *dy, const float *x, const float *mean, const float *rstd, const float *gamma, float *dx,
                                         float *dgamma, float *dbeta, int64_t rows, int C, void *stream_)
{
  if (rows < 0 || C <= 0 || (C & 3) || C > 3072) return pd_set_error(PD_ERR_INVALID_ARG, "pd_layernorm_rows_f32_bwd: rows=%lld C=%d (a multiple of 4 up to 3072)", (long long)rows, C);
  if (rows == 0) return PD_OK;
  if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)gamma) & 15))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_layernorm_rows_f32_bwd: null / misaligned pointer");
  int64_t nb = (rows + 31) / 32;                                  // >= 4 rows per wavefront, at most 256 workgroups
  const dim3 g((unsigned)(nb < 1 ? 1 : (nb > 256 ? 256 : nb)));
  const dim3 b(C > 2048 ? 128 : (C > 768 ? 256 : 512));
  hipStream_t st = (hipStream_t)stream_;
  switch ((C + 255) / 256) {
    case 1: hipLaunchKernelGGL(ln_rows_f32_bwd<1>, g, b, 0, st, dy, x, mean, rstd, gamma, dx, dgamma, dbeta, rows, C); break;
    case 2: hipLaunchKernelGGL(ln_rows_f32_bwd<2>, g, b, 0, st, dy, x, mean, rstd, gamma, dx, dgamma, dbeta, rows, C); break;
    case 3: hipLaunchKernelGGL(ln_rows_f32_bwd<3>, g, b, 0, st, dy, x, mean, rstd, gamma, dx, dgamma, dbeta, rows, C); break;
    case 4: hipLaunchKernelGGL(ln_rows_f32_bwd<4>, g, b, 0, st, dy, x, mean, rstd, gamma, dx, dgamma, dbeta, rows, C); break;
    case 5: case 6: hipLaunchKernelGGL(ln_rows_f32_bwd<6>, g, b, 0, st, dy, x, mean, rstd, gamma, dx, dgamma, dbeta, rows, C); break;
    case 7: case 8: hipLaunchKernelGGL(ln_rows_f32_bwd<8>, g, b, 0, st, dy, x, mean, rstd, gamma, dx, dgamma, dbeta, rows, C); break;
    default: hipLaunchKernelGGL(ln_rows_f32_bwd<12>, g, b, 0, st, dy, x, mean, rstd, gamma, dx, dgamma, dbeta, rows, C); break;
  }
  return pd_check_launch("pd_layernorm_rows_f32_bwd");
}

#define NQ_SWITCH(C4, BODY)                                           \
  switch (((C4) + 255) / 256) {                                       \
    case 1: { constexpr int NQ = 1; BODY; } break;                    \
    case 2: { constexpr int NQ = 2; BODY; } break;                    \
    case 3: { constexpr int NQ = 3; BODY; } break;                    \
    case 4: { constexpr int NQ = 4; BODY; } break;                    \
    case 5: case 6: { constexpr int NQ = 6; BODY; } break;            \
    case 7: case 8: { constexpr int NQ = 8; BODY; } break;            \
    default: { constexpr int NQ = 12; BODY; } break;                  \
  }

extern "C" int pd_swin_tail_ln_fwd(const float *cur, const void *r, const float *rscale, int L, const float *gamma, const float *beta, float eps, float *s,
                                   float *y, float *mean, float *rstd, int64_t rows, int C, void *stream_)
{
  if (rows < 0 || C <= 0 || (C & 3) || C > 3072 || L <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_swin_tail_ln_fwd: rows=%lld C=%d L=%d (C a multiple of 4 up to 3072)", (long long)rows, C, L);
  if (rows == 0) return PD_OK;
  if (!cur || !r || !gamma || !beta || !s || !y || !mean || !rstd || (((uintptr_t)cur | (uintptr_t)s | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) || ((uintptr_t)r & 7))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_swin_tail_ln_fwd: null / misaligned pointer");
  const dim3 g((unsigned)((rows + 3) / 4)), b(256);
  hipStream_t st = (hipStream_t)stream_;
  NQ_SWITCH(C, hipLaunchKernelGGL(ln_tail_fwd<NQ>, g, b, 0, st, cur, (const bf16_t *)r, rscale, L, gamma, beta, eps, s, y, mean, rstd, rows, C));
  return pd_check_launch("pd_swin_tail_ln_fwd");
}

extern "C" int pd_swin_tail_ln_bwd(const float *dy, const float *dsum, const float *s, const float *mean, const float *rstd, const float *gamma, const float *rscale,
                                   int L, float *dsup, void *df, float *dgamma, float *dbeta, int64_t rows, int C, void *stream_)
{
  if (rows < 0 || C <= 0 || (C & 3) || C > 3072 || L <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_swin_tail_ln_bwd: rows=%lld C=%d L=%d (C a multiple of 4 up to 3072)", (long long)rows, C, L);
  if (rows == 0) return PD_OK;
  if (!dy || !s || !mean || !rstd || !gamma || !dsup || !df || !dgamma || !dbeta
      || (((uintptr_t)dy | (uintptr_t)dsum | (uintptr_t)s | (uintptr_t)dsup | (uintptr_t)gamma) & 15) || ((uintptr_t)df & 7))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_swin_tail_ln_bwd: null / misaligned pointer");
  int64_t nb = (rows + 31) / 32;
  const dim3 g((unsigned)(nb < 1 ? 1 : (nb > 256 ? 256 : nb)));
  const dim3 b(C > 2048 ? 128 : (C > 768 ? 256 : 512));
  hipStream_t st = (hipStream_t)stream_;
  NQ_SWITCH(C, hipLaunchKernelGGL(ln_tail_bwd<NQ>, g, b, 0, st, dy, dsum, s, mean, rstd, gamma, rscale, L, dsup, (bf16_t *)df, dgamma, dbeta, rows, C));
  return pd_check_launch("pd_swin_tail_ln_bwd");
}

static int merge_check(int B, int H, int W, int C, const char *who)
{
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || (H & 1) || (W & 1) || (C & 3) || 4 * C > 3072)
    return pd_set_error(PD_ERR_INVALID_ARG, "%s: B=%d H=%d W=%d C=%d (even H and W, C a multiple of 4, 4 C <= 3072)", who, B, H, W, C);
  return PD_OK;
}

extern "C" int pd_swin_merge_ln_fwd(const float *x, const float *gamma, const float *beta, float eps, void *y, float *mean, float *rstd, int B, int H, int W,
                                    int C, void *stream_)
{
  int rc = merge_check(B, H, W, C, "pd_swin_merge_ln_fwd");
  if (rc) return rc;
  const int64_t rows = (int64_t)B * (H / 2) * (W / 2);
  if (rows == 0) return PD_OK;
  if (!x || !gamma || !beta || !y || !mean || !rstd || (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta) & 15) || ((uintptr_t)y & 7))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_swin_merge_ln_fwd: null / misaligned pointer");
  const dim3 g((unsigned)((rows + 3) / 4)), b(256);
  hipStream_t st = (hipStream_t)stream_;
  NQ_SWITCH(4 * C, hipLaunchKernelGGL(ln_merge_fwd<NQ>, g, b, 0, st, x, gamma, beta, eps, (bf16_t *)y, mean, rstd, rows, H, W, C));
  return pd_check_launch("pd_swin_merge_ln_fwd");
}

extern "C" int pd_swin_merge_ln_bwd(const void *dy, const float *x, const float *mean, const float *rstd, const float *gamma, float *dx, float *dgamma,
                                    float *dbeta, int B, int H, int W, int C, void *stream_)
{
  int rc = merge_check(B, H, W, C, "pd_swin_merge_ln_bwd");
  if (rc) return rc;
  const int64_t rows = (int64_t)B * (H / 2) * (W / 2);
  if (rows == 0) return PD_OK;
  if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma || !dbeta || (((uintptr_t)x | (uintptr_t)dx | (uintptr_t)gamma) & 15) || ((uintptr_t)dy & 7))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_swin_merge_ln_bwd: null / misaligned pointer");
  int64_t nb = (rows + 31) / 32;                                  // >= 4 rows per wavefront, at most 256 workgroups
  const dim3 g((unsigned)(nb < 1 ? 1 : (nb > 256 ? 256 : nb)));
  const dim3 b(4 * C > 2048 ? 128 : (4 * C > 768 ? 256 : 512));
  hipStream_t st = (hipStream_t)stream_;
  NQ_SWITCH(4 * C, hipLaunchKernelGGL(ln_merge_bwd<NQ>, g, b, 0, st, (const bf16_t *)dy, x, mean, rstd, gamma, dx, dgamma, dbeta, rows, H, W, C));
  return pd_check_launch("pd_swin_merge_ln_bwd");
}

