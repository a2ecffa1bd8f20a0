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
//                       K oversampled points, then a scan-ordered compaction of the chosen coordinates (criterion.py:72-88, 181-189)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mfma_bf16.h"
#include "pd_common.h"
#include "pd_criterion.h"
#include "pd_msda.h"

int g_crit_abl = 0;                                  // pd_debug_set "crit_abl" (tools/bench_criterion_ops.py only): phases of uncertain_points removed

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
constexpr int QR = 4, MT = 512;               // query rows per workgroup, threads; JT = target columns per pass (4: two workgroups per CU)

// softplus(v) and sigmoid(v) from ONE exponential: e = exp(-|v|), sigmoid = v >= 0 ? 1 / (1 + e) : e / (1 + e), softplus = max(v, 0) +
// log(1 + e) (torch's threshold-20 branch returns v there, which differs from this by < 2.1e-9).  The hardware exp2 / log2 behind __expf /
// __logf are good to ~1e-6 relative / 1e-7 absolute here — the sums run over 10^4 points of O(1) terms.
__device__ __forceinline__ void sp_sg(float v, float &sp, float &sg)
{
  const float e = __expf(-fabsf(v));
  const float r = __frcp_rn(1.f + e);
  sg = v >= 0.f ? r : e * r;
  sp = fmaxf(v, 0.f) + __logf(1.f + e);
}

template <typename T, int VEC, int JT>
__global__ __launch_bounds__(MT, JT == 4 ? 2 : 1) void matcher_costs(const T *__restrict__ x, const float *__restrict__ t, int64_t t_sb, int64_t t_sd,
                                                    int64_t t_sj, const float *__restrict__ prob, const int64_t *__restrict__ labels,
                                                    float *__restrict__ cost, int heads, int Q, int n, int nt, int classes, float w_mask,
                                                    float w_class, float w_dice)
{
  constexpr int NACC = QR * (2 + 2 * JT) + JT;
  __shared__ float red[MT / 64][NACC];
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
    for (int i = threadIdx.x * VEC; i < n; i += MT * VEC) {
      float tv[JT][VEC], xv[QR][VEC];
#pragma unroll
      for (int j = 0; j < JT; ++j) {                               // all loads of the step first: one memory latency per step
        if (j0 + j < nt) {
          if constexpr (VEC == 2) { const float2 u = *reinterpret_cast<const float2 *>(tp + (int64_t)(j0 + j) * t_sj + i); tv[j][0] = u.x; tv[j][1] = u.y; }
          else tv[j][0] = tp[(int64_t)(j0 + j) * t_sj + i];
        } else {
#pragma unroll
          for (int u = 0; u < VEC; ++u) tv[j][u] = 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < QR; ++r) {
        const T *xr = xp + (int64_t)(q0 + r < Q ? r : 0) * n + i;   // rows beyond Q re-read row 0 (their results are not stored)
        if constexpr (sizeof(T) == 2) {
          if constexpr (VEC == 2) { const unsigned u = *reinterpret_cast<const unsigned *>(xr); xv[r][0] = bf2f((bf16_t)(u & 0xffffu)); xv[r][1] = bf2f((bf16_t)(u >> 16)); }
          else xv[r][0] = bf2f(xr[0]);
        } else {
          if constexpr (VEC == 2) { const float2 u = *reinterpret_cast<const float2 *>(xr); xv[r][0] = u.x; xv[r][1] = u.y; }
          else xv[r][0] = xr[0];
        }
      }
#pragma unroll
      for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int u = 0; u < VEC; ++u) ts[j] += tv[j][u];
#pragma unroll
      for (int r = 0; r < QR; ++r)
#pragma unroll
        for (int u = 0; u < VEC; ++u) {
          float a, s;
          sp_sg(xv[r][u], a, s);
          sp[r] += a;
          sg[r] += s;
#pragma unroll
          for (int j = 0; j < JT; ++j) { xt[r][j] = fmaf(xv[r][u], tv[j][u], xt[r][j]); st[r][j] = fmaf(s, tv[j][u], st[r][j]); }
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
        auto tot = [&](int k) {
          float a = 0.f;
#pragma unroll
          for (int w = 0; w < MT / 64; ++w) a += red[w][k];
          return a;
        };
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
constexpr int LT = 1024;                          // threads per mask in the forward: all of a row's loads in flight at once
template <int VEC>
__global__ __launch_bounds__(LT) void mask_point_losses_fwd(const float *__restrict__ x, const float *__restrict__ y,
                                                            float *__restrict__ bce, float *__restrict__ dice, float *__restrict__ stats, int n)
{
  __shared__ float red[LT / 64][4];
  const float *xr = x + (int64_t)blockIdx.x * n, *yr = y + (int64_t)blockIdx.x * n;
  float a = 0.f, sy = 0.f, s1 = 0.f, y1 = 0.f;
  for (int i = threadIdx.x * VEC; i < n; i += LT * VEC) {
    float v[VEC], t[VEC];
    if constexpr (VEC == 4) {
      const float4 a4 = *reinterpret_cast<const float4 *>(xr + i), b4 = *reinterpret_cast<const float4 *>(yr + i);
      v[0] = a4.x; v[1] = a4.y; v[2] = a4.z; v[3] = a4.w; t[0] = b4.x; t[1] = b4.y; t[2] = b4.z; t[3] = b4.w;
    } else { v[0] = xr[i]; t[0] = yr[i]; }
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
      const float e = expf(-fabsf(v[u]));
      a += fmaxf(v[u], 0.f) - v[u] * t[u] + log1pf(e);
      const float r = 1.f / (1.f + e);
      const float s = v[u] >= 0.f ? r : e * r;
      sy = fmaf(s, t[u], sy); s1 += s; y1 += t[u];
    }
  }
  a = wave_sum(a); sy = wave_sum(sy); s1 = wave_sum(s1); y1 = wave_sum(y1);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wave][0] = a; red[wave][1] = sy; red[wave][2] = s1; red[wave][3] = y1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    auto tot = [&](int k) {
      float r = 0.f;
#pragma unroll
      for (int w = 0; w < LT / 64; ++w) r += red[w][k];
      return r;
    };
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
constexpr int UT = 1024, UV = PD_UNCERTAIN_MAX_K / UT;            // threads per row, keys per thread

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

// adds 1 to hist[bin] for every active lane.  The first digit of |logit| is its exponent: a handful of bins take nearly every key, and a
// ds_add with 64 lanes on one address is serialised — so the lanes that share the first active lane's bin are counted with a ballot and
// added once (up to PEEL rounds), whoever is left adds for itself.
template <int PEEL>
__device__ __noinline__ void hist_add(int *hist, int bin, bool active)
{
#pragma unroll
  for (int r = 0; r < PEEL; ++r) {
    const unsigned long long act = __ballot(active);
    if (!act) return;
    const int leader = __ffsll((long long)act) - 1;
    const int lb = __shfl(bin, leader, 64);
    const unsigned long long same = __ballot(active && bin == lb);
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[lb], __popcll(same));
    active = active && bin != lb;
  }
  if (active) atomicAdd(&hist[bin], 1);
}

__global__ __launch_bounds__(UT) void uncertain_points(const float *__restrict__ logits, const float2 *__restrict__ coords,
                                                       const float2 *__restrict__ rnd, float2 *__restrict__ out, int K, int k, int n_random, int abl)
{
  __shared__ int hist[256];
  __shared__ int wsum[UT / 64];
  __shared__ unsigned s_prefix;
  __shared__ int s_remaining;
  const int row = blockIdx.x, tid = threadIdx.x;
  const float *lr = logits + (int64_t)row * K;
  const int per = (K + UT - 1) / UT;                               // <= UV; thread tid owns points tid, tid + UT, ... (coalesced)
  unsigned key[UV];
#pragma unroll
  for (int m = 0; m < UV; ++m) {
    const int i = m * UT + tid;
    const unsigned bits = __float_as_uint(lr[i < K ? i : K - 1]);                           // unconditional (clamped) loads: all in flight at once
    key[m] = (m < per && i < K) ? (bits & 0x7fffffffu) : 0xffffffffu;                       // |logit| as an ordered integer; padding sorts last
  }
  unsigned prefix = 0, mask = 0;
  int remaining = k;
#pragma unroll 1
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    if (shift == 24 && !(abl & 1)) {
#pragma unroll
      for (int m = 0; m < UV; ++m)
        if (m < per) hist_add<4>(hist, (int)(key[m] >> 24), key[m] != 0xffffffffu);
    } else if (abl & 2) {
      if (tid == 0) hist[(prefix >> shift) & 255 ? 1 : 1] = K;     // (timing only) no counting at all
    } else {
#pragma unroll
      for (int m = 0; m < UV; ++m)
        if (key[m] != 0xffffffffu && (key[m] & mask) == prefix) atomicAdd(&hist[(key[m] >> shift) & 255], 1);
    }
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
  // prefix = the k-th smallest key T; every smaller key is taken, and `remaining` of the keys equal to T — the first in the (thread, round)
  // order of this kernel, a fixed choice.  Output order: the smaller keys by (thread, round), then the ties.
  int nless = 0, neq = 0;
#pragma unroll
  for (int m = 0; m < UV; ++m) { nless += key[m] < prefix; neq += key[m] == prefix; }
  int tot_less, tot_eq;
  const int pos_less = block_excl_scan(nless, wsum, tot_less);
  const int pos_eq = block_excl_scan(neq, wsum, tot_eq);
  float2 *orow = out + (int64_t)row * (k + n_random);
  const float2 *crow = coords + (int64_t)row * K;
  int a = pos_less, e = pos_eq;
  if (abl & 4) return;                                             // (timing only) no compaction
#pragma unroll
  for (int m0 = 0; m0 < UV; m0 += 8) {                             // eight coordinate loads in flight, then the (conditional) stores:
    float2 c[8];                                                   // a load inside the branch would be one memory round trip per point
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = (m0 + u) * UT + tid;
      c[u] = crow[i < K ? i : K - 1];                              // (clamped, not predicated: a predicated load is its own basic block)
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (key[m0 + u] < prefix) orow[a++] = c[u];
      else if (key[m0 + u] == prefix) { if (e < remaining) orow[tot_less + e] = c[u]; ++e; }
    }
  }
  for (int i = tid; i < n_random; i += UT) orow[k + i] = rnd[(int64_t)row * n_random + i];
}
// ------------------------------------------------------------------------------------------------ target masks at points
// F.grid_sample(bilinear, zeros, align_corners=False) of one-byte 0 / 1 masks (the padded target masks as stored: no fp32 copy of them)
// at per-row points: row r reads map map_idx[r] (or r) at the coordinates of row r / coords_div.  One lane per point; the arithmetic is
// the planar sampler's (rowwise.hip), the four corners are bytes.
__global__ __launch_bounds__(256) void point_sample_u8(const uint8_t *__restrict__ maps, const int64_t *__restrict__ map_idx,
                                                       const float *__restrict__ coords, float *__restrict__ out, int64_t total, int P, int H,
                                                       int W, int coords_div)
{
  for (int64_t pt = (int64_t)blockIdx.x * 256 + threadIdx.x; pt < total; pt += (int64_t)gridDim.x * 256) {
    const int64_t r = pt / P;
    const int p = (int)(pt - r * P);
    const int64_t m = map_idx ? map_idx[r] : r;
    const float2 c = *reinterpret_cast<const float2 *>(coords + ((r / coords_div) * P + p) * 2);
    const float gx = 2.0f * c.x - 1.0f, gy = 2.0f * c.y - 1.0f;
    const float ix = ((gx + 1.f) * W - 1.f) / 2.f, iy = ((gy + 1.f) * H - 1.f) / 2.f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wnw = (x1 - ix) * (y1 - iy), wne = (ix - x0) * (y1 - iy), wsw = (x1 - ix) * (iy - y0), wse = (ix - x0) * (iy - y0);
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    const uint8_t *mp = maps + m * (int64_t)H * W;
    // the four corner bytes as straight-line loads from clamped (always valid) addresses, a corner outside the map contributing an exact 0 —
    // behind `if (valid)` each load was followed by its own s_waitcnt: four dependent memory round trips per point
    const int xa = min(max(x0, 0), W - 1), xb = min(max(x1, 0), W - 1), ya = min(max(y0, 0), H - 1), yb = min(max(y1, 0), H - 1);
    const uint8_t b00 = mp[(int64_t)ya * W + xa], b01 = mp[(int64_t)ya * W + xb], b10 = mp[(int64_t)yb * W + xa], b11 = mp[(int64_t)yb * W + xb];
    float acc = 0.f;
    acc += ((vy0 && vx0 && b00) ? 1.f : 0.f) * wnw;
    acc += ((vy0 && vx1 && b01) ? 1.f : 0.f) * wne;
    acc += ((vy1 && vx0 && b10) ? 1.f : 0.f) * wsw;
    acc += ((vy1 && vx1 && b11) ? 1.f : 0.f) * wse;
    out[pt] = acc;
  }
}

// ------------------------------------------------------------------------------------------------ matcher point logits, fused
// The matcher's logits of all Q masks at its random points (reference matcher.py:108-125: point_sample(out_mask, point_coords) per image
// and head) are, by the linearity of bilinear sampling, mask_embed . point_sample(mask_features): rounds 2-5 ran that as a sampler launch
// (250 880 points x 256 channels: 1 GB of corner rows gathered, 128 MB of bf16 samples written: 159 us) followed by a batched skinny GEMM that
// read them back with 32 x 128 tiles at one workgroup per CU (143 us for 12.8 GFLOP).  Here a workgroup samples 64 points of one (image, head)
// problem straight into LDS as the bf16 MFMA operand (a wavefront per point: each corner one 1 KB burst, two points = eight loads in flight
// per lane) and multiplies them with the problem's query embeddings, which each wavefront holds in registers as the other operand (32 queries x
// 256 channels = 64 VGPRs); the samples never leave the CU.  Arithmetic as the two launches: fp32 bilinear sum in the sampler's order, one
// rounding to bf16, v_mfma_f32_32x32x8_bf16_1k over k = 0 .. 255 in order, one rounding of the logit to bf16 — the results are bit-identical.
constexpr int MPL_PTS = 64, MPL_C = 256, MPL_PITCH = 260;    // 520-byte LDS rows: the 8-byte fragment reads of 32 rows touch every bank once
template <int TPW>                                           // point tiles per workgroup (the embeddings' fragments are loaded once)
__global__ __launch_bounds__(256, 4) void match_point_logits(const float *__restrict__ feat, const float *__restrict__ coords,
                                                             const bf16_t *__restrict__ emb, bf16_t *__restrict__ out, int heads, int Q, int Pm,
                                                             int H, int W)
{
  using namespace pdmfma;
  __shared__ __attribute__((aligned(16))) bf16_t Fs[MPL_PTS][MPL_PITCH];
  const int prob = blockIdx.y, b = prob / heads;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  const int q = 32 * wave + r;
  bf16x4 ef[32];                                             // Y operand: query q, channels 8 s + 4 hh .. + 3
  {
    const bf16_t *e = emb + ((int64_t)prob * Q + min(q, Q - 1)) * MPL_C + 4 * hh;
#pragma unroll
    for (int s = 0; s < 32; ++s) ef[s] = *reinterpret_cast<const bf16x4 *>(e + 8 * s);
  }
  const float *base = feat + (int64_t)b * H * W * MPL_C + lane * 4;
  const float *cr = coords + (int64_t)prob * Pm * 2;
  for (int tile = 0; tile < TPW; ++tile) {
    const int p0 = ((int)blockIdx.x * TPW + tile) * MPL_PTS;
    if (p0 >= Pm) break;
    if (tile) __syncthreads();                               // the previous tile's fragment reads are done
#pragma unroll 1
    for (int i = 0; i < 16; i += 2) {
      float4 v[2][4];
      float wt[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int pt = min(p0 + 16 * wave + i + u, Pm - 1);
        // the exact arithmetic of point_sample_nhwc (rowwise.hip) / the torch path: g = 2 c - 1, then ((g + 1) size - 1) / 2
        const float gx = 2.0f * cr[pt * 2] - 1.0f, gy = 2.0f * cr[pt * 2 + 1] - 1.0f;
        const float ix = ((gx + 1.f) * W - 1.f) / 2.f, iy = ((gy + 1.f) * H - 1.f) / 2.f;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
        const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
        // a corner outside the map: weight 0 on a clamped (valid) address, so that all eight loads are straight-line code
        wt[u][0] = vy0 && vx0 ? (x1 - ix) * (y1 - iy) : 0.f; wt[u][1] = vy0 && vx1 ? (ix - x0) * (y1 - iy) : 0.f;
        wt[u][2] = vy1 && vx0 ? (x1 - ix) * (iy - y0) : 0.f; wt[u][3] = vy1 && vx1 ? (ix - x0) * (iy - y0) : 0.f;
        const int xa = min(max(x0, 0), W - 1), xb = min(max(x1, 0), W - 1), ya = min(max(y0, 0), H - 1), yb = min(max(y1, 0), H - 1);
        v[u][0] = *reinterpret_cast<const float4 *>(base + ((int64_t)ya * W + xa) * MPL_C);
        v[u][1] = *reinterpret_cast<const float4 *>(base + ((int64_t)ya * W + xb) * MPL_C);
        v[u][2] = *reinterpret_cast<const float4 *>(base + ((int64_t)yb * W + xa) * MPL_C);
        v[u][3] = *reinterpret_cast<const float4 *>(base + ((int64_t)yb * W + xb) * MPL_C);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc.x += v[u][c].x * wt[u][c]; acc.y += v[u][c].y * wt[u][c]; acc.z += v[u][c].z * wt[u][c]; acc.w += v[u][c].w * wt[u][c];
        }
        *reinterpret_cast<bf16x4 *>(&Fs[16 * wave + i + u][lane * 4]) = pack4(acc.x, acc.y, acc.z, acc.w);
      }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int s = 0; s < 32; ++s) mma(acc, lds4(&Fs[32 * t + r][8 * s + 4 * hh]), ef[s]);   // acc[point][query] (mfma_bf16.h)
      if (q < Q) {
        bf16_t *o = out + ((int64_t)prob * Q + q) * Pm + p0 + 32 * t + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (p0 + 32 * t + 8 * g + 4 * hh < Pm) *reinterpret_cast<bf16x4 *>(o + 8 * g) = pack4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
      }
    }
  }
}
}  // namespace

extern "C" int pd_match_point_logits(const float *feat_nhwc, const float *coords, const void *emb, void *out, int B, int heads, int Q, int Pm,
                                     int H, int W, int C, void *stream_)
{
  if (B < 0 || heads <= 0 || Q <= 0 || Q > 128 || Pm <= 0 || (Pm & 3) || H <= 0 || W <= 0 || C != MPL_C)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_match_point_logits: B=%d heads=%d Q=%d (<= 128) points=%d (%% 4) H=%d W=%d C=%d (256)", B, heads, Q, Pm, H, W, C);
  if (B == 0) return PD_OK;
  if (!feat_nhwc || !coords || !emb || !out) return pd_set_error(PD_ERR_INVALID_ARG, "pd_match_point_logits: null pointer");
  constexpr int TPW = 2;
  hipLaunchKernelGGL(match_point_logits<TPW>, dim3((unsigned)((Pm + MPL_PTS * TPW - 1) / (MPL_PTS * TPW)), (unsigned)(B * heads)), dim3(256), 0,
                     (hipStream_t)stream_, feat_nhwc, coords, (const bf16_t *)emb, (bf16_t *)out, heads, Q, Pm, H, W);
  return pd_check_launch("pd_match_point_logits");
}

// ------------------------------------------------------------------------------------------------ the three loss vectors
// One workgroup per prediction head h (criterion order; d = d_of_h[h] its place in the decoder's stack): the weighted cross entropy of
// the head's B Q queries (F.cross_entropy with class weights: sum w[t] nll / sum w[t], criterion.py:126-145) and the sums of its matched
// masks' BCE / dice terms over num_masks (criterion.py:203-206).  Was log_softmax + nll_loss + two reductions + an index + a division, and
// two reductions + two divisions, and their ~20 backward launches.
namespace {
__device__ __forceinline__ float block_sum_256(float v, float *red)
{
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void loss_vectors_fwd(const float *__restrict__ logits, int64_t sb, int64_t sd, const int64_t *__restrict__ tclass,
                                                        const float *__restrict__ w, const int64_t *__restrict__ d_of_h, const float *__restrict__ bce,
                                                        const float *__restrict__ dice, const float *__restrict__ num_masks, float *__restrict__ vec,
                                                        float *__restrict__ lse, float *__restrict__ den, int B, int H, int Q, int K1, int Nh)
{
  __shared__ float red[4];
  const int h = blockIdx.x, d = (int)d_of_h[h];
  float num = 0.f, dn = 0.f;
  for (int i = threadIdx.x; i < B * Q; i += 256) {
    const int b = i / Q, q = i - b * Q;
    const float *x = logits + b * sb + d * sd + (int64_t)q * K1;
    const int64_t at = ((int64_t)b * H + d) * Q + q;
    const int t = (int)tclass[at];
    float m = x[0];
    for (int k = 1; k < K1; ++k) m = fmaxf(m, x[k]);
    float sum = 0.f;
    for (int k = 0; k < K1; ++k) sum += expf(x[k] - m);
    const float l = m + logf(sum), wt = w[t];
    lse[at] = l;
    num += wt * (l - x[t]);
    dn += wt;
  }
  num = block_sum_256(num, red);
  dn = block_sum_256(dn, red);
  float sm = 0.f, sdc = 0.f;
  for (int n = threadIdx.x; n < Nh; n += 256) sm += bce[(int64_t)h * Nh + n], sdc += dice[(int64_t)h * Nh + n];
  sm = block_sum_256(sm, red);
  sdc = block_sum_256(sdc, red);
  if (threadIdx.x == 0) {
    const float nm = *num_masks;
    vec[h] = num / dn;
    vec[H + h] = sm / nm;
    vec[2 * H + h] = sdc / nm;
    den[d] = dn;
  }
}

// d logits[b, d, q, k] = dvec[0, h] / den[d] * w[t] (softmax_k - [k == t]);  d bce[n] = dvec[1, h(n)] / num_masks, d dice likewise
__global__ __launch_bounds__(256) void loss_vectors_bwd(const float *__restrict__ logits, int64_t sb, int64_t sd, const int64_t *__restrict__ tclass,
                                                        const float *__restrict__ w, const int64_t *__restrict__ d_of_h, const float *__restrict__ num_masks,
                                                        const float *__restrict__ lse, const float *__restrict__ den, const float *__restrict__ dvec,
                                                        float *__restrict__ d_logits, float *__restrict__ d_bce, float *__restrict__ d_dice, int B, int H,
                                                        int Q, int K1, int Nh)
{
  const int h = blockIdx.x, d = (int)d_of_h[h];
  const float gce = dvec[h] / den[d], nm = *num_masks;
  for (int i = threadIdx.x; i < B * Q; i += 256) {
    const int b = i / Q, q = i - b * Q;
    const int64_t off = b * sb + d * sd + (int64_t)q * K1, at = ((int64_t)b * H + d) * Q + q;
    const int t = (int)tclass[at];
    const float c = gce * w[t], l = lse[at];
    for (int k = 0; k < K1; ++k) d_logits[off + k] = c * (expf(logits[off + k] - l) - (k == t ? 1.f : 0.f));
  }
  const float gm = dvec[H + h] / nm, gd = dvec[2 * H + h] / nm;
  for (int n = threadIdx.x; n < Nh; n += 256) d_bce[(int64_t)h * Nh + n] = gm, d_dice[(int64_t)h * Nh + n] = gd;
}
}  // namespace

extern "C" int pd_loss_vectors_fwd(const float *logits, int64_t image_stride, int64_t head_stride, const int64_t *tclass, const float *class_weight,
                                   const int64_t *d_of_h, const float *bce, const float *dice, const float *num_masks, float *vec, float *lse, float *den,
                                   int B, int H, int Q, int K1, int Nh, void *stream_)
{
  if (B <= 0 || H <= 0 || Q <= 0 || K1 <= 0 || Nh < 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_loss_vectors_fwd: B=%d H=%d Q=%d K1=%d Nh=%d", B, H, Q, K1, Nh);
  if (!logits || !tclass || !class_weight || !d_of_h || !num_masks || !vec || !lse || !den || (Nh && (!bce || !dice)))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_loss_vectors_fwd: null pointer");
  hipLaunchKernelGGL(loss_vectors_fwd, dim3((unsigned)H), dim3(256), 0, (hipStream_t)stream_, logits, image_stride, head_stride, tclass, class_weight, d_of_h,
                     bce, dice, num_masks, vec, lse, den, B, H, Q, K1, Nh);
  return pd_check_launch("pd_loss_vectors_fwd");
}

extern "C" int pd_loss_vectors_bwd(const float *logits, int64_t image_stride, int64_t head_stride, const int64_t *tclass, const float *class_weight,
                                   const int64_t *d_of_h, const float *num_masks, const float *lse, const float *den, const float *dvec, float *d_logits,
                                   float *d_bce, float *d_dice, int B, int H, int Q, int K1, int Nh, void *stream_)
{
  if (B <= 0 || H <= 0 || Q <= 0 || K1 <= 0 || Nh < 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_loss_vectors_bwd: B=%d H=%d Q=%d K1=%d Nh=%d", B, H, Q, K1, Nh);
  if (!logits || !tclass || !class_weight || !d_of_h || !num_masks || !lse || !den || !dvec || !d_logits || (Nh && (!d_bce || !d_dice)))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_loss_vectors_bwd: null pointer");
  hipLaunchKernelGGL(loss_vectors_bwd, dim3((unsigned)H), dim3(256), 0, (hipStream_t)stream_, logits, image_stride, head_stride, tclass, class_weight, d_of_h,
                     num_masks, lse, den, dvec, d_logits, d_bce, d_dice, B, H, Q, K1, Nh);
  return pd_check_launch("pd_loss_vectors_bwd");
}

extern "C" int pd_point_sample_u8(const uint8_t *maps, const int64_t *map_idx, const float *coords, float *out, int rows, int P, int H, int W,
                                  int coords_div, void *stream_)
{
  if (rows < 0 || P < 0 || H <= 0 || W <= 0 || coords_div <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_point_sample_u8: rows=%d P=%d H=%d W=%d coords_div=%d", rows, P, H, W, coords_div);
  if (rows == 0 || P == 0) return PD_OK;
  if (!maps || !coords || !out) return pd_set_error(PD_ERR_INVALID_ARG, "pd_point_sample_u8: null pointer");
  const int64_t total = (int64_t)rows * P;
  const unsigned grid = (unsigned)((total + 255) / 256 > 65536 * 16 ? 65536 * 16 : (total + 255) / 256);
  hipLaunchKernelGGL(point_sample_u8, dim3(grid), dim3(256), 0, (hipStream_t)stream_, maps, map_idx, coords, out, total, P, H, W, coords_div);
  return pd_check_launch("pd_point_sample_u8");
}

extern "C" int pd_matcher_costs(const void *x, int dtype, const float *t, int64_t t_image_stride, int64_t t_head_stride,
                                int64_t t_target_stride, const float *prob, const int64_t *labels, float *cost, int problems, int heads, int Q,
                                int n, int n_targets, int classes, float w_mask, float w_class, float w_dice, void *stream_)
{
  if (problems < 0 || heads <= 0 || Q < 0 || n <= 0 || n_targets < 0 || classes <= 0 || (dtype != PD_F32 && dtype != PD_BF16) || (problems % heads))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_matcher_costs: problems=%d heads=%d Q=%d n=%d n_targets=%d classes=%d dtype=%d", problems, heads, Q, n,
                        n_targets, classes, dtype);
  if (problems == 0 || Q == 0 || n_targets == 0) return PD_OK;
  if (!x || !t || !prob || !labels || !cost) return pd_set_error(PD_ERR_INVALID_ARG, "pd_matcher_costs: null pointer");
  const dim3 g((unsigned)(problems * ((Q + QR - 1) / QR))), b(MT);
  // two points per lane when every row starts on an even element (8-byte target loads, 4- / 8-byte logit loads)
  const bool v2 = !(n & 1) && !((t_image_stride | t_head_stride | t_target_stride) & 1) && !((uintptr_t)t & 7) && !((uintptr_t)x & 7);
#define PD_MC(T, V, J) hipLaunchKernelGGL((matcher_costs<T, V, J>), g, b, 0, (hipStream_t)stream_, (const T *)x, t, t_image_stride, t_head_stride, \
                                          t_target_stride, prob, labels, cost, heads, Q, n, n_targets, classes, w_mask, w_class, w_dice)
#define PD_MCJ(T, V) do { if (n_targets <= 4 || (n_targets > 8 && n_targets <= 12)) PD_MC(T, V, 4); else PD_MC(T, V, 8); } while (0)
  if (dtype == PD_BF16) { if (v2) PD_MCJ(bf16_t, 2); else PD_MCJ(bf16_t, 1); }
  else { if (v2) PD_MCJ(float, 2); else PD_MCJ(float, 1); }
#undef PD_MCJ
#undef PD_MC
  return pd_check_launch("pd_matcher_costs");
}

extern "C" int pd_mask_point_losses_fwd(const float *x, const float *y, float *bce, float *dice, float *stats, int rows, int n, void *stream_)
{
  if (rows < 0 || n <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_mask_point_losses_fwd: rows=%d n=%d", rows, n);
  if (rows == 0) return PD_OK;
  if (!x || !y || !bce || !dice || !stats) return pd_set_error(PD_ERR_INVALID_ARG, "pd_mask_point_losses_fwd: null pointer");
  if (!(n & 3) && !(((uintptr_t)x | (uintptr_t)y) & 15))
    hipLaunchKernelGGL(mask_point_losses_fwd<4>, dim3(rows), dim3(LT), 0, (hipStream_t)stream_, x, y, bce, dice, stats, n);
  else
    hipLaunchKernelGGL(mask_point_losses_fwd<1>, dim3(rows), dim3(LT), 0, (hipStream_t)stream_, x, y, bce, dice, stats, n);
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
                     (float2 *)out, K, k, n_random, g_crit_abl);
  return pd_check_launch("pd_uncertain_points");
}
