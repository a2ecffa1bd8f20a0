// Masked multi-head attention for few queries x many keys (head_dim 32), forward + backward, gfx950.
//
// The decoder attends Q = 100..200 queries to up to 25 600 keys with a boolean mask: ~3.4 GFLOP per call at the
// largest level — nothing for the chip — but the library flash kernel tiles over QUERIES and leaves most CUs idle
// (230 us forward in profiles/r01_*).  Here the KEYS are split across workgroups:
//   attn_fwd_partial   one workgroup per (key chunk, head, image): thread = (query, key half); K/V tiles are staged
//                      in LDS as fp32 and read as wave-uniform (broadcast) rows; online softmax across the tiles of
//                      the chunk; emits (max, sum, sum p*v) per query
//   attn_fwd_combine   merges the chunk partials with log-sum-exp weights, writes o and lse
//   attn_bwd_dq        thread = (query, key half): recomputes p from lse, ds = p*(dO.v - delta), dq += ds*k
//   attn_bwd_dkv       thread = key: owns dk/dv rows in registers (no atomics), loops over the queries staged in LDS
// All accumulation is fp32; inputs/outputs are bf16 (autocast) or fp32.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "pd_attention.h"
#include "pd_common.h"
#include "mfma_bf16.h"
#include "pd_msda.h"

int g_attn_bwd_kc = 0;            // pd_debug_set "attn_bwd_kc" (tools/): keys per workgroup of attn_bwd_mfma forced to 128 / 256 / 512

namespace {

constexpr int D = 32;          // head dim
constexpr int QP = 128;        // queries handled per workgroup pass (threads 0..127 and 128..255 = two key halves)
constexpr int KT = 64;         // keys per LDS tile
constexpr int KC_FWD = 256;    // keys per workgroup (forward, dq)
constexpr int KC_DQ = 512;

typedef unsigned short bf16_t;

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ bf16_t f2bf(float f)
{
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

template <typename T> struct IO;
template <> struct IO<float> {
  static __device__ __forceinline__ void load32(const float *p, float *dst)
  {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 v = reinterpret_cast<const float4 *>(p)[i];
      dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
    }
  }
  static __device__ __forceinline__ void store32(float *p, const float *src)
  {
#pragma unroll
    for (int i = 0; i < 8; ++i) reinterpret_cast<float4 *>(p)[i] = make_float4(src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]);
  }
  static __device__ __forceinline__ void load8(const float *p, float *dst)
  {
    const float4 a = reinterpret_cast<const float4 *>(p)[0], b = reinterpret_cast<const float4 *>(p)[1];
    dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w; dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
  }
};
template <> struct IO<bf16_t> {
  static __device__ __forceinline__ void load8(const bf16_t *p, float *dst)
  {
    const uint4 v = *reinterpret_cast<const uint4 *>(p);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { dst[2 * i] = __uint_as_float(w[i] << 16); dst[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void load32(const bf16_t *p, float *dst)
  {
#pragma unroll
    for (int i = 0; i < 4; ++i) load8(p + 8 * i, dst + 8 * i);
  }
  static __device__ __forceinline__ void store32(bf16_t *p, const float *src)
  {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint4 v;
      unsigned *w = &v.x;
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = (unsigned)f2bf(src[8 * i + 2 * j]) | ((unsigned)f2bf(src[8 * i + 2 * j + 1]) << 16);
      reinterpret_cast<uint4 *>(p)[i] = v;
    }
  }
};

// stage `rows` consecutive key rows [k0, k0+rows) of head h, image b into LDS as fp32 [KT][D]; rows past Lk are zero
template <typename T>
__device__ __forceinline__ void stage_rows(const T *__restrict__ src, float (*dst)[D], int k0, int Lk, int b, int h, int B, int H)
{
  // KT*D/8 = 256 groups of 8 elements: one per thread
  const int r = threadIdx.x >> 2, c = (threadIdx.x & 3) * 8;
  float tmp[8];
  if (k0 + r < Lk) IO<T>::load8(src + ((int64_t)(k0 + r) * B + b) * (H * D) + h * D + c, tmp);
  else {
#pragma unroll
    for (int i = 0; i < 8; ++i) tmp[i] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) dst[r][c + i] = tmp[i];
}

// ---------------------------------------------------------------------------------------------- forward
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_partial(const T *__restrict__ q, const T *__restrict__ k, const T *__restrict__ v,
                                                         const uint8_t *__restrict__ mask, float *__restrict__ part_o,
                                                         float *__restrict__ part_ml, int B, int H, int Lq, int Lk, float scale,
                                                         int nchunk, int qpass)
{
  __shared__ __attribute__((aligned(16))) float Ks[KT][D];
  __shared__ __attribute__((aligned(16))) float Vs[KT][D];
  __shared__ float red[QP][D + 2];
  const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int qi = qpass * QP + (threadIdx.x & (QP - 1)), half = threadIdx.x >> 7;
  const bool active = qi < Lq;
  float qv[D];
  if (active) {
    IO<T>::load32(q + ((int64_t)qi * B + b) * (H * D) + h * D, qv);
#pragma unroll
    for (int d = 0; d < D; ++d) qv[d] *= scale;
  }
  float m_run = -INFINITY, l_run = 0.f, o[D];
#pragma unroll
  for (int d = 0; d < D; ++d) o[d] = 0.f;
  const int kbeg = chunk * KC_FWD, kend = min(Lk, kbeg + KC_FWD);
  for (int k0 = kbeg; k0 < kend; k0 += KT) {
    __syncthreads();
    stage_rows<T>(k, Ks, k0, Lk, b, h, B, H);
    stage_rows<T>(v, Vs, k0, Lk, b, h, B, H);
    __syncthreads();
    if (!active) continue;
    const int kk0 = half * (KT / 2);
    float s[KT / 2];
    float m_t = -INFINITY;
    const uint8_t *mrow = mask ? mask + ((int64_t)b * Lq + qi) * Lk + k0 + kk0 : nullptr;
#pragma unroll
    for (int kk = 0; kk < KT / 2; ++kk) {
      float acc = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) acc += qv[d] * Ks[kk0 + kk][d];
      const bool blocked = (k0 + kk0 + kk >= Lk) || (mrow && mrow[kk]);
      s[kk] = blocked ? -INFINITY : acc;
      m_t = fmaxf(m_t, s[kk]);
    }
    const float m_new = fmaxf(m_run, m_t);
    if (m_new == -INFINITY) continue;                       // everything blocked so far
    const float alpha = __expf(m_run - m_new);              // m_run = -inf -> 0
    l_run *= alpha;
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] *= alpha;
#pragma unroll
    for (int kk = 0; kk < KT / 2; ++kk) {
      const float p = __expf(s[kk] - m_new);                // blocked -> exp(-inf) = 0
      l_run += p;
#pragma unroll
      for (int d = 0; d < D; ++d) o[d] += p * Vs[kk0 + kk][d];
    }
    m_run = m_new;
  }
  // merge the two key halves of every query through LDS
  __syncthreads();
  const int ql = threadIdx.x & (QP - 1);
  if (half == 1) {
#pragma unroll
    for (int d = 0; d < D; ++d) red[ql][d] = o[d];
    red[ql][D] = m_run; red[ql][D + 1] = l_run;
  }
  __syncthreads();
  if (half == 0 && active) {
    const float m2 = red[ql][D], l2 = red[ql][D + 1];
    const float M = fmaxf(m_run, m2);
    const float a1 = (M == -INFINITY) ? 0.f : __expf(m_run - M), a2 = (M == -INFINITY) ? 0.f : __expf(m2 - M);
    const int64_t base = (((int64_t)b * H + h) * nchunk + chunk) * Lq + qi;
    float *po = part_o + base * D;
#pragma unroll
    for (int d = 0; d < D; d += 4)
      *reinterpret_cast<float4 *>(po + d) = make_float4(a1 * o[d] + a2 * red[ql][d], a1 * o[d + 1] + a2 * red[ql][d + 1],
                                                        a1 * o[d + 2] + a2 * red[ql][d + 2], a1 * o[d + 3] + a2 * red[ql][d + 3]);
    part_ml[base * 2] = M;
    part_ml[base * 2 + 1] = a1 * l_run + a2 * l2;
  }
}

// one 32-lane group per (b, h, q): lane = channel
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_combine(const float *__restrict__ part_o, const float *__restrict__ part_ml,
                                                         T *__restrict__ o, float *__restrict__ lse, int B, int H, int Lq, int nchunk)
{
  const int g = blockIdx.x * 8 + (threadIdx.x >> 5), d = threadIdx.x & 31;
  if (g >= B * H * Lq) return;
  const int qi = g % Lq, bh = g / Lq, h = bh % H, b = bh / H;
  // (up to 64 chunks at 16 384 keys: the loads of eight chunks are issued together — one load per iteration was a chain of ~60 latencies)
  float M = -INFINITY;
  const float2 *ml = reinterpret_cast<const float2 *>(part_ml) + (int64_t)bh * nchunk * Lq + qi;
  const float *po = part_o + ((int64_t)bh * nchunk * Lq + qi) * D + d;
  int c = 0;
  for (; c + 8 <= nchunk; c += 8) {
    float m8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) m8[u] = ml[(int64_t)(c + u) * Lq].x;
#pragma unroll
    for (int u = 0; u < 8; ++u) M = fmaxf(M, m8[u]);
  }
  for (; c < nchunk; ++c) M = fmaxf(M, ml[(int64_t)c * Lq].x);
  float L = 0.f, acc = 0.f;
  if (M != -INFINITY) {
    c = 0;
    for (; c + 8 <= nchunk; c += 8) {
      float2 s8[8];
      float o8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { s8[u] = ml[(int64_t)(c + u) * Lq]; o8[u] = po[(int64_t)(c + u) * Lq * D]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {                        // same chunk order as the one-by-one loop: identical sums
        const float w = __expf(s8[u].x - M);
        L += w * s8[u].y;
        acc += w * o8[u];
      }
    }
    for (; c < nchunk; ++c) {
      const float2 sv = ml[(int64_t)c * Lq];
      const float w = __expf(sv.x - M);
      L += w * sv.y;
      acc += w * po[(int64_t)c * Lq * D];
    }
  }
  const float outv = L > 0.f ? acc / L : 0.f;
  T *dst = o + ((int64_t)qi * B + b) * (H * D) + h * D + d;
  if (sizeof(T) == 2) *reinterpret_cast<bf16_t *>(dst) = f2bf(outv); else *reinterpret_cast<float *>(dst) = outv;
  if (d == 0) lse[(int64_t)bh * Lq + qi] = L > 0.f ? M + __logf(L) : -INFINITY;
}

// ---------------------------------------------------------------------------------------------- backward: dq
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq(const T *__restrict__ q, const T *__restrict__ k, const T *__restrict__ v,
                                                    const uint8_t *__restrict__ mask, const T *__restrict__ o, const T *__restrict__ d_o,
                                                    const float *__restrict__ lse, float *__restrict__ part_dq, int B, int H, int Lq,
                                                    int Lk, float scale, int nchunk, int qpass)
{
  __shared__ __attribute__((aligned(16))) float Ks[KT][D];
  __shared__ __attribute__((aligned(16))) float Vs[KT][D];
  __shared__ float red[QP][D];
  const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int qi = qpass * QP + (threadIdx.x & (QP - 1)), half = threadIdx.x >> 7;
  const bool active = qi < Lq;
  float qv[D], dov[D], dq[D];
  float lse_q = 0.f, delta = 0.f;
#pragma unroll
  for (int d = 0; d < D; ++d) dq[d] = 0.f;
  if (active) {
    const int64_t off = ((int64_t)qi * B + b) * (H * D) + h * D;
    float ov[D];
    IO<T>::load32(q + off, qv);
    IO<T>::load32(d_o + off, dov);
    IO<T>::load32(o + off, ov);
#pragma unroll
    for (int d = 0; d < D; ++d) { delta += dov[d] * ov[d]; qv[d] *= scale; }
    lse_q = lse[((int64_t)b * H + h) * Lq + qi];
  }
  const int kbeg = chunk * KC_DQ, kend = min(Lk, kbeg + KC_DQ);
  for (int k0 = kbeg; k0 < kend; k0 += KT) {
    __syncthreads();
    stage_rows<T>(k, Ks, k0, Lk, b, h, B, H);
    stage_rows<T>(v, Vs, k0, Lk, b, h, B, H);
    __syncthreads();
    if (!active || lse_q == -INFINITY) continue;
    const int kk0 = half * (KT / 2);
    const uint8_t *mrow = mask ? mask + ((int64_t)b * Lq + qi) * Lk + k0 + kk0 : nullptr;
#pragma unroll 4
    for (int kk = 0; kk < KT / 2; ++kk) {
      const bool blocked = (k0 + kk0 + kk >= Lk) || (mrow && mrow[kk]);
      if (blocked) continue;
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { s += qv[d] * Ks[kk0 + kk][d]; dp += dov[d] * Vs[kk0 + kk][d]; }
      const float ds = __expf(s - lse_q) * (dp - delta);
#pragma unroll
      for (int d = 0; d < D; ++d) dq[d] += ds * Ks[kk0 + kk][d];
    }
  }
  __syncthreads();
  const int ql = threadIdx.x & (QP - 1);
  if (half == 1) {
#pragma unroll
    for (int d = 0; d < D; ++d) red[ql][d] = dq[d];
  }
  __syncthreads();
  if (half == 0 && active) {
    float *dst = part_dq + ((((int64_t)b * H + h) * nchunk + chunk) * Lq + qi) * D;
#pragma unroll
    for (int d = 0; d < D; d += 4)
      *reinterpret_cast<float4 *>(dst + d) = make_float4((dq[d] + red[ql][d]) * scale, (dq[d + 1] + red[ql][d + 1]) * scale,
                                                         (dq[d + 2] + red[ql][d + 2]) * scale, (dq[d + 3] + red[ql][d + 3]) * scale);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_reduce(const float *__restrict__ part_dq, T *__restrict__ dq, int B, int H, int Lq, int nchunk)
{
  const int g = blockIdx.x * 8 + (threadIdx.x >> 5), d = threadIdx.x & 31;
  if (g >= B * H * Lq) return;
  const int qi = g % Lq, bh = g / Lq, h = bh % H, b = bh / H;
  float acc = 0.f;
  for (int c = 0; c < nchunk; ++c) acc += part_dq[(((int64_t)bh * nchunk + c) * Lq + qi) * D + d];
  T *dst = dq + ((int64_t)qi * B + b) * (H * D) + h * D + d;
  if (sizeof(T) == 2) *reinterpret_cast<bf16_t *>(dst) = f2bf(acc); else *reinterpret_cast<float *>(dst) = acc;
}

// ---------------------------------------------------------------------------------------------- backward: dk, dv
// thread = key; the queries (q*scale, dO, lse, delta) of this (b, h) are staged through LDS QS at a time
constexpr int QS = 32;

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv(const T *__restrict__ q, const T *__restrict__ k, const T *__restrict__ v,
                                                     const uint8_t *__restrict__ mask, const T *__restrict__ o, const T *__restrict__ d_o,
                                                     const float *__restrict__ lse, T *__restrict__ dk, T *__restrict__ dv, int B,
                                                     int H, int Lq, int Lk, float scale)
{
  __shared__ __attribute__((aligned(16))) float Qs[QS][D];
  __shared__ __attribute__((aligned(16))) float Os[QS][D];
  __shared__ float Ls[QS], Ds[QS];
  const int h = blockIdx.y, b = blockIdx.z;
  const int ki = blockIdx.x * 256 + threadIdx.x;
  const bool active = ki < Lk;
  float kv[D], vv[D], dkv[D], dvv[D];
#pragma unroll
  for (int d = 0; d < D; ++d) { dkv[d] = 0.f; dvv[d] = 0.f; }
  if (active) {
    const int64_t off = ((int64_t)ki * B + b) * (H * D) + h * D;
    IO<T>::load32(k + off, kv);
    IO<T>::load32(v + off, vv);
  }
  for (int q0 = 0; q0 < Lq; q0 += QS) {
    __syncthreads();
    {   // stage QS queries: 256 threads = QS rows x 8 groups of 4 channels (q and dO), delta by the row's 8 threads
      const int r = threadIdx.x >> 3, c = (threadIdx.x & 7) * 4;
      float part = 0.f;
      if (q0 + r < Lq) {
        const int64_t off = ((int64_t)(q0 + r) * B + b) * (H * D) + h * D + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float qx, dox, ox;
          if (sizeof(T) == 2) {
            qx = bf2f(reinterpret_cast<const bf16_t *>(q)[off + i]); dox = bf2f(reinterpret_cast<const bf16_t *>(d_o)[off + i]);
            ox = bf2f(reinterpret_cast<const bf16_t *>(o)[off + i]);
          } else {
            qx = reinterpret_cast<const float *>(q)[off + i]; dox = reinterpret_cast<const float *>(d_o)[off + i];
            ox = reinterpret_cast<const float *>(o)[off + i];
          }
          Qs[r][c + i] = qx * scale; Os[r][c + i] = dox;
          part += dox * ox;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { Qs[r][c + i] = 0.f; Os[r][c + i] = 0.f; }
      }
      part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64); part += __shfl_xor(part, 4, 64);
      if ((threadIdx.x & 7) == 0) {
        Ds[r] = part;
        Ls[r] = (q0 + r < Lq) ? lse[((int64_t)b * H + h) * Lq + q0 + r] : -INFINITY;
      }
    }
    __syncthreads();
    if (!active) continue;
    const int nq = min(QS, Lq - q0);
    for (int r = 0; r < nq; ++r) {
      const float lq = Ls[r];
      if (lq == -INFINITY) continue;
      if (mask && mask[((int64_t)b * Lq + q0 + r) * Lk + ki]) continue;
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) { s += Qs[r][d] * kv[d]; dp += Os[r][d] * vv[d]; }
      const float p = __expf(s - lq), ds = p * (dp - Ds[r]);
#pragma unroll
      for (int d = 0; d < D; ++d) { dkv[d] += ds * Qs[r][d]; dvv[d] += p * Os[r][d]; }
    }
  }
  if (active) {
    const int64_t off = ((int64_t)ki * B + b) * (H * D) + h * D;
    IO<T>::store32(dk + off, dkv);          // Qs already carries the softmax scale: d(q.k*scale)/dk = scale*q
    IO<T>::store32(dv + off, dvv);
  }
}

// ================================================================================================ MFMA path (bf16)
// Lq <= 128 queries, head dim 32, bf16 operands: both GEMMs of the attention run on v_mfma_f32_32x32x8_bf16_1k.
// With X and Y row-major and each lane holding 4 consecutive contraction elements of row (lane % 32),
//     mma(c, x, y):  c[i][j] += sum_k X[i][k] * Y[j][k]        (C = X . Y^T)
// and the result sits with lane = j (Y row), register e = X row (e&3) + 8*(e>>2) + 4*(lane>>5).  Score tiles are
// therefore produced directly in the layout the second GEMM wants as an operand (lane = query, 4 consecutive keys
// per register group, or lane = key, 4 consecutive queries), so nothing is transposed through LDS except the
// "4 keys at fixed channel" operands (V^T forward, K^T backward), gathered from row-major LDS tiles with 16-bit reads.
using pdmfma::bf16x4;
using pdmfma::f32x16;
using pdmfma::gather4;
using pdmfma::lds4;
using pdmfma::mma;
using pdmfma::pack4;

constexpr int LR = 36;     // bf16 per padded LDS row of a [rows][32] tile: 72-byte rows, 8-byte reads conflict-free
constexpr int LT = 132;    // bf16 per padded row of a transposed [32][128] tile
constexpr int MQ = 128;    // queries the MFMA path covers

// 4 mask bytes of keys key0..key0+3 of one mask row (as a little-endian u32); bytes at or past Lk read as 0
__device__ __forceinline__ unsigned mask4(const uint8_t *row, int key0, int Lk, bool aligned)
{
  if (key0 >= Lk) return 0u;
  if (aligned) return *reinterpret_cast<const unsigned *>(row + key0);
  unsigned m = 0;
  for (int i = 0; i < 4; ++i)
    if (key0 + i < Lk) m |= (unsigned)row[key0 + i] << (8 * i);
  return m;
}

template <bool MASK>
__global__ __launch_bounds__(256) void attn_fwd_mfma(const bf16_t *__restrict__ q, const bf16_t *__restrict__ k,
                                                     const bf16_t *__restrict__ v, const uint8_t *__restrict__ mask,
                                                     bf16_t *__restrict__ o, float *__restrict__ lse, float *__restrict__ part_o,
                                                     float *__restrict__ part_ml, int B, int H, int Lq, int Lk, float scale,
                                                     int nchunk, int ldkv, int kc)
{
  // kc: keys per workgroup (mfma_fwd_chunk).  ldkv: elements between consecutive (key, image) rows of k and v — H * 32 for dense tensors, more when k / v are column slices of a wider
  // matrix (the decoder's key / value projections of the layers that share a memory level come out of ONE product: functions/decoder_core.py)
  __shared__ __attribute__((aligned(16))) bf16_t Ks[2][32][LR];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[2][32][LR];
  const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int qi = wave * 32 + r;                     // this lane's query (column of every score tile of the wave)
  const bool qvalid = qi < Lq;
  const int64_t rs = (int64_t)H * D;                // elements per (position, image) row
  bf16x4 qreg[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qreg[s] = bf16x4{0, 0, 0, 0};
    if (qvalid) qreg[s] = *reinterpret_cast<const bf16x4 *>(q + ((int64_t)qi * B + b) * rs + h * D + 8 * s + 4 * hh);
  }
  const int kbeg = chunk * kc, kend = min(Lk, kbeg + kc);
  const int ntile = (kend - kbeg + 31) / 32;
  // staging: threads 0..127 move the K tile, 128..255 the V tile; 4 x 16 bytes per key row
  const bf16_t *src = (tid < 128) ? k : v;
  const int srow = (tid & 127) >> 2, spiece = tid & 3;
  auto gload = [&](int t) {
    const int key = kbeg + 32 * t + srow;
    uint4 val = make_uint4(0, 0, 0, 0);
    if (key < kend) val = *reinterpret_cast<const uint4 *>(src + ((int64_t)key * B + b) * ldkv + h * D + spiece * 8);
    return val;
  };
  auto lstore = [&](int buf, uint4 val) {
    bf16_t *dst = ((tid < 128) ? &Ks[buf][srow][0] : &Vs[buf][srow][0]) + spiece * 8;
    *reinterpret_cast<uint2 *>(dst) = make_uint2(val.x, val.y);
    *reinterpret_cast<uint2 *>(dst + 4) = make_uint2(val.z, val.w);
  };
  const bool aligned = (Lk & 3) == 0;
  const uint8_t *mrow = (MASK && qvalid) ? mask + ((int64_t)b * Lq + qi) * Lk : nullptr;
  float m_run = -INFINITY, l_part = 0.f;
  f32x16 oacc;
#pragma unroll
  for (int e = 0; e < 16; ++e) oacc[e] = 0.f;
  if (ntile > 0) lstore(0, gload(0));
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const int buf = t & 1, kbase = kbeg + 32 * t;
    uint4 nxt = make_uint4(0, 0, 0, 0);
    if (t + 1 < ntile) nxt = gload(t + 1);
    f32x16 sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) mma(sacc, lds4(&Ks[buf][r][8 * s + 4 * hh]), qreg[s]);        // S^T = K . Q^T
    float m_t = -INFINITY;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int key0 = kbase + 8 * g + 4 * hh;
      const unsigned m4 = mrow ? mask4(mrow, key0, Lk, aligned) : 0u;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool blocked = (key0 + i >= kend) || ((m4 >> (8 * i)) & 0xffu);
        const float sv = blocked ? -INFINITY : sacc[4 * g + i] * scale;
        sacc[4 * g + i] = sv;
        m_t = fmaxf(m_t, sv);
      }
    }
    m_t = fmaxf(m_t, __shfl_xor(m_t, 32, 64));                 // the two lane halves hold different keys of the same query
    const float m_new = fmaxf(m_run, m_t);
    const bool dead = m_new == -INFINITY;                       // everything blocked so far
    const float alpha = dead ? 1.f : __expf(m_run - m_new);
    float psum = 0.f;
    bf16x4 preg[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float p[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        p[i] = dead ? 0.f : __expf(sacc[4 * g + i] - m_new);
        psum += p[i];
      }
      preg[g] = pack4(p[0], p[1], p[2], p[3]);
    }
    l_part = l_part * alpha + psum;
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[e] *= alpha;
#pragma unroll
    for (int j = 0; j < 4; ++j)                                  // O^T[d][query] += V^T[d][keys] . P[query][keys]
      mma(oacc, gather4(&Vs[buf][8 * j + 4 * hh][r], LR), preg[j]);
    m_run = m_new;
    if (t + 1 < ntile) lstore(buf ^ 1, nxt);
    __syncthreads();
  }
  const float l_run = l_part + __shfl_xor(l_part, 32, 64);
  if (!qvalid) return;
  const int64_t bh = (int64_t)b * H + h;
  if (nchunk == 1) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    bf16_t *dst = o + ((int64_t)qi * B + b) * rs + h * D;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<bf16x4 *>(dst + 8 * g + 4 * hh) = pack4(oacc[4 * g] * inv, oacc[4 * g + 1] * inv, oacc[4 * g + 2] * inv, oacc[4 * g + 3] * inv);
    if (hh == 0) lse[bh * Lq + qi] = l_run > 0.f ? m_run + __logf(l_run) : -INFINITY;
  } else {
    const int64_t base = (bh * nchunk + chunk) * Lq + qi;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4 *>(part_o + base * D + 8 * g + 4 * hh) = make_float4(oacc[4 * g], oacc[4 * g + 1], oacc[4 * g + 2], oacc[4 * g + 3]);
    if (hh == 0) { part_ml[base * 2] = m_run; part_ml[base * 2 + 1] = l_run; }
  }
}

// one workgroup = KC_DQ keys of one (image, head); wave w owns key tiles w, w+4, ...: dK / dV of a tile are complete
// in registers after the loop over the query sub-tiles, dQ is accumulated over the wave's tiles, reduced over the
// four waves through LDS and leaves as one partial per workgroup (attn_bwd_dq_reduce sums the partials).
template <bool MASK>
__global__ __launch_bounds__(256, 2) void attn_bwd_mfma(const bf16_t *__restrict__ q, const bf16_t *__restrict__ k,
                                                     const bf16_t *__restrict__ v, const uint8_t *__restrict__ mask,
                                                     const bf16_t *__restrict__ o, const bf16_t *__restrict__ d_o,
                                                     const float *__restrict__ lse, float *__restrict__ part_dq,
                                                     bf16_t *__restrict__ dq, bf16_t *__restrict__ dk, bf16_t *__restrict__ dv,
                                                     int B, int H, int Lq, int Lk, float scale, int nchunk, int kc, int ldkv)
{
  __shared__ __attribute__((aligned(16))) bf16_t Qs[MQ][LR];
  __shared__ __attribute__((aligned(16))) bf16_t Os[MQ][LR];          // dO rows
  __shared__ __attribute__((aligned(16))) bf16_t Qt[D][LT];
  __shared__ __attribute__((aligned(16))) bf16_t Ot[D][LT];           // dO^T
  __shared__ __attribute__((aligned(16))) float Ls[MQ];
  __shared__ __attribute__((aligned(16))) float Ds[MQ];
  __shared__ __attribute__((aligned(16))) bf16_t Kw[4][32][LR];       // per-wave K tile for the K^T gathers
  __shared__ float dqs[MQ][D + 1];
  const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int64_t rs = (int64_t)H * D;
  const int64_t bh = (int64_t)b * H + h;
  // ---- stage all queries of this (image, head): Q, dO row-major and transposed, delta = rowsum(dO * O), lse
  for (int idx = tid; idx < MQ * 4; idx += 256) {
    const int row = idx >> 2, piece = idx & 3;
    // (straight-line loads from a clamped row, zeroed past Lq: behind `if (row < Lq)` each of the three was a memory round trip of its own)
    const bool live = row < Lq;
    const int64_t off = ((int64_t)(live ? row : 0) * B + b) * rs + h * D + piece * 8;
    uint4 qv = *reinterpret_cast<const uint4 *>(q + off), gv = *reinterpret_cast<const uint4 *>(d_o + off), ov = *reinterpret_cast<const uint4 *>(o + off);
    if (!live) { qv = make_uint4(0, 0, 0, 0); gv = qv; ov = qv; }
    *reinterpret_cast<uint2 *>(&Qs[row][piece * 8]) = make_uint2(qv.x, qv.y);
    *reinterpret_cast<uint2 *>(&Qs[row][piece * 8 + 4]) = make_uint2(qv.z, qv.w);
    *reinterpret_cast<uint2 *>(&Os[row][piece * 8]) = make_uint2(gv.x, gv.y);
    *reinterpret_cast<uint2 *>(&Os[row][piece * 8 + 4]) = make_uint2(gv.z, gv.w);
    const unsigned qa[4] = {qv.x, qv.y, qv.z, qv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w}, oa[4] = {ov.x, ov.y, ov.z, ov.w};
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      Qt[piece * 8 + 2 * i][row] = (bf16_t)(qa[i] & 0xffffu);
      Qt[piece * 8 + 2 * i + 1][row] = (bf16_t)(qa[i] >> 16);
      Ot[piece * 8 + 2 * i][row] = (bf16_t)(ga[i] & 0xffffu);
      Ot[piece * 8 + 2 * i + 1][row] = (bf16_t)(ga[i] >> 16);
      part += bf2f((bf16_t)(ga[i] & 0xffffu)) * bf2f((bf16_t)(oa[i] & 0xffffu)) + bf2f((bf16_t)(ga[i] >> 16)) * bf2f((bf16_t)(oa[i] >> 16));
    }
    part += __shfl_xor(part, 1, 64);
    part += __shfl_xor(part, 2, 64);
    if (piece == 0) {
      Ds[row] = part;
      float l = INFINITY;                       // rows past Lq, and rows whose keys were all blocked: p = exp(s - inf) = 0
      if (row < Lq) {
        l = lse[bh * Lq + row];
        if (l == -INFINITY) l = INFINITY;
      }
      Ls[row] = l;
    }
  }
  __syncthreads();
  const int kbeg = chunk * kc, kend = min(Lk, kbeg + kc);          // kc keys per workgroup (128 / 256 / 512: see mfma_bwd_chunk)
  const int ntile = (kend - kbeg + 31) / 32;
  const int nsub = (Lq + 31) / 32;
  const bool aligned = (Lk & 3) == 0;
  f32x16 dqacc[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int e = 0; e < 16; ++e) dqacc[s][e] = 0.f;
  for (int t0 = 0; t0 < ntile; t0 += 4) {                    // uniform trip count: every wave reaches the barriers
    const int t = t0 + wave;
    const bool tvalid = t < ntile;
    const int kbase = kbeg + 32 * t;
    const int key = kbase + r;
    const bool kvalid = tvalid && key < kend;
    bf16x4 kreg[4], vreg[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kreg[s] = bf16x4{0, 0, 0, 0};
      vreg[s] = bf16x4{0, 0, 0, 0};
      if (kvalid) {
        const int64_t off = ((int64_t)key * B + b) * ldkv + h * D + 8 * s + 4 * hh;        // (k / v rows: ldkv apart; dk / dv are dense)
        kreg[s] = *reinterpret_cast<const bf16x4 *>(k + off);
        vreg[s] = *reinterpret_cast<const bf16x4 *>(v + off);
      }
    }
    // the mask bytes of this round's 4 query sub-tiles x 4 key groups, all in flight together and under the barriers below: read where
    // they are used (one dependent 4-byte load per key group) they were up to 16 global-load latencies in a row per round
    unsigned m4s[4][4];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      const int ql = 32 * sub + r;
      const uint8_t *mrow = (MASK && tvalid && sub < nsub && ql < Lq) ? mask + ((int64_t)b * Lq + ql) * Lk : nullptr;
#pragma unroll
      for (int g = 0; g < 4; ++g) m4s[sub][g] = mrow ? mask4(mrow, kbase + 8 * g + 4 * hh, Lk, aligned) : 0u;
    }
    __syncthreads();                                           // previous round's K^T gathers are done
#pragma unroll
    for (int s = 0; s < 4; ++s) *reinterpret_cast<bf16x4 *>(&Kw[wave][r][8 * s + 4 * hh]) = kreg[s];
    __syncthreads();
    bf16x4 ktreg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ktreg[j] = gather4(&Kw[wave][8 * j + 4 * hh][r], LR);
    f32x16 dkacc, dvacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) { dkacc[e] = 0.f; dvacc[e] = 0.f; }
    if (tvalid) {
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        if (sub < nsub) {
          const int ql = 32 * sub + r;
          bf16x4 qreg[4], greg[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            qreg[s] = lds4(&Qs[ql][8 * s + 4 * hh]);
            greg[s] = lds4(&Os[ql][8 * s + 4 * hh]);
          }
          // ---- phase A: lane = query, registers = keys
          f32x16 sT, dpT;
#pragma unroll
          for (int e = 0; e < 16; ++e) { sT[e] = 0.f; dpT[e] = 0.f; }
#pragma unroll
          for (int s = 0; s < 4; ++s) { mma(sT, kreg[s], qreg[s]); mma(dpT, vreg[s], greg[s]); }
          const float lse_q = Ls[ql], delta_q = Ds[ql];
          unsigned mybits = 0;                                 // bit i: query (32*sub + i) may not attend key (kbase + lane%32)
          bf16x4 dsT[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int key0 = kbase + 8 * g + 4 * hh;
            const unsigned m4 = m4s[sub][g];
            float ds[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const bool blocked = (key0 + i >= kend) || ((m4 >> (8 * i)) & 0xffu);
              const float p = blocked ? 0.f : __expf(sT[4 * g + i] * scale - lse_q);
              ds[i] = p * (dpT[4 * g + i] - delta_q);
              const unsigned long long bal = __ballot(blocked);
              if (r == 8 * g + i) mybits = (unsigned)bal;              // lanes of half 0 carry key 8g + i
              if (r == 8 * g + 4 + i) mybits = (unsigned)(bal >> 32);  // lanes of half 1 carry key 8g + 4 + i
            }
            dsT[g] = pack4(ds[0], ds[1], ds[2], ds[3]);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) mma(dqacc[sub], ktreg[j], dsT[j]);            // dQ^T[d][query] += K^T[d][keys] . dS^T
          // ---- phase B: lane = key, registers = queries
          f32x16 sN, dpN;
#pragma unroll
          for (int e = 0; e < 16; ++e) { sN[e] = 0.f; dpN[e] = 0.f; }
#pragma unroll
          for (int s = 0; s < 4; ++s) { mma(sN, qreg[s], kreg[s]); mma(dpN, greg[s], vreg[s]); }
          bf16x4 pN[4], dsN[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 l4 = *reinterpret_cast<const float4 *>(&Ls[32 * sub + 8 * g + 4 * hh]);
            const float4 d4 = *reinterpret_cast<const float4 *>(&Ds[32 * sub + 8 * g + 4 * hh]);
            const float la[4] = {l4.x, l4.y, l4.z, l4.w}, da[4] = {d4.x, d4.y, d4.z, d4.w};
            float p[4], ds[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const bool blocked = (mybits >> (8 * g + 4 * hh + i)) & 1u;
              p[i] = blocked ? 0.f : __expf(sN[4 * g + i] * scale - la[i]);
              ds[i] = p[i] * (dpN[4 * g + i] - da[i]);
            }
            pN[g] = pack4(p[0], p[1], p[2], p[3]);
            dsN[g] = pack4(ds[0], ds[1], ds[2], ds[3]);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            mma(dvacc, lds4(&Ot[r][32 * sub + 8 * j + 4 * hh]), pN[j]);                // dV^T[d][key] += dO^T[d][queries] . P
            mma(dkacc, lds4(&Qt[r][32 * sub + 8 * j + 4 * hh]), dsN[j]);               // dK^T[d][key] += Q^T[d][queries] . dS
          }
        }
      }
      if (kvalid) {
        bf16_t *dkp = dk + ((int64_t)key * B + b) * rs + h * D, *dvp = dv + ((int64_t)key * B + b) * rs + h * D;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          *reinterpret_cast<bf16x4 *>(dkp + 8 * g + 4 * hh) = pack4(dkacc[4 * g] * scale, dkacc[4 * g + 1] * scale, dkacc[4 * g + 2] * scale, dkacc[4 * g + 3] * scale);
          *reinterpret_cast<bf16x4 *>(dvp + 8 * g + 4 * hh) = pack4(dvacc[4 * g], dvacc[4 * g + 1], dvacc[4 * g + 2], dvacc[4 * g + 3]);
        }
      }
    }
  }
  // ---- dQ: sum the four waves through LDS, then one partial per workgroup (or dq itself when there is one chunk)
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int sub = 0; sub < 4; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int d = (e & 3) + 8 * (e >> 2) + 4 * hh;
          if (w == 0) dqs[32 * sub + r][d] = dqacc[sub][e];
          else dqs[32 * sub + r][d] += dqacc[sub][e];
        }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < Lq * D; idx += 256) {
    const int qq = idx >> 5, d = idx & 31;
    const float val = dqs[qq][d] * scale;
    if (nchunk == 1) dq[((int64_t)qq * B + b) * rs + h * D + d] = f2bf(val);
    else part_dq[((bh * nchunk + chunk) * Lq + qq) * D + d] = val;
  }
}

inline int nchunks(int Lk, int kc) { return Lk <= 0 ? 1 : (Lk + kc - 1) / kc; }
// keys per workgroup of attn_fwd_mfma: KC_FWD = 256 keys are 8 tiles in a row at ~2 us each; at 1 024 / 4 096 keys that is 64 / 256 workgroups on
// 256 CUs.  Fewer keys per workgroup (down to 64) until ~512 workgroups exist — only where the keys are split over workgroups anyway (a single
// chunk needs no combine launch, and stays single).
inline int mfma_fwd_chunk(int B, int H, int Lk)
{
  int kc = KC_FWD;
  if (Lk > KC_FWD)
    while (kc > 64 && (int64_t)B * H * nchunks(Lk, kc) < 512) kc >>= 1;
  return kc;
}
// keys per workgroup of attn_bwd_mfma: a workgroup walks its keys 128 at a time (4 waves x 32) and a round costs ~15 us whatever the
// machine's load, so KC_DQ = 512 keys are 4 rounds in a row — fine at 16 384 keys (512 workgroups), wasteful at 1 024 (32 workgroups on
// 256 CUs).  Fewer keys per workgroup until ~256 workgroups exist: 1 024 keys -> 128 per workgroup (1 round), 4 096 -> 256.
inline int mfma_bwd_chunk(int B, int H, int Lk)
{
  if (g_attn_bwd_kc == 128 || g_attn_bwd_kc == 256 || g_attn_bwd_kc == 512) return g_attn_bwd_kc;
  int kc = KC_DQ;
  while (kc > 128 && (int64_t)B * H * nchunks(Lk, kc) < 256) kc >>= 1;
  return kc;
}

int check(const void *a, const void *b, const void *c, int B, int H, int Lq, int Lk, int dtype, const char *who)
{
  if (B < 0 || H <= 0 || Lq < 0 || Lk < 0) return pd_set_error(PD_ERR_INVALID_ARG, "%s: bad sizes B=%d H=%d Lq=%d Lk=%d", who, B, H, Lq, Lk);
  if (dtype != PD_F32 && dtype != PD_BF16) return pd_set_error(PD_ERR_INVALID_ARG, "%s: dtype %d (float32 / bfloat16 only)", who, dtype);
  if (B * Lq * Lk != 0 && (!a || !b || !c)) return pd_set_error(PD_ERR_INVALID_ARG, "%s: null pointer", who);
  return PD_OK;
}

}  // namespace

int g_pd_dbg_attn_scalar = 0;     // experiment knob: 1 forces the scalar kernels for bf16 too

extern "C" int64_t pd_attn_workspace_floats(int B, int H, int Lq, int Lk)
{
  const int64_t n1 = (int64_t)B * H * nchunks(Lk, mfma_fwd_chunk(B, H, Lk)) * Lq * (D + 2);
  const int64_t n2 = (int64_t)B * H * nchunks(Lk, 128) * Lq * D;   // the matrix-core backward may cut the keys as fine as 128 per workgroup
  return (n1 > n2 ? n1 : n2) + 64;
}

extern "C" int pd_attn_fwd_d32(const void *q, const void *k, const void *v, const uint8_t *mask, void *o, float *lse,
                               float *workspace, int B, int H, int Lq, int Lk, float scale, int dtype, void *stream_)
{
  return pd_attn_fwd_d32_ld(q, k, v, mask, o, lse, workspace, B, H, Lq, Lk, scale, dtype, H * D, stream_);
}

extern "C" int pd_attn_fwd_d32_ld(const void *q, const void *k, const void *v, const uint8_t *mask, void *o, float *lse,
                                  float *workspace, int B, int H, int Lq, int Lk, float scale, int dtype, int ld_kv, void *stream_)
{
  int rc = check(q, k, v, B, H, Lq, Lk, dtype, "pd_attn_fwd_d32");
  if (rc) return rc;
  if (ld_kv < H * D || (ld_kv & 7)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_attn_fwd_d32_ld: ld_kv=%d (>= H * 32, a multiple of 8)", ld_kv);
  if (ld_kv != H * D && !(dtype == PD_BF16 && Lq <= MQ && Lk > 0 && !g_pd_dbg_attn_scalar))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_attn_fwd_d32_ld: strided k / v only on the matrix-core path (bf16, <= %d queries)", MQ);
  if (B * Lq == 0) return PD_OK;
  if (!o || !lse || !workspace) return pd_set_error(PD_ERR_INVALID_ARG, "pd_attn_fwd_d32: null output");
  hipStream_t s = (hipStream_t)stream_;
  if (dtype == PD_BF16 && Lq <= MQ && Lk > 0 && !g_pd_dbg_attn_scalar) {           // matrix-core path
    const int kc = mfma_fwd_chunk(B, H, Lk), nc = nchunks(Lk, kc);
    float *part_o = workspace, *part_ml = workspace + (int64_t)B * H * nc * Lq * D;
    if (mask) hipLaunchKernelGGL(attn_fwd_mfma<true>, dim3(nc, H, B), dim3(256), 0, s, (const bf16_t *)q, (const bf16_t *)k, (const bf16_t *)v,
                                 mask, (bf16_t *)o, lse, part_o, part_ml, B, H, Lq, Lk, scale, nc, ld_kv, kc);
    else hipLaunchKernelGGL(attn_fwd_mfma<false>, dim3(nc, H, B), dim3(256), 0, s, (const bf16_t *)q, (const bf16_t *)k, (const bf16_t *)v,
                            mask, (bf16_t *)o, lse, part_o, part_ml, B, H, Lq, Lk, scale, nc, ld_kv, kc);
    if (nc > 1) hipLaunchKernelGGL(attn_fwd_combine<bf16_t>, dim3((B * H * Lq + 7) / 8), dim3(256), 0, s, part_o, part_ml, (bf16_t *)o, lse, B, H, Lq, nc);
    return pd_check_launch("pd_attn_fwd_d32");
  }
  const int nc = nchunks(Lk, KC_FWD);
  float *part_o = workspace, *part_ml = workspace + (int64_t)B * H * nc * Lq * D;
  for (int qp = 0; qp * QP < Lq; ++qp) {
    if (dtype == PD_BF16)
      hipLaunchKernelGGL(attn_fwd_partial<bf16_t>, dim3(nc, H, B), dim3(256), 0, s, (const bf16_t *)q, (const bf16_t *)k,
                         (const bf16_t *)v, mask, part_o, part_ml, B, H, Lq, Lk, scale, nc, qp);
    else
      hipLaunchKernelGGL(attn_fwd_partial<float>, dim3(nc, H, B), dim3(256), 0, s, (const float *)q, (const float *)k,
                         (const float *)v, mask, part_o, part_ml, B, H, Lq, Lk, scale, nc, qp);
  }
  const int groups = B * H * Lq;
  if (dtype == PD_BF16)
    hipLaunchKernelGGL(attn_fwd_combine<bf16_t>, dim3((groups + 7) / 8), dim3(256), 0, s, part_o, part_ml, (bf16_t *)o, lse, B, H, Lq, nc);
  else
    hipLaunchKernelGGL(attn_fwd_combine<float>, dim3((groups + 7) / 8), dim3(256), 0, s, part_o, part_ml, (float *)o, lse, B, H, Lq, nc);
  return pd_check_launch("pd_attn_fwd_d32");
}

extern "C" int pd_attn_bwd_d32(const void *q, const void *k, const void *v, const uint8_t *mask, const void *o, const void *d_o,
                               const float *lse, void *dq, void *dk, void *dv, float *workspace, int B, int H, int Lq,
                               int Lk, float scale, int dtype, void *stream_)
{
  return pd_attn_bwd_d32_ld(q, k, v, mask, o, d_o, lse, dq, dk, dv, workspace, B, H, Lq, Lk, scale, dtype, H * D, stream_);
}

extern "C" int pd_attn_bwd_d32_ld(const void *q, const void *k, const void *v, const uint8_t *mask, const void *o, const void *d_o,
                                  const float *lse, void *dq, void *dk, void *dv, float *workspace, int B, int H, int Lq,
                                  int Lk, float scale, int dtype, int ld_kv, void *stream_)
{
  int rc = check(q, k, v, B, H, Lq, Lk, dtype, "pd_attn_bwd_d32");
  if (rc) return rc;
  if (ld_kv < H * D || (ld_kv & 3)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_attn_bwd_d32_ld: ld_kv=%d (>= H * 32, a multiple of 4)", ld_kv);
  if (ld_kv != H * D && !(dtype == PD_BF16 && Lq <= MQ && !g_pd_dbg_attn_scalar))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_attn_bwd_d32_ld: strided k / v only on the matrix-core path (bf16, <= %d queries)", MQ);
  if (B * Lq * Lk == 0) return PD_OK;
  if (!o || !d_o || !lse || !dq || !dk || !dv || !workspace) return pd_set_error(PD_ERR_INVALID_ARG, "pd_attn_bwd_d32: null pointer");
  hipStream_t s = (hipStream_t)stream_;
  const int groups = B * H * Lq;
  if (dtype == PD_BF16 && Lq <= MQ && !g_pd_dbg_attn_scalar) {                     // matrix-core path
    const int kc = mfma_bwd_chunk(B, H, Lk), nc = nchunks(Lk, kc);
    if (mask) hipLaunchKernelGGL(attn_bwd_mfma<true>, dim3(nc, H, B), dim3(256), 0, s, (const bf16_t *)q, (const bf16_t *)k, (const bf16_t *)v,
                                 mask, (const bf16_t *)o, (const bf16_t *)d_o, lse, workspace, (bf16_t *)dq, (bf16_t *)dk, (bf16_t *)dv,
                                 B, H, Lq, Lk, scale, nc, kc, ld_kv);
    else hipLaunchKernelGGL(attn_bwd_mfma<false>, dim3(nc, H, B), dim3(256), 0, s, (const bf16_t *)q, (const bf16_t *)k, (const bf16_t *)v,
                            mask, (const bf16_t *)o, (const bf16_t *)d_o, lse, workspace, (bf16_t *)dq, (bf16_t *)dk, (bf16_t *)dv,
                            B, H, Lq, Lk, scale, nc, kc, ld_kv);
    if (nc > 1) hipLaunchKernelGGL(attn_bwd_dq_reduce<bf16_t>, dim3((groups + 7) / 8), dim3(256), 0, s, workspace, (bf16_t *)dq, B, H, Lq, nc);
    return pd_check_launch("pd_attn_bwd_d32");
  }
  const int nc = nchunks(Lk, KC_DQ);
  if (dtype == PD_BF16) {
    for (int qp = 0; qp * QP < Lq; ++qp)
      hipLaunchKernelGGL(attn_bwd_dq<bf16_t>, dim3(nc, H, B), dim3(256), 0, s, (const bf16_t *)q, (const bf16_t *)k, (const bf16_t *)v,
                         mask, (const bf16_t *)o, (const bf16_t *)d_o, lse, workspace, B, H, Lq, Lk, scale, nc, qp);
    hipLaunchKernelGGL(attn_bwd_dq_reduce<bf16_t>, dim3((groups + 7) / 8), dim3(256), 0, s, workspace, (bf16_t *)dq, B, H, Lq, nc);
    hipLaunchKernelGGL(attn_bwd_dkv<bf16_t>, dim3((Lk + 255) / 256, H, B), dim3(256), 0, s, (const bf16_t *)q, (const bf16_t *)k,
                       (const bf16_t *)v, mask, (const bf16_t *)o, (const bf16_t *)d_o, lse, (bf16_t *)dk, (bf16_t *)dv, B, H, Lq, Lk, scale);
  } else {
    for (int qp = 0; qp * QP < Lq; ++qp)
      hipLaunchKernelGGL(attn_bwd_dq<float>, dim3(nc, H, B), dim3(256), 0, s, (const float *)q, (const float *)k, (const float *)v,
                         mask, (const float *)o, (const float *)d_o, lse, workspace, B, H, Lq, Lk, scale, nc, qp);
    hipLaunchKernelGGL(attn_bwd_dq_reduce<float>, dim3((groups + 7) / 8), dim3(256), 0, s, workspace, (float *)dq, B, H, Lq, nc);
    hipLaunchKernelGGL(attn_bwd_dkv<float>, dim3((Lk + 255) / 256, H, B), dim3(256), 0, s, (const float *)q, (const float *)k,
                       (const float *)v, mask, (const float *)o, (const float *)d_o, lse, (float *)dk, (float *)dv, B, H, Lq, Lk, scale);
  }
  return pd_check_launch("pd_attn_bwd_d32");
}
