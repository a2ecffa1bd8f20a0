// Swin (shifted-)window attention for gfx950: 12 x 12 windows (144 tokens), head_dim 32, bf16 on
// v_mfma_f32_16x16x16_bf16_1k, fp32 softmax.  C-ABI and the math it replaces: include/pd_window_attention.h.
//
// One WAVEFRONT owns one (window, head) at a time (workgroup = 1 wave, so LDS traffic needs no cross-wave barrier) and
// walks `chunk` windows of the same head so the bias table / its gradient are staged once.
//
// MFMA operand convention used below (16x16x16, lane = 16*g + c):
//     mma16(acc, x, y): acc[i][j] += sum_k X[i][k] * Y[j][k]
//     x: this lane holds X[row c][k = 4g .. 4g+3],   y: this lane holds Y[row c][k = 4g .. 4g+3]
//     acc: this lane holds acc[i = 4g + e][j = c], e = 0..3
// A lane therefore ends up with 4 CONSECUTIVE i for one j — which is again the "4 consecutive k" an operand needs, so
// softmax probabilities feed the next MFMA straight from registers:
//   forward  S^T = K.Q^T   -> lane: query c, keys 4g+e   -> P is the y-operand of O^T[d][q] = Vt[d][key] . P[q][key]
//   backward S   = Q.K^T   -> lane: key c, queries 4g+e  -> P / dS are the y-operands of dV^T = dOt.P, dK^T = Qt.dS;
//                                                           only dQ needs dS transposed (16 x 16 tile through LDS).
// "t"-suffixed operands (Vt, Qt, dOt) are [32][144] transposes staged in LDS; Kt is gathered from global per key tile.
//
// Relative-position bias: 4 consecutive tokens starting at a multiple of 4 lie in one row of the 12 x 12 window, so with
// A(t) = t + 11*(t/12) the 4 table indices a lane needs, A(q) - A(key) + 264, are consecutive: one address, 4 LDS reads.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mfma_bf16.h"
#include "pd_common.h"
#include "pd_msda.h"
#include "pd_window_attention.h"

namespace {
using namespace pdmfma;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int N = 144, D = 32, NT = 9, TBL = 529;
constexpr int TP = 148;                      // pitch (elements) of the [32][144] transposed LDS tiles
constexpr int TR = 20;                       // pitch of the 16 x 16 dS transposition tile
constexpr float LOG2E = 1.4426950408889634f;
constexpr float MASKED = -100.0f * LOG2E;    // the reference's additive -100 in base-2 units

__device__ __forceinline__ void mma16(f32x4 &c, bf16x4 x, bf16x4 y) { c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, c, 0, 0, 0); }
__device__ __forceinline__ int rel_a(int t) { return t + 11 * ((t * 171) >> 11); }            // t + 11*(t/12), t < 144
__device__ __forceinline__ bf16x4 ld4(const bf16_t *p) { return *reinterpret_cast<const bf16x4 *>(p); }
__device__ __forceinline__ void st4(bf16_t *p, float a, float b, float c, float d) { *reinterpret_cast<bf16x4 *>(p) = pack4(a, b, c, d); }

// rows [144] x 32 columns at `src` (row pitch ld) -> dst[d][row] (pitch TP)
__device__ __forceinline__ void stage_transposed(const bf16_t *__restrict__ src, int64_t ld, bf16_t *dst, int lane)
{
#pragma unroll
  for (int it = 0; it < 9; ++it) {
    const int idx = it * 64 + lane, row = idx >> 2, ch = idx & 3;
    const uint4 v = *reinterpret_cast<const uint4 *>(src + row * ld + ch * 8);
    bf16_t *d = dst + (ch * 8) * TP + row;
    d[0 * TP] = (bf16_t)(v.x & 0xffff); d[1 * TP] = (bf16_t)(v.x >> 16);
    d[2 * TP] = (bf16_t)(v.y & 0xffff); d[3 * TP] = (bf16_t)(v.y >> 16);
    d[4 * TP] = (bf16_t)(v.z & 0xffff); d[5 * TP] = (bf16_t)(v.z >> 16);
    d[6 * TP] = (bf16_t)(v.w & 0xffff); d[7 * TP] = (bf16_t)(v.w >> 16);
  }
}

template <bool MASK>
__global__ __launch_bounds__(64) void wattn_fwd(const bf16_t *__restrict__ qkv, const float *__restrict__ table,
                                                const uint8_t *__restrict__ region, const uint8_t *__restrict__ flags,
                                                bf16_t *__restrict__ out, float *__restrict__ lse, int B_, int nW, int heads,
                                                float c1, int chunk)
{
  __shared__ float tbl[TBL + 3];
  __shared__ __attribute__((aligned(16))) bf16_t vt[D * TP];
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4, h = blockIdx.y;
  const int C = heads * D;
  const int64_t ld = 3 * C;
  for (int i = lane; i < TBL; i += 64) tbl[i] = table[i * heads + h] * LOG2E;

  for (int u = 0; u < chunk; ++u) {
    const int b = blockIdx.x * chunk + u;
    if (b >= B_) break;
    const bf16_t *base = qkv + (int64_t)b * N * ld + h * D;
    __syncthreads();
    stage_transposed(base + 2 * C, ld, vt, lane);
    bf16x4 kf[NT][2];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      kf[kt][0] = ld4(base + C + (16 * kt + c) * ld + 4 * g);
      kf[kt][1] = ld4(base + C + (16 * kt + c) * ld + 16 + 4 * g);
    }
    const int w = b % nW;
    const bool masked = MASK && flags[w] != 0;
    const uint8_t *r = MASK ? region + w * N : nullptr;
    __syncthreads();
#pragma unroll 1
    for (int qt = 0; qt < NT; ++qt) {
      const int q = 16 * qt + c;
      const bf16x4 q0 = ld4(base + q * ld + 4 * g), q1 = ld4(base + q * ld + 16 + 4 * g);
      f32x4 s[NT];
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        mma16(s[kt], kf[kt][0], q0);
        mma16(s[kt], kf[kt][1], q1);
      }
      const int aq = rel_a(q) + 264 - 3;
      const unsigned rq = masked ? r[q] : 0u;
      float m = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        const int key0 = 16 * kt + 4 * g;
        const float *tb = tbl + (aq - rel_a(key0));                     // index for key0+e is tb[3 - e]
        s[kt][0] = fmaf(s[kt][0], c1, tb[3]); s[kt][1] = fmaf(s[kt][1], c1, tb[2]);
        s[kt][2] = fmaf(s[kt][2], c1, tb[1]); s[kt][3] = fmaf(s[kt][3], c1, tb[0]);
        if (masked) {
          const unsigned rk = *reinterpret_cast<const unsigned *>(r + key0);
#pragma unroll
          for (int e = 0; e < 4; ++e) if (((rk >> (8 * e)) & 255u) != rq) s[kt][e] += MASKED;
        }
        m = fmaxf(fmaxf(m, fmaxf(s[kt][0], s[kt][1])), fmaxf(s[kt][2], s[kt][3]));
      }
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      float sum = 0.f;
      bf16x4 p[NT];
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[kt][e] = __builtin_amdgcn_exp2f(s[kt][e] - m); sum += s[kt][e]; }
        p[kt] = pack4(s[kt][0], s[kt][1], s[kt][2], s[kt][3]);
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        mma16(o0, lds4(vt + c * TP + 16 * kt + 4 * g), p[kt]);            // O^T[d = 4g+e][q = c]
        mma16(o1, lds4(vt + (16 + c) * TP + 16 * kt + 4 * g), p[kt]);
      }
      const float inv = 1.f / sum;
      bf16_t *o = out + ((int64_t)b * N + q) * C + h * D + 4 * g;
      st4(o, o0[0] * inv, o0[1] * inv, o0[2] * inv, o0[3] * inv);
      st4(o + 16, o1[0] * inv, o1[1] * inv, o1[2] * inv, o1[3] * inv);
      if (g == 0) lse[((int64_t)b * heads + h) * N + q] = m + log2f(sum);
    }
  }
}

template <bool MASK>
__global__ __launch_bounds__(64) void wattn_bwd(const bf16_t *__restrict__ qkv, const float *__restrict__ table,
                                                const uint8_t *__restrict__ region, const uint8_t *__restrict__ flags,
                                                const bf16_t *__restrict__ out, const bf16_t *__restrict__ dout,
                                                const float *__restrict__ lse, bf16_t *__restrict__ dqkv,
                                                float *__restrict__ dtable, int B_, int nW, int heads, float scale, float c1,
                                                int chunk)
{
  __shared__ float tbl[TBL + 3];
  __shared__ float dtb[TBL + 3];
  __shared__ __attribute__((aligned(16))) bf16_t qts[D * TP];
  __shared__ __attribute__((aligned(16))) bf16_t dots[D * TP];
  __shared__ __attribute__((aligned(16))) float lse_s[N];
  __shared__ __attribute__((aligned(16))) float delta_s[N];
  __shared__ __attribute__((aligned(16))) bf16_t tr[2][16 * TR];
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4, h = blockIdx.y;
  const int C = heads * D;
  const int64_t ld = 3 * C;
  for (int i = lane; i < TBL; i += 64) { tbl[i] = table[i * heads + h] * LOG2E; dtb[i] = 0.f; }

  for (int u = 0; u < chunk; ++u) {
    const int b = blockIdx.x * chunk + u;
    if (b >= B_) break;
    const bf16_t *base = qkv + (int64_t)b * N * ld + h * D;
    const bf16_t *dob = dout + (int64_t)b * N * C + h * D;
    const bf16_t *ob = out + (int64_t)b * N * C + h * D;
    bf16_t *dbase = dqkv + (int64_t)b * N * ld + h * D;
    __syncthreads();
    stage_transposed(base, ld, qts, lane);
    stage_transposed(dob, C, dots, lane);
    for (int row = lane; row < N; row += 64) {                          // delta = rowsum(dO * O), lse
      float acc = 0.f;
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        const uint4 a = *reinterpret_cast<const uint4 *>(dob + (int64_t)row * C + ch * 8);
        const uint4 o = *reinterpret_cast<const uint4 *>(ob + (int64_t)row * C + ch * 8);
        acc += bf_lo(a.x) * bf_lo(o.x) + bf_hi(a.x) * bf_hi(o.x) + bf_lo(a.y) * bf_lo(o.y) + bf_hi(a.y) * bf_hi(o.y)
             + bf_lo(a.z) * bf_lo(o.z) + bf_hi(a.z) * bf_hi(o.z) + bf_lo(a.w) * bf_lo(o.w) + bf_hi(a.w) * bf_hi(o.w);
      }
      delta_s[row] = acc;
      lse_s[row] = lse[((int64_t)b * heads + h) * N + row];
    }
    bf16x4 qf[NT][2], dof[NT][2];
    f32x4 dq[NT][2];
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
      qf[qt][0] = ld4(base + (16 * qt + c) * ld + 4 * g);
      qf[qt][1] = ld4(base + (16 * qt + c) * ld + 16 + 4 * g);
      dof[qt][0] = ld4(dob + (int64_t)(16 * qt + c) * C + 4 * g);
      dof[qt][1] = ld4(dob + (int64_t)(16 * qt + c) * C + 16 + 4 * g);
      dq[qt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      dq[qt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int w = b % nW;
    const bool masked = MASK && flags[w] != 0;
    const uint8_t *r = MASK ? region + w * N : nullptr;
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < NT; ++kt) {
      const int key = 16 * kt + c;
      const bf16_t *kp = base + C + key * ld, *vp = base + 2 * C + key * ld;
      const bf16x4 k0 = ld4(kp + 4 * g), k1 = ld4(kp + 16 + 4 * g), v0 = ld4(vp + 4 * g), v1 = ld4(vp + 16 + 4 * g);
      const bf16_t *ktp = base + C + (16 * kt + 4 * g) * ld + c;        // Kt[d = c (+16)][keys 16kt + 4g ..+3]
      const bf16x4 kt0 = gather4(ktp, (int)ld), kt1 = gather4(ktp + 16, (int)ld);
      f32x4 dk0 = {0.f, 0.f, 0.f, 0.f}, dk1 = dk0, dv0 = dk0, dv1 = dk0;
      const int ak = rel_a(key) - 264;
      const unsigned rk = masked ? r[key] : 0u;
#pragma unroll
      for (int qt = 0; qt < NT; ++qt) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = s;
        mma16(s, qf[qt][0], k0);                                        // S[q = 4g+e][key = c]
        mma16(s, qf[qt][1], k1);
        mma16(dp, dof[qt][0], v0);
        mma16(dp, dof[qt][1], v1);
        const int q0 = 16 * qt + 4 * g;
        const int ti = rel_a(q0) - ak;                                  // table index of (q0 + e, key) is ti + e
        const f32x4 L = *reinterpret_cast<const f32x4 *>(lse_s + q0), Dl = *reinterpret_cast<const f32x4 *>(delta_s + q0);
        unsigned rq = 0;
        if (masked) rq = *reinterpret_cast<const unsigned *>(r + q0);
        float p[4], ds[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = fmaf(s[e], c1, tbl[ti + e]);
          if (masked && ((rq >> (8 * e)) & 255u) != rk) t += MASKED;
          p[e] = __builtin_amdgcn_exp2f(t - L[e]);
          ds[e] = p[e] * (dp[e] - Dl[e]);
          atomicAdd(&dtb[ti + e], ds[e]);
        }
        const bf16x4 pb = pack4(p[0], p[1], p[2], p[3]);
        const bf16x4 dsb = pack4(ds[0] * scale, ds[1] * scale, ds[2] * scale, ds[3] * scale);
        const bf16x4 dot0 = lds4(dots + c * TP + q0), dot1 = lds4(dots + (16 + c) * TP + q0);
        const bf16x4 qt0 = lds4(qts + c * TP + q0), qt1 = lds4(qts + (16 + c) * TP + q0);
        mma16(dv0, dot0, pb);                                           // dV^T[d = 4g+e][key = c]
        mma16(dv1, dot1, pb);
        mma16(dk0, qt0, dsb);
        mma16(dk1, qt1, dsb);
        bf16_t *t = tr[qt & 1];                                         // dS tile -> lane: query c, keys 4g..4g+3
        t[(4 * g + 0) * TR + c] = (bf16_t)dsb[0]; t[(4 * g + 1) * TR + c] = (bf16_t)dsb[1];
        t[(4 * g + 2) * TR + c] = (bf16_t)dsb[2]; t[(4 * g + 3) * TR + c] = (bf16_t)dsb[3];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");           // one wave: LDS executes in issue order
        const bf16x4 dst = lds4(t + c * TR + 4 * g);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        mma16(dq[qt][0], kt0, dst);                                     // dQ^T[d = 4g+e][q = c]
        mma16(dq[qt][1], kt1, dst);
      }
      bf16_t *dkp = dbase + C + key * ld + 4 * g, *dvp = dbase + 2 * C + key * ld + 4 * g;
      st4(dkp, dk0[0], dk0[1], dk0[2], dk0[3]);
      st4(dkp + 16, dk1[0], dk1[1], dk1[2], dk1[3]);
      st4(dvp, dv0[0], dv0[1], dv0[2], dv0[3]);
      st4(dvp + 16, dv1[0], dv1[1], dv1[2], dv1[3]);
    }
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
      bf16_t *dqp = dbase + (16 * qt + c) * ld + 4 * g;
      st4(dqp, dq[qt][0][0], dq[qt][0][1], dq[qt][0][2], dq[qt][0][3]);
      st4(dqp + 16, dq[qt][1][0], dq[qt][1][1], dq[qt][1][2], dq[qt][1][3]);
    }
  }
  __syncthreads();
  for (int i = lane; i < TBL; i += 64) atomicAdd(dtable + i * heads + h, dtb[i]);
}

int check(const void *qkv, const float *table, const uint8_t *region, const uint8_t *flags, int B_, int nW, int heads, const char *who)
{
  if (B_ < 0 || nW <= 0 || heads <= 0 || (B_ % nW) != 0)
    return pd_set_error(PD_ERR_INVALID_ARG, "%s: bad sizes B_=%d nW=%d heads=%d (B_ must be a multiple of nW)", who, B_, nW, heads);
  if (B_ == 0) return PD_OK;
  if (!qkv || !table) return pd_set_error(PD_ERR_INVALID_ARG, "%s: null input", who);
  if ((region == nullptr) != (flags == nullptr)) return pd_set_error(PD_ERR_INVALID_ARG, "%s: region and win_flags go together", who);
  if (((uintptr_t)qkv & 15) != 0) return pd_set_error(PD_ERR_INVALID_ARG, "%s: qkv must be 16-byte aligned", who);
  if (region && ((uintptr_t)region & 3) != 0) return pd_set_error(PD_ERR_INVALID_ARG, "%s: region must be 4-byte aligned", who);
  return PD_OK;
}

// windows one wavefront walks: enough wavefronts to fill 256 CUs x 4 SIMDs a few times over, at most 16 windows each
int chunk_of(int B_, int heads)
{
  const int64_t units = (int64_t)B_ * heads;
  int chunk = (int)(units / 8192);
  return chunk < 1 ? 1 : (chunk > 16 ? 16 : chunk);
}
}  // namespace

extern "C" int pd_window_attn_fwd_w12(const void *qkv, const float *table, const uint8_t *region, const uint8_t *win_flags,
                                      void *out, float *lse, int B_, int nW, int heads, float scale, void *stream_)
{
  int rc = check(qkv, table, region, win_flags, B_, nW, heads, "pd_window_attn_fwd_w12");
  if (rc || B_ == 0) return rc;
  if (!out || !lse) return pd_set_error(PD_ERR_INVALID_ARG, "pd_window_attn_fwd_w12: null output");
  const int chunk = chunk_of(B_, heads);
  const dim3 grid((B_ + chunk - 1) / chunk, heads);
  hipStream_t s = (hipStream_t)stream_;
  if (region) hipLaunchKernelGGL(wattn_fwd<true>, grid, dim3(64), 0, s, (const bf16_t *)qkv, table, region, win_flags, (bf16_t *)out, lse, B_, nW, heads, scale * LOG2E, chunk);
  else hipLaunchKernelGGL(wattn_fwd<false>, grid, dim3(64), 0, s, (const bf16_t *)qkv, table, region, win_flags, (bf16_t *)out, lse, B_, nW, heads, scale * LOG2E, chunk);
  return pd_check_launch("pd_window_attn_fwd_w12");
}

extern "C" int pd_window_attn_bwd_w12(const void *qkv, const float *table, const uint8_t *region, const uint8_t *win_flags,
                                      const void *out, const void *d_out, const float *lse, void *dqkv, float *dtable, int B_,
                                      int nW, int heads, float scale, void *stream_)
{
  int rc = check(qkv, table, region, win_flags, B_, nW, heads, "pd_window_attn_bwd_w12");
  if (rc || B_ == 0) return rc;
  if (!out || !d_out || !lse || !dqkv || !dtable) return pd_set_error(PD_ERR_INVALID_ARG, "pd_window_attn_bwd_w12: null pointer");
  if ((((uintptr_t)out | (uintptr_t)d_out | (uintptr_t)dqkv) & 15) != 0)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_window_attn_bwd_w12: out / d_out / dqkv must be 16-byte aligned");
  const int chunk = chunk_of(B_, heads);
  const dim3 grid((B_ + chunk - 1) / chunk, heads);
  hipStream_t s = (hipStream_t)stream_;
  if (region) hipLaunchKernelGGL(wattn_bwd<true>, grid, dim3(64), 0, s, (const bf16_t *)qkv, table, region, win_flags, (const bf16_t *)out, (const bf16_t *)d_out, lse, (bf16_t *)dqkv, dtable, B_, nW, heads, scale, scale * LOG2E, chunk);
  else hipLaunchKernelGGL(wattn_bwd<false>, grid, dim3(64), 0, s, (const bf16_t *)qkv, table, region, win_flags, (const bf16_t *)out, (const bf16_t *)d_out, lse, (bf16_t *)dqkv, dtable, B_, nW, heads, scale, scale * LOG2E, chunk);
  return pd_check_launch("pd_window_attn_bwd_w12");
}
