// fp32 GEMM on the fp16 matrix cores with TWO planes per operand and THREE products per term (round 3).
//
// gemm_x3.hip splits an fp32 element exactly into three bf16 values (3 x 8 significand bits) and needs six MFMA products per
// term.  fp16 carries 11 significand bits, so two planes hold 22 of fp32's 24:
//     x' = 2^s x,   hi = fp16(x'),   lo = fp16(x' - hi)            |x' - hi - lo| <= 2^-22 |x'|
//     a b ~= hi_a hi_b + hi_a lo_b + lo_a hi_b                      (dropped: lo_a lo_b <= 2^-22 |a b|)
// — three v_mfma_f32_32x32x16_f16 (same rate as the bf16 form) instead of six, two LDS planes (4 bytes per element, what the
// fp32 tile itself takes) instead of three.  The price is fp16's 5-bit exponent: every operand ROW is scaled by a power of two
// taken from that row's absolute maximum (the caller passes row maxima it already has: the kernel that produced the operand
// emits them — row-wise kernels as a plain store, the GEMM epilogues below through an atomic max per row), so the largest
// element of a row lands in [2^14, 2^15) and
//     per-element error <= max(2^-22 |x|, 2^-39 max_row|x|)        (fp16 subnormals are kept by the matrix cores; if they were
// flushed the second term would be 2^-29) — below fp32's own 2^-24 relative to the row maximum, i.e. the product has the
// normwise accuracy of an fp32 GEMM (measured against fp64 in tests/test_gemm_gpu.py: max error / max|C| ~ 3e-7 at K = 256..1024,
// the exact-fp32 library GEMM ~ 2e-7, the 3-plane bf16 kernel ~ 3.4e-7).  The row scales are undone in the epilogue
// (powers of two: exact).  A NULL maxima pointer means "this operand is O(1)": scale 1.
//
//   gemm_tn_f16x2<TM, TN, BKK, ...>   C[M,N] = A[M,K] B[N,K]^T (+bias)(ReLU | ReLU mask + column sums)(row maxima of C out)
// Reference contract: the fp32 pixel decoder (pixel_decoder/msdeformattn.py:318 autocast(enabled=False); :120-135 FFN,
// ops/modules/ms_deform_attn.py:102-130 projections).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "f16x2.h"
#include "pd_common.h"
#include "xcd.h"
#include "pd_gemm.h"
#include "pd_msda.h"

// tools/ablate_gemm_h2.sh builds diagnostic copies of this file with -DPD_ABL=<bits> (1: no MFMA, 2: no LDS fragment reads, 4: no split /
// LDS writes, 8: no global operand loads, 16: no C stores, 32: unit scales without reading the maxima, 64: row stream without the weight loads); the product build has PD_ABL 0
#ifndef PD_ABL
#define PD_ABL 0
#endif

namespace {
using namespace pdh2;
__device__ __forceinline__ int xcd_chunk(int bid, int nb) { return pd_xcd_chunk(bid, nb); }   // xcd.h: any workgroup count
__device__ __forceinline__ unsigned short f2bf_rne(float f)
{
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// TM x TN tile, waves laid out (TM / 64) x (TN / WN), each wave 64 x WN = 2 x (WN / 32) MFMA tiles.
// LDS: [stage][plane][k panel of 8][row][8 halves]: an MFMA operand (8 consecutive k of row lane % 32, panel lane / 32 of the
// 16-wide sub-step) is ONE ds_read_b128 and the 32 lanes of a half read 512 contiguous bytes.
// MODE 0: C = A B^T + bias; 1: relu(...) and, when bits != NULL, its sign bits in accumulator order; 2: (A B^T) where the
// recorded bit is set, colsum += column sums.  c_amax != NULL: atomic max of |C| per row (as uint bits; caller zero-fills).
// CONV: A is an NHWC image [*, H, W, Ci] (lda = Ci) and row m of the GEMM is output pixel m of a 3 x 3, stride 1, pad 1
// convolution, K = 9 Ci ordered (tap, channel) like the channels-last filter [Co][3][3][Ci]: a 16-wide step lies inside one tap and
// its A tile is the input at pixel m + dy W + dx (zeros outside the image) — an implicit GEMM (gemm_x3.hip's CONV form).  The scale
// of output row m then has to cover the nine input pixels it reads: the largest of their maxima.
template <int TM, int TN, int WN, int BKK, int MODE, bool CONV = false, int NRS = 1, int FAST = 0>
__global__ __launch_bounds__((TM / 64) * (TN / WN) * 64, (TM == 256 ? 1 : 2))
void gemm_tn_f16x2(const float *__restrict__ A, const float *__restrict__ B, const float *__restrict__ bias, float *__restrict__ C,
                   int M, int N, int K, int lda, int ldb, int ldc, int ntiles_n, uint32_t *__restrict__ bits,
                   float *__restrict__ colsum, const float *__restrict__ a_amax, const float *__restrict__ b_amax,
                   unsigned *__restrict__ c_amax, int H, int W, int tap_minor)
{
  constexpr int WVN = TN / WN, NW = (TM / 64) * WVN, NTH = NW * 64, NJ = WN / 32;
  constexpr int TPR = BKK / 4, RPP = NTH / TPR, APASS = TM / RPP, BPASS = TN / RPP, NPAN = BKK / 8;
  static_assert(TM % RPP == 0 && TN % RPP == 0, "staging passes");
  constexpr int ASZ = 2 * NPAN * TM * 8, BSZ = 2 * NPAN * TN * 8;               // halves per stage and operand
  extern __shared__ __attribute__((aligned(16))) float sc[];                 // [0, TM): A scales, [TM, TM+TN): B scales, then the inverses
  h16_t *smem = reinterpret_cast<h16_t *>(sc + 2 * (TM + TN));
  auto As = [&](int buf, int pl, int pan, int r) -> h16_t * { return smem + buf * (ASZ + BSZ) + (((pl * NPAN + pan) * TM + r) << 3); };
  auto Bs = [&](int buf, int pl, int pan, int r) -> h16_t * { return smem + buf * (ASZ + BSZ) + ASZ + (((pl * NPAN + pan) * TN + r) << 3); };
  const int lb = xcd_chunk(blockIdx.x, gridDim.x);
  const int m0 = (lb / ntiles_n) * TM, n0 = (lb % ntiles_n) * TN;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = (wave / WVN) * 64, wn = (wave % WVN) * WN;
  const int lr = t / TPR, lk = (t % TPR) * 4;
  for (int r = t; r < TM + TN; r += NTH) {
    const bool isa = r < TM;
    const int g = isa ? m0 + r : n0 + r - TM;
    const float *am = isa ? a_amax : b_amax;
    float s = 1.f, inv = 1.f;
    if (!(PD_ABL & 32) && am && g < (isa ? M : N)) {
      float mx = am[g];
      if (CONV && isa) {
        const int pix = g % (H * W), y = pix / W, x = pix - y * W;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx)
            if (y + dy >= 0 && y + dy < H && x + dx >= 0 && x + dx < W) mx = fmaxf(mx, am[g + dy * W + dx]);
      }
      row_scale(mx, s, inv);
    }
    sc[r] = s; sc[TM + TN + r] = inv;
  }
  __syncthreads();
  float sa[APASS], sb[BPASS];
#pragma unroll
  for (int j = 0; j < APASS; ++j) sa[j] = sc[lr + RPP * j];
#pragma unroll
  for (int j = 0; j < BPASS; ++j) sb[j] = sc[TM + lr + RPP * j];
  float4 ra[NRS][APASS], rb[NRS][BPASS];                           // NRS register stages: global loads run NRS steps ahead of their split
  int py[APASS], px[APASS];                                        // CONV: image row / column of this thread's A rows
  if (CONV) {
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      const int pix = (m0 + lr + RPP * j) % (H * W);
      py[j] = pix / W;
      px[j] = pix - py[j] * W;
    }
  }
  auto gload = [&](int rs, int k0) {
    if (PD_ABL & 8) {
#pragma unroll
      for (int j = 0; j < APASS; ++j) ra[rs][j] = make_float4(1.f, 0.5f, 0.25f, (float)k0);
#pragma unroll
      for (int j = 0; j < BPASS; ++j) rb[rs][j] = make_float4(1.f, 0.5f, 0.25f, (float)k0);
      return;
    }
    const int k = k0 + lk;
    int dy = 0, dx = 0, kc = k, kb = k;                            // kc: channel of A, kb: column of B
    if (CONV) {
      // contraction order (channel block, tap), NOT (tap, channel): the nine taps of a 16-channel block re-touch the same three pixel
      // rows within nine steps, while they are still in the XCD's L2.  In (tap, channel) order a tap pass streams the whole 256 KB x 32
      // workgroups before the next tap comes back to the same lines: every tap was a fabric fetch (PMC: 1.37 GB per launch for 0.27 GB).
      int tap;
      if (tap_minor) { const int s = k0 / BKK; const int cb = s / 9; tap = s - cb * 9; kc = cb * BKK + lk; kb = tap * lda + kc; }
      else { tap = k / lda; kc = k - tap * lda; }
      dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1;
    }
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      const int r = lr + RPP * j;
      if (CONV) {
        const int yy = py[j] + dy, xx = px[j] + dx;
        const bool ok = m0 + r < M && k < K && yy >= 0 && yy < H && xx >= 0 && xx < W;
        ra[rs][j] = ok ? *reinterpret_cast<const float4 *>(A + ((int64_t)(m0 + r) + dy * W + dx) * lda + kc) : make_float4(0, 0, 0, 0);
      } else {
        ra[rs][j] = (m0 + r < M && k < K) ? *reinterpret_cast<const float4 *>(A + (int64_t)(m0 + r) * lda + k) : make_float4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < BPASS; ++j) {
      const int r = lr + RPP * j;
      rb[rs][j] = (n0 + r < N && k < K) ? *reinterpret_cast<const float4 *>(B + (int64_t)(n0 + r) * ldb + kb) : make_float4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int rs, int buf) {
    if (PD_ABL & 4) {
#pragma unroll
      for (int j = 0; j < APASS; ++j) asm volatile("" ::"v"(ra[rs][j].x), "v"(ra[rs][j].y), "v"(ra[rs][j].z), "v"(ra[rs][j].w));
#pragma unroll
      for (int j = 0; j < BPASS; ++j) asm volatile("" ::"v"(rb[rs][j].x), "v"(rb[rs][j].y), "v"(rb[rs][j].z), "v"(rb[rs][j].w));
      return;
    }
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
      const SplitH x = split4h(ra[rs][j], sa[j]);
      h16_t *p = As(buf, 0, lk >> 3, lr + RPP * j) + (lk & 7);
      *reinterpret_cast<uint2 *>(p) = x.hi; *reinterpret_cast<uint2 *>(p + NPAN * TM * 8) = x.lo;
    }
#pragma unroll
    for (int j = 0; j < BPASS; ++j) {
      const SplitH x = split4h(rb[rs][j], sb[j]);
      h16_t *p = Bs(buf, 0, lk >> 3, lr + RPP * j) + (lk & 7);
      *reinterpret_cast<uint2 *>(p) = x.hi; *reinterpret_cast<uint2 *>(p + NPAN * TN * 8) = x.lo;
    }
  };
  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int KT = (K + BKK - 1) / BKK;
  gload(0, 0);
  lstore(0, 0);
  if (KT > 1) gload(NRS == 2 ? 1 : 0, BKK);
  if (NRS == 2 && KT > 2) gload(0, 2 * BKK);
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
  auto step = [&](int kt, int par) {
    // register-staging order of the 3-plane kernel: tile kt + 1 is split and written right after the barrier, the loads of tile
    // kt + 2 re-issued into the same registers, then the wave turns to tile kt's fragments
    if (kt + 1 < KT) {
      const int rs = NRS == 2 ? (par ^ 1) : 0;                     // tile kt + 1 sits in register stage (kt + 1) & 1
      lstore(rs, par ^ 1);
      if (kt + 1 + NRS < KT) gload(rs, (kt + 1 + NRS) * BKK);
    }
#pragma unroll
    for (int ks = 0; ks < ((PD_ABL & 2) ? 0 : BKK / 16); ++ks) {
      h16x8 a[2][2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int i = 0; i < 2; ++i) a[pl][i] = *reinterpret_cast<const h16x8 *>(As(par, pl, 2 * ks + fh, wm + i * 32 + fr));
#pragma unroll
      for (int jp = 0; jp < NJ / 2; ++jp) {                        // two column tiles at a time
        h16x8 b[2][2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
          for (int j = 0; j < 2; ++j) b[pl][j] = *reinterpret_cast<const h16x8 *>(Bs(par, pl, 2 * ks + fh, wn + (jp * 2 + j) * 32 + fr));
#if PD_ABL & 1
#define HTERM(PA, PB)                                                        \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                              \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(a[PA][i]), "v"(b[PB][j]));
#else
#define HTERM(PA, PB)                                                        \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                              \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) mmah(acc[i][jp * 2 + j], a[PA][i], b[PB][j]);
#endif
        HTERM(1, 0)
        HTERM(0, 1)
        HTERM(0, 0)
#undef HTERM
      }
    }
    __syncthreads();
  };
  int kt0 = 0;
  if constexpr (FAST != 0 && BKK == 16) {
    // Interior workgroups (whole tile inside C, K a multiple of 32): the same step as ONE basic block — no bounds branches around
    // the loads (the tile after the last is clamped onto the last: an L2 hit nobody reads) — so that the scheduler may lay the
    // split's VALU work, the LDS stores and the next loads into the shadow of the matrix instructions (sched_group_barrier:
    // one MFMA, then its share of the VALU / DS / VMEM instructions).  The last two steps run through the guarded form below.
    // CONV: the halo test becomes a select on a load from an address that always exists (the centre pixel's), not a branch.
    if (m0 + TM <= M && n0 + TN <= N && (K % (2 * BKK)) == 0 && KT >= 4 && (!CONV || tap_minor)) {
      const float *pa[APASS], *pb[BPASS];
#pragma unroll
      for (int j = 0; j < APASS; ++j) pa[j] = A + (int64_t)(m0 + lr + RPP * j) * lda + lk;
#pragma unroll
      for (int j = 0; j < BPASS; ++j) pb[j] = B + (int64_t)(n0 + lr + RPP * j) * ldb + lk;
      auto fstep = [&](int kt, int par) {
        const int rs = NRS == 2 ? (par ^ 1) : 0;
        int kn = min((kt + 1 + NRS) * BKK, K - BKK), knb = kn, dy = 0, dx = 0;
        if (CONV) {                                                 // (channel block, tap) order: step s = 9 cb + tap
          const int sidx = kn / BKK, cb = sidx / 9, tap = sidx - cb * 9;
          dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1;
          kn = cb * BKK; knb = tap * lda + kn;
        }
        h16x8 a[2][2], b[2][NJ];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
          for (int i = 0; i < 2; ++i) a[pl][i] = *reinterpret_cast<const h16x8 *>(As(par, pl, fh, wm + i * 32 + fr));
#pragma unroll
          for (int j = 0; j < NJ; ++j) b[pl][j] = *reinterpret_cast<const h16x8 *>(Bs(par, pl, fh, wn + j * 32 + fr));
        }
#pragma unroll
        for (int j = 0; j < APASS; ++j) {
          const SplitH x = split4h(ra[rs][j], sa[j]);
          h16_t *p = As(par ^ 1, 0, lk >> 3, lr + RPP * j) + (lk & 7);
          *reinterpret_cast<uint2 *>(p) = x.hi; *reinterpret_cast<uint2 *>(p + NPAN * TM * 8) = x.lo;
          if (CONV) {
            const int yy = py[j] + dy, xx = px[j] + dx;
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
            const float4 v = *reinterpret_cast<const float4 *>(pa[j] + (ok ? (dy * W + dx) * lda : 0) + kn);
            ra[rs][j] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
          } else {
            ra[rs][j] = *reinterpret_cast<const float4 *>(pa[j] + kn);
          }
        }
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
          const SplitH x = split4h(rb[rs][j], sb[j]);
          h16_t *p = Bs(par ^ 1, 0, lk >> 3, lr + RPP * j) + (lk & 7);
          *reinterpret_cast<uint2 *>(p) = x.hi; *reinterpret_cast<uint2 *>(p + NPAN * TN * 8) = x.lo;
          rb[rs][j] = *reinterpret_cast<const float4 *>(pb[j] + knb);
        }
#define FTERM(PA, PB)                                                        \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                              \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) mmah(acc[i][j], a[PA][i], b[PB][j]);
        FTERM(1, 0)
        FTERM(0, 1)
        FTERM(0, 0)
#undef FTERM
        if constexpr (FAST == 1) {
          constexpr int NM = 6 * NJ, NV = (12 * (APASS + BPASS) + 8 + NM - 1) / NM, EV = NM / (APASS + BPASS);
          __builtin_amdgcn_sched_group_barrier(0x100, 4 + 2 * NJ, 0);
#pragma unroll
          for (int g = 0; g < NM; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
            if (g % EV == EV - 1) {
              __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
          }
        }
        __syncthreads();
      };
      for (; kt0 + 2 < KT; kt0 += 2) {
        fstep(kt0, 0);
        fstep(kt0 + 1, 1);
      }
    }
  }
  for (int kt = kt0; kt < KT; kt += 2) {
    step(kt, 0);
    if (kt + 1 < KT) step(kt + 1, 1);
  }
  // C layout of the 32x32 MFMA: col = lane & 31 (B row), row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) (A row).  Everything a store
  // depends on is loaded before the first store; interior tiles take a branch-free path (gemm_x3.hip, round 3).  Rows outside,
  // column tiles inside: the row's inverse scale is one LDS broadcast read and its maximum a scalar.
  float bv[NJ], ib[NJ], csum[NJ];
  uint32_t word[NJ];
  int64_t widx[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = n0 + wn + j * 32 + (lane & 31);
    bv[j] = (bias && col < N) ? bias[col] : 0.f;
    ib[j] = sc[2 * TM + TN + wn + j * 32 + (lane & 31)];
    widx[j] = (((int64_t)lb * NW + wave) * 64 + lane) * NJ + j;
    word[j] = MODE == 2 ? bits[widx[j]] : 0u;
    csum[j] = 0.f;
  }
  // row maxima of this wave's 64 x WN block go through the (now free) staging space, [wave][row][32 lanes + 1]
  float *red = reinterpret_cast<float *>(smem) + wave * 64 * 33;
  const bool full = m0 + TM <= M && n0 + TN <= N;
  auto store_tile = [&](auto guard) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rl = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), row = m0 + wm + rl;
        const float ia = sc[TM + TN + wm + rl];
        float rm = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int col = n0 + wn + j * 32 + (lane & 31);
          const bool ok = !decltype(guard)::value || (row < M && col < N);
          float v = acc[i][j][e] * (ia * ib[j]) + bv[j];
          if (MODE == 1) {
            v = fmaxf(v, 0.f);
            if (ok) word[j] |= (v > 0.f ? 1u : 0u) << (i * 16 + e);
          }
          if (MODE == 2) {
            v = ((word[j] >> (i * 16 + e)) & 1u) ? v : 0.f;
            if (ok) csum[j] += v;
          }
          if (PD_ABL & 16) { asm volatile("" ::"v"(v)); rm = fmaxf(rm, fabsf(v)); }
          else if (ok) {
            if (!CONV && (H & 1)) reinterpret_cast<unsigned short *>(C)[(int64_t)row * ldc + col] = f2bf_rne(v);   // H = flags when not a convolution
            else C[(int64_t)row * ldc + col] = v;
            rm = fmaxf(rm, fabsf(v));
          }
        }
        if (c_amax) red[rl * 33 + (lane & 31)] = rm;
      }
  };
  if (full) store_tile(std::false_type{});
  else store_tile(std::true_type{});
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int col = n0 + wn + j * 32 + (lane & 31);
    if (MODE == 1 && bits && col < N) bits[widx[j]] = word[j];
    if (MODE == 2) {
      float cs = csum[j];
      cs += __shfl_xor(cs, 32, 64);                                // the two row halves of the wavefront hold the same column
      if (lane < 32 && cs != 0.f && col < N) unsafeAtomicAdd(colsum + col, cs);
    }
  }
  if (c_amax) {
    // lane l owns row l of the block: ONE atomic max per row and wave (the bits of a non-negative float order like unsigned
    // integers).  The block is private to the wave and a wave's LDS operations complete in order: no barrier
    __builtin_amdgcn_wave_barrier();
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) v = fmaxf(v, red[lane * 33 + c]);
    const int row = m0 + wm + lane;
    if (row < M && v > 0.f) atomicMax(c_amax + row, __float_as_uint(v));
  }
}

// row maxima of an fp32 matrix [rows, cols] (row stride ld): for operands whose producer cannot emit them.  One wavefront per row.
__global__ __launch_bounds__(256) void row_amax_f32(const float *__restrict__ X, int rows, int cols, int ld, float *__restrict__ out)
{
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float *p = X + (int64_t)row * ld;
  float v = 0.f;
  if (!(cols & 3) && !(ld & 3) && !((uintptr_t)X & 15)) {
    for (int c = lane * 4; c < cols; c += 256) {
      const float4 x = *reinterpret_cast<const float4 *>(p + c);
      v = fmaxf(fmaxf(v, fmaxf(fabsf(x.x), fabsf(x.y))), fmaxf(fabsf(x.z), fabsf(x.w)));
    }
  } else {
    for (int c = lane; c < cols; c += 64) v = fmaxf(v, fabsf(p[c]));
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if (lane == 0) out[row] = v;
}


// bf16 rows -> their fp32 copy AND their absolute maxima in one pass (the backbone's bf16 feature maps entering the fp32 pixel decoder:
// was a cast launch + a row-maxima launch, i.e. the fp32 copy written and read back).  One wavefront per row, 8 channels per lane-step.
__global__ __launch_bounds__(256) void cast_bf16_f32_amax(const unsigned short *__restrict__ X, int rows, int cols, float *__restrict__ Y,
                                                          float *__restrict__ out)
{
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const unsigned short *p = X + (int64_t)row * cols;
  float *q = Y + (int64_t)row * cols;
  float v = 0.f;
  for (int c = lane * 8; c < cols; c += 512) {
    const uint4 u = *reinterpret_cast<const uint4 *>(p + c);
    float f[8];
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u); f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u); f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
#pragma unroll
    for (int k = 0; k < 8; ++k) v = fmaxf(v, fabsf(f[k]));
    *reinterpret_cast<float4 *>(q + c) = make_float4(f[0], f[1], f[2], f[3]);
    *reinterpret_cast<float4 *>(q + c + 4) = make_float4(f[4], f[5], f[6], f[7]);
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if (lane == 0) out[row] = v;
}

// ---------------------------------------------------------------------------------------------------------------------------
// gemm_rows_f16x2_k256: the K = 256 shapes of the encoder (value / output projections, the 1 x 1 convolutions on 256 channels;
// M = 43 520 .. 131 072 rows, N = 256 per panel) as a ROW STREAM.  The tiled kernel above runs all its workgroups at once and in
// phase — everyone loads, then everyone multiplies, then everyone stores (ablation, tools/ablate_gemm_h2.py: 10.5 us fixed + 9.5
// loads + 11 products = the 31 us of a launch whose traffic takes 11) — and re-splits the weight tile in every workgroup and step.
// Here ONE persistent workgroup per CU (8 wavefronts) keeps its whole weight panel split in REGISTERS — wavefront w holds columns
// 32 w .. 32 w + 31 of the panel for all of K as MFMA operands, 2 planes x 16 steps x 4 VGPRs = 128 — and streams 32-row tiles of
// A through a double-buffered 2-plane LDS image: A is read from HBM once, by one wavefront per row in whole 1 KB lines; loads run
// two tiles ahead in registers (64 KB per CU in flight all the time), one barrier per tile (48 MFMAs per wavefront) instead of one
// per 16-deep step, and the C stores of tile t drain under the products of tile t + 1.
// EXPERIMENTAL, not the product's choice (pd_debug_set("f16x2_tile", 61) selects it): correct (tests/test_gemm_gpu.py) and 3-8 % faster
// than the tiled kernel in isolation (43 520 x 256 <- 256: 31.0 vs 32.7 us), but 0.2 ms SLOWER over the training step (24.86 vs
// 24.65 ms, same-box A/B).  Its own ablation (tools/ablate_gemm_h2.py) shows why it stops there: the weight panel's fetch as MFMA
// fragments (32-byte pieces of 32 rows per load instruction, 256 KB per CU, 64 MB over the chip from L2) costs 8 us before the first
// product, and the fragment reads, the C stores and the products still add up (4.7 + 4.7 + 4.5 us) instead of overlapping.
// LDS image of a tile: [plane][k panel of 8 (32)][row (32)][8 halves], k panels 528 bytes apart (512 + 16: the 64 lanes of a row's
// wavefront store 8 bytes each into 32 different k panels — without the pad all of them on the same four banks).
constexpr int RS_PANEL = 528, RS_PLANE = 32 * RS_PANEL, RS_BUF = 2 * RS_PLANE;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE (round 5): 0 bias; 1 ReLU and its sign bits (16 per lane and tile, stored as one half-word at ((tile * npanels + panel) * 8 + wave) * 64 + lane:
// this kernel's own order — a mode-2 launch over the same [M, N] reads them back); 2 keep where the bit is set + column sums into `colsum`.
// the row stream stages with the UNPACKED split (f16x2.h): 256 <- 256 31.5 -> 29.9 us, 131 072 rows 73 -> 61 us, the step 21.44 -> 21.32 ms (three
// same-box A/B pairs); -DPD_ROWS_PACKED_SPLIT restores v_pk_mul_f32 / v_pk_fma_f32
#ifdef PD_ROWS_PACKED_SPLIT
#define ROWS_SPLIT split4h
#else
#define ROWS_SPLIT split4h_u
#endif
__device__ int g_rows_wstage = 1;      // 0 (PD_H2_ROWS_WSTAGE=0, set by the launcher through hipMemcpyToSymbol): the weight fragments straight from memory (A/B)
template <bool AM, int MODE = 0>   // AM: both operands come with row maxima (scaled rows); false: neither (unit scales)
__global__ __launch_bounds__(512)
void gemm_rows_f16x2_k256(const float *__restrict__ A, const float *__restrict__ B, const float *__restrict__ bias, float *__restrict__ C,
                          int M, int npanels, int lda, int ldb, int ldc, const float *__restrict__ a_amax,
                          const float *__restrict__ b_amax, unsigned *__restrict__ c_amax, unsigned short *__restrict__ bits = nullptr,
                          float *__restrict__ colsum = nullptr)
{
  // the two images are two objects: the stores into one provably do not alias the fragment reads of the other, so the scheduler
  // may move them (and the split feeding them) up between the matrix instructions
  __shared__ __attribute__((aligned(16))) unsigned char lds0[RS_BUF], lds1[RS_BUF];
  __shared__ __attribute__((aligned(16))) float sinv0[32], sinv1[32], sdump[64];    // inverse row scales of the staged tiles; unread slots
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), fr = lane & 31, fh = lane >> 5;   // w in an SGPR: row addresses and row maxima become scalar
  // workgroup -> (row group, panel): the panels of one row group sit on the same XCD (ids 8 apart) and share A in its L2
  const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int panel = q % npanels, rg = (q / npanels) * 8 + x, nrg = (gridDim.x / 8 / npanels) * 8;
  const int ntile = (M + 31) >> 5;
  const int t0 = (int)((int64_t)rg * ntile / nrg), t1 = (int)((int64_t)(rg + 1) * ntile / nrg), T = t1 - t0;
  if (T <= 0) return;
  const int n0 = panel * 256 + w * 32;

  float4 R[2][4];
  float am[2][4];
  auto gload = [&](int rs, int tile) {
    tile = min(tile, t1 - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = min(tile * 32 + w + 8 * j, M - 1);
      if (PD_ABL & 8) R[rs][j] = make_float4(1.f, 0.5f, 0.25f, (float)row);
      else R[rs][j] = *reinterpret_cast<const float4 *>(A + (int64_t)row * lda + 4 * lane);
      am[rs][j] = AM ? a_amax[row] : 0.f;
    }
  };
  auto split_store = [&](int rs, int buf) {
    float invs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = 1.f;
      invs[j] = 1.f;
      if (AM) row_scale(am[rs][j], s, invs[j]);
      const SplitH v = ROWS_SPLIT(R[rs][j], s);
      unsigned char *p = (buf ? lds1 : lds0) + (lane >> 1) * RS_PANEL + (w + 8 * j) * 16 + (lane & 1) * 8;
      if (PD_ABL & 4) { asm volatile("" ::"v"(R[rs][j].x), "v"(R[rs][j].y), "v"(R[rs][j].z), "v"(R[rs][j].w)); continue; }
      *reinterpret_cast<uint2 *>(p) = v.hi;
      *reinterpret_cast<uint2 *>(p + RS_PLANE) = v.lo;
    }
    // lanes 0..3 leave the four inverse scales; the other lanes store into unread slots (no branch in the step)
    const unsigned l0 = -(unsigned)(lane == 0), l1 = -(unsigned)(lane == 1), l2 = -(unsigned)(lane == 2);
    const unsigned iv = (__float_as_uint(invs[0]) & l0) | (__float_as_uint(invs[1]) & l1) | (__float_as_uint(invs[2]) & l2) |
                        (__float_as_uint(invs[3]) & ~(l0 | l1 | l2));
    *reinterpret_cast<unsigned *>(lane < 4 ? (buf ? sinv1 : sinv0) + w + 8 * lane : sdump + lane) = iv;
  };
  gload(0, t0);
  gload(1, t0 + 1);
  // this wavefront's 32 columns of the weight panel, split once: lane (fr, fh) holds k = 16 s + 8 fh .. + 7 of row n0 + fr
  h16x8 bh[16], bl[16];
  float ibv = 1.f;
  {
    float sbv = 1.f;
    if (AM) row_scale(b_amax[n0 + fr], sbv, ibv);
    if (g_rows_wstage) {
      // The wavefront's 32 x 256 weight slab arrives in four passes of 64 k — a load instruction reads 256 contiguous bytes of each of 4 rows —
      // and is dealt to the lanes through the still unused tile images: as MFMA fragments straight from memory an instruction touched 32 rows x
      // 32 bytes, the pattern in which one workgroup pulls ~30 GB/s (tools/probes/stream_probe.hip) — 256 KB per workgroup, ~8 us in front of
      // the first product of every launch.  Every pass fills fragments 4 R .. 4 R + 3 of ALL lanes (no divergence, no selects).
      constexpr int WP = 66;                                           // floats per staged row (264 bytes: 8-byte pieces, rows 2 banks apart)
      float *wl = reinterpret_cast<float *>(w < 4 ? lds0 : lds1) + (w & 3) * (32 * WP);    // 32 rows x 64 k per pass and wavefront: 4 x 8 448 bytes per image
      const float *bsrc = B + (int64_t)(n0 + (lane >> 4)) * ldb + 4 * (lane & 15);
      float *wdst = wl + (lane >> 4) * WP + 4 * (lane & 15);
      const float *wsrc = wl + fr * WP + 8 * fh;
      float4 wa[8], wb[8];                                             // the next pass's pieces fly while this one is dealt
#pragma unroll
      for (int i = 0; i < 8; ++i) wa[i] = *reinterpret_cast<const float4 *>(bsrc + (int64_t)(4 * i) * ldb);
#define PD_WPASS(R, CUR, NXT)                                                                                                  \
      {                                                                                                                        \
        if (R < 3) {                                                                                                           \
          _Pragma("unroll") for (int i = 0; i < 8; ++i) NXT[i] = *reinterpret_cast<const float4 *>(bsrc + (int64_t)(4 * i) * ldb + 64 * (R + 1)); \
        }                                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                        \
          *reinterpret_cast<float2 *>(wdst + 4 * i * WP) = make_float2(CUR[i].x, CUR[i].y);                                    \
          *reinterpret_cast<float2 *>(wdst + 4 * i * WP + 2) = make_float2(CUR[i].z, CUR[i].w);                                \
        }                                                                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       /* (a wavefront reads only what it wrote itself: no barrier) */ \
        _Pragma("unroll") for (int sq = 0; sq < 4; ++sq) {                                                                     \
          const float2 f0 = *reinterpret_cast<const float2 *>(wsrc + 16 * sq), f1 = *reinterpret_cast<const float2 *>(wsrc + 16 * sq + 2), \
                       f2 = *reinterpret_cast<const float2 *>(wsrc + 16 * sq + 4), f3 = *reinterpret_cast<const float2 *>(wsrc + 16 * sq + 6); \
          const SplitH u = split4h(make_float4(f0.x, f0.y, f1.x, f1.y), sbv), v = split4h(make_float4(f2.x, f2.y, f3.x, f3.y), sbv); \
          bh[4 * R + sq] = __builtin_bit_cast(h16x8, u32x4{u.hi.x, u.hi.y, v.hi.x, v.hi.y});                                   \
          bl[4 * R + sq] = __builtin_bit_cast(h16x8, u32x4{u.lo.x, u.lo.y, v.lo.x, v.lo.y});                                   \
        }                                                                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       /* the reads are done before the next pass overwrites the rows */ \
      }
      PD_WPASS(0, wa, wb) PD_WPASS(1, wb, wa) PD_WPASS(2, wa, wb) PD_WPASS(3, wb, wa)
#undef PD_WPASS
      __syncthreads();                                                 // everyone is done with its staging rows: the tile images are free
    } else {
      const float *bp = B + (int64_t)(n0 + fr) * ldb + 8 * fh;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const SplitH u = split4h((PD_ABL & 64) ? make_float4(1.f, 2.f, (float)s, sbv) : *reinterpret_cast<const float4 *>(bp + 16 * s), sbv),
                     v = split4h((PD_ABL & 64) ? make_float4(1.f, 2.f, (float)s, ibv) : *reinterpret_cast<const float4 *>(bp + 16 * s + 4), sbv);
        bh[s] = __builtin_bit_cast(h16x8, u32x4{u.hi.x, u.hi.y, v.hi.x, v.hi.y});
        bl[s] = __builtin_bit_cast(h16x8, u32x4{u.lo.x, u.lo.y, v.lo.x, v.lo.y});
      }
    }
  }
  const float bv = bias ? bias[n0 + fr] : 0.f;
  float csum = 0.f;                                                  // MODE 2: column sum of this lane's column over its rows
  split_store(0, 0);
  gload(0, t0 + 2);
  __syncthreads();

  auto iter = [&](int it, int par) {
    const unsigned char *base = (par ? lds1 : lds0) + fh * RS_PANEL + fr * 16;
    // MODE 2: the tile's sign bits are fetched HERE, under the 48 matrix instructions — read in the epilogue, where they are used, each tile's
    // stores waited for a global load's latency (the mode-2 launches ran 119 us against mode 1's 88 on the same shape)
    unsigned wbits = 0u;
    if (MODE == 2) wbits = (unsigned)bits[(((int64_t)(t0 + it) * npanels + panel) * 8 + w) * 64 + lane];
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
    for (int s = 0; s < ((PD_ABL & 1) ? 0 : 16); ++s) {
      const h16x8 ah = (PD_ABL & 2) ? bl[15 - s] : *reinterpret_cast<const h16x8 *>(base + 2 * s * RS_PANEL);
      const h16x8 al = (PD_ABL & 2) ? bh[15 - s] : *reinterpret_cast<const h16x8 *>(base + 2 * s * RS_PANEL + RS_PLANE);
#ifdef PD_ROWS_TWO_CHAINS
      if (s & 1) { mmah(acc1, al, bh[s]); mmah(acc0, ah, bl[s]); mmah(acc1, ah, bh[s]); }
      else       { mmah(acc0, al, bh[s]); mmah(acc1, ah, bl[s]); mmah(acc0, ah, bh[s]); }
#else
      mmah(acc0, al, bh[s]); mmah(acc0, ah, bl[s]); mmah(acc0, ah, bh[s]);
#endif
    }
    // tile it + 1 into the other image, its registers re-loaded with tile it + 3 (past the end: the last tile again, unread)
    split_store(par ^ 1, par ^ 1);
    gload(par ^ 1, t0 + it + 3);
    // lay the split (VALU), its LDS stores and the loads into the shadow of the 48 matrix instructions
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      if (s & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      if ((s & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);                              // the epilogue's VALU work is not material for the groups above
    // C rows of the 32 x 32 accumulator: (e & 3) + 8 (e >> 2) + 4 fh, column fr
    const int row0 = (t0 + it) * 32;
    float o[16];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const float4 ia = *reinterpret_cast<const float4 *>((par ? sinv1 : sinv0) + 8 * qd + 4 * fh);
      o[4 * qd + 0] = (acc0[4 * qd + 0] + acc1[4 * qd + 0]) * (ia.x * ibv) + bv;
      o[4 * qd + 1] = (acc0[4 * qd + 1] + acc1[4 * qd + 1]) * (ia.y * ibv) + bv;
      o[4 * qd + 2] = (acc0[4 * qd + 2] + acc1[4 * qd + 2]) * (ia.z * ibv) + bv;
      o[4 * qd + 3] = (acc0[4 * qd + 3] + acc1[4 * qd + 3]) * (ia.w * ibv) + bv;
    }
    if (MODE != 0) {
      const int64_t widx = (((int64_t)(t0 + it) * npanels + panel) * 8 + w) * 64 + lane;
      unsigned word = 0u;
      if (MODE == 2) { asm volatile("" : "+v"(wbits)); word = wbits; }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rl = (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (MODE == 1) {
          o[e] = fmaxf(o[e], 0.f);
          word |= (o[e] > 0.f ? 1u : 0u) << e;
        } else {
          o[e] = ((word >> e) & 1u) ? o[e] : 0.f;
          if (row0 + rl < M) csum += o[e];
        }
      }
      if (MODE == 1 && bits) bits[widx] = (unsigned short)word;
    }
    float *cp = C + (int64_t)row0 * ldc + n0 + fr;
    if (PD_ABL & 16) {
#pragma unroll
      for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(o[e]));
    } else if (row0 + 32 <= M) {
#pragma unroll
      for (int e = 0; e < 16; ++e) cp[(int64_t)((e & 3) + 8 * (e >> 2) + 4 * fh) * ldc] = o[e];
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rl = (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (row0 + rl < M) cp[(int64_t)rl * ldc] = o[e];
      }
    }
    if (c_amax) {
      // row maxima over this wavefront's 32 columns: a halving butterfly over the 32 lanes of a half (16 values -> 8 -> .. -> 1:
      // 15 exchanges instead of 16 x 5), lane pair (2 i, 2 i + 1) ends with row e = lane bits [4:1]
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = fabsf(o[e]);
#pragma unroll
      for (int n = 8, m = 16; n >= 1; n >>= 1, m >>= 1) {
        const bool up = (lane & m) != 0;
#pragma unroll
        for (int i = 0; i < n; ++i) {
          const float mine = up ? v[n + i] : v[i], send = up ? v[i] : v[n + i];
          v[i] = fmaxf(mine, __shfl_xor(send, m, 64));
        }
      }
      const float r = fmaxf(v[0], __shfl_xor(v[0], 1, 64));
      const int e = (lane >> 1) & 15, rl = (e & 3) + 8 * (e >> 2) + 4 * fh;
      if (!(lane & 1) && row0 + rl < M && r > 0.f) atomicMax(c_amax + row0 + rl, __float_as_uint(r));
    }
    __syncthreads();
  };
  for (int it = 0; it < T; it += 2) {
    iter(it, 0);
    if (it + 1 < T) iter(it + 1, 1);
  }
  if (MODE == 2) {                                                   // this wavefront's 32 columns over all its rows: one atomic per column
    csum += __shfl_xor(csum, 32, 64);
    if (lane < 32 && csum != 0.f) unsafeAtomicAdd(colsum + n0 + fr, csum);
  }
}


// (gemm_ra_f16x2_k256, gemm_kres_f16x2, split_planes_f16x2: opt-in kernels that lost their A/B — tools/probes/gemm_f16x2_optin.h, built only with -DPD_PROBES)
constexpr int KR_RB = 192, KR_KC = 32;
constexpr int KR_APANEL = KR_RB * 16 + 16, KR_APLANE = (KR_KC / 8) * KR_APANEL, KR_ABUF = 2 * KR_APLANE;
constexpr int KR_BPANEL = 256 * 16 + 16, KR_BPLANE = (KR_KC / 8) * KR_BPANEL, KR_BBUF = 2 * KR_BPLANE;
#ifdef PD_PROBES
constexpr size_t KR_LDS = (size_t)2 * (KR_ABUF + KR_BBUF) + KR_RB * sizeof(float);      // (gemm_kres_f16x2's layout: tools/probes)
#endif

// ---------------------------------------------------------------------------------------------------------------------------
// gemm_kpc_f16x2 (round 5): gemm_kres_f16x2's data path with PRODUCER and CONSUMER wavefronts instead of workgroup barriers.  Every structure of
// this family so far ends the same way: the wavefronts of a workgroup meet at a barrier per chunk, so they wait for memory, split, read
// fragments and multiply in step — the phases add up.  Here 4 producer wavefronts only load / split / store the chunk images (two register
// stages: loads two chunks ahead), 8 consumer wavefronts only read fragments and multiply (192 rows x 256 columns resident, as gemm_kres);
// they meet through two LDS counters per image — `full` (4 producer arrivals per chunk) and `empty` (8 consumer arrivals) — that a
// wavefront polls on its own (s_sleep between polls); no s_barrier inside the K loop.
constexpr size_t KP_LDS = (size_t)2 * (KR_ABUF + KR_BBUF) + 2 * KR_RB * sizeof(float) + 4 * sizeof(int);

#ifdef PD_PROBES
#include "../../tools/probes/gemm_f16x2_optin.h"
#endif
// tools only (KABL & 16): wall-clock stamps of workgroup 0's first consumer and first producer wavefront, read back by pd_debug_read_kpc_trace
__device__ unsigned long long g_kp_trace[2][64][8];
__device__ __forceinline__ void kp_stamp(int role, int chunk, int ev)
{
  if (chunk < 64) g_kp_trace[role][chunk][ev] = __builtin_readcyclecounter();
}

// The counters and the data they guard both live in LDS, which serves a CU's wavefronts in order: the arrival that made the count was issued
// after the LDS accesses it reports (s_waitcnt before it), so a RELAXED poll and a compiler barrier are enough.  An acquire load makes the
// compiler wait for vmcnt(0) as well — every global load the producer has in flight (tools/debug/kpc_trace.py: 3 000 of a chunk's 5 000 cycles).
__device__ __forceinline__ void kp_wait_ge(int *p, int target)
{
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}

template <bool AM, int KABL = 0, bool BP = false, bool CONV = false>     // KABL (tools only): 2 no products, 4 no fragment reads, 8 no global loads in the loop, 1 no split / LDS stores
// BP: the weights come PRE-SPLIT — `B` points at two fp16 planes [2][256 npanels][K] (hi, lo; rows scaled by row_scale(b_amax), written by
// split_planes_f16x2) — and go global -> LDS without touching the VALU: the stamps (tools/debug/kpc_trace.py) show that the split's VALU work does
// not overlap the consumers' matrix instructions on a SIMD (a half's 970 cycles of split + stores become 3 050 while they multiply, 1 600 while
// they only read fragments): the chunk period is split time + product time, which is what "the phases add up" was in every kernel of this family.
__global__ __launch_bounds__(768)
void gemm_kpc_f16x2(const float *__restrict__ A, const float *__restrict__ B, const float *__restrict__ bias, float *__restrict__ C, int M, int K,
                    int lda, int ldb, int ldc, int npanels, const float *__restrict__ a_amax, const float *__restrict__ b_amax, int H = 0, int W = 0)
{
  // CONV: A is an NHWC image [*, H, W, Ci = lda], row m = output pixel m of a 3 x 3 / stride 1 / pad 1 convolution, K = 9 Ci in (tap, channel)
  // order (gemm_tn_f16x2's CONV form): a 32-deep chunk lies inside one tap; its rows are the input at pixel m + dy W + dx — the same byte offset
  // for every row, added to the row's buffer offset; rows whose tap falls outside the image get an out-of-range offset and read zeros.  A
  // row's scale covers the nine pixels it reads (the largest of their maxima).
  auto conv_amax = [&](int row) -> float {
    float mx = a_amax[row];
    const int pix = row % (H * W), y = pix / W, x = pix - y * W;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx)
        if (y + dy >= 0 && y + dy < H && x + dx >= 0 && x + dx < W) mx = fmaxf(mx, a_amax[row + dy * W + dx]);
    return mx;
  };
  extern __shared__ __attribute__((aligned(16))) unsigned char kp_lds[];
  unsigned char *const aimg = kp_lds, *const bimg = kp_lds + 2 * KR_ABUF;
  float *sinv = reinterpret_cast<float *>(kp_lds + 2 * (KR_ABUF + KR_BBUF));   // [2][KR_RB]: inverse row scales of the current and the next work item
  int *ctr = reinterpret_cast<int *>(sinv + 2 * KR_RB);                 // full[0], full[1], empty[0], empty[1]: counts over the whole launch
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nblk = (M + KR_RB - 1) / KR_RB, NC = K / KR_KC;
  // The two roles are two loops over the same work list.  NO barrier between work items: the counters run on over the whole launch (chunk g of
  // this workgroup uses image g & 1), so the producers stage the next item's first chunks while the consumers still store the last one's results
  // (with a barrier per item a K = 256 item cost ~33 us for ~10 us of chunks).
  if (t < 4) ctr[t] = 0;
  __syncthreads();
  if (w >= 8) {
    // ---------------- producers: 256 threads, 8 per row (32 k = 8 float4): A rows prow + 32 j (6), weight rows prow + 32 j (8)
    const int pt = t - 512, prow = pt >> 3, pc4 = pt & 7;
    int g0 = 0, seq = 0;                                                   // chunks this workgroup has staged before the current item; items done
    for (int work = blockIdx.x; work < nblk * npanels; work += gridDim.x, g0 += NC, ++seq) {
      const int blk = work / npanels, panel = work - blk * npanels, row0 = blk * KR_RB, c0 = panel * 256;
      if (pt < KR_RB) {                                                    // (slot seq & 1 was last read in the epilogue of item seq - 2: over before the consumers took item seq - 1's chunks)
        float sc = 1.f, inv = 1.f;
        if (AM) row_scale(CONV ? conv_amax(min(row0 + pt, M - 1)) : a_amax[min(row0 + pt, M - 1)], sc, inv);
        sinv[(seq & 1) * KR_RB + pt] = inv;
      }
      float sa[6], sb[8];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        float inv;
        sa[j] = 1.f;
        if (AM) row_scale(CONV ? conv_amax(min(row0 + prow + 32 * j, M - 1)) : a_amax[min(row0 + prow + 32 * j, M - 1)], sa[j], inv);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float inv;
        sb[j] = 1.f;
        if (AM) row_scale(b_amax[c0 + prow + 32 * j], sb[j], inv);
      }
      // a chunk is staged as two halves (A rows 0 .. 95 + weight rows 0 .. 127, then the rest), THREE halves in flight (84 registers; a whole
      // chunk per stage, 2 x 56, spilled).  The stamps of the first version (two halves in flight, tools/debug/kpc_trace.py) showed the
      // producers as the bottleneck: per chunk 970 cycles of split + stores per half, ~1 900 cycles waiting for loads issued a chunk earlier
      // (memory latency under this load is > 4 000 cycles), and the consumers idle 57 % of the time waiting for `full`.
      // buffer loads: one 32-bit byte offset per staged row (a 64-bit pointer each, 28 registers, spilled — and a spill reload waits for
      // vmcnt(0), i.e. for every load in flight); the chunk's k offset rides in the scalar offset; rows past M are out of range and read as 0
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A), 0, (int)((int64_t)M * lda * 4), 0x00020000);
      // BP: planes [2][npanels * 256][K] halves; piece q = pt + 256 j of a half: plane q >> 9, column (q & 511) >> 2 of the half's 128, k panel q & 3
      const int64_t bbytes = BP ? (int64_t)2 * npanels * 256 * K * 2 : (int64_t)(c0 + 256) * ldb * 4;
      const __amdgpu_buffer_rsrc_t rbs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B), 0, (int)bbytes, 0x00020000);
      unsigned oa[6], ob[8];
      unsigned pyx[6];                                                   // CONV: bit t = tap t of the staged row lies inside the image (0 past M)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int row = row0 + prow + 32 * j;
        oa[j] = ((unsigned)row * (unsigned)lda + 4u * pc4) * 4u;
        pyx[j] = 0u;
        if (CONV && row < M) {
          const int pix = row % (H * W), y = pix / W, x = pix - y * W;
#pragma unroll
          for (int t9 = 0; t9 < 9; ++t9) {
            const int dy = t9 / 3 - 1, dx = t9 % 3 - 1;
            pyx[j] |= (unsigned)(y + dy >= 0 && y + dy < H && x + dx >= 0 && x + dx < W) << t9;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (BP) {
          const int q = pt + 256 * (j & 3), h = j >> 2, plane = q >> 9, col = 128 * h + ((q & 511) >> 2), pan = q & 3;
          ob[j] = (unsigned)(((int64_t)plane * npanels * 256 + c0 + col) * K + 8 * pan) * 2u;
        } else {
          ob[j] = ((unsigned)(c0 + prow + 32 * j) * (unsigned)ldb + 4u * pc4) * 4u;
        }
      }
      float4 RA[3][3], RB[3][4];
      auto gload = [&](int rs, int u) {                                  // half u & 1 of chunk u >> 1
        const int kcl = min(u >> 1, NC - 1), ko = kcl * (KR_KC * 4), h = u & 1;
        int tap = 0, shift = 0;
        if (CONV) {
          // (Ci / 32 is a power of two — the dispatcher checks —: no integer division in the loop; tap / 3 by comparison)
          const int lcb = __builtin_ctz((unsigned)(lda / KR_KC));
          tap = kcl >> lcb;
          const int cb = kcl - (tap << lcb), t3 = (tap >= 3) + (tap >= 6), dy = t3 - 1, dx = tap - 3 * t3 - 1;
          shift = ((dy * W + dx) * lda + cb * KR_KC) * 4;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          unsigned off = h ? oa[3 + j] : oa[j];
          if (CONV) off = (((h ? pyx[3 + j] : pyx[j]) >> tap) & 1u) ? off + (unsigned)shift : 0x80000000u;
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra, off, CONV ? 0 : ko, 0);
          RA[rs][j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rbs, h ? ob[4 + j] : ob[j], BP ? ko / 2 : ko, 0);      // (planes: 2 bytes per k)
          RB[rs][j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
      };
      auto split_store = [&](int rs, int buf, int h) {
        unsigned char *ab = aimg + buf * KR_ABUF + (pc4 >> 1) * KR_APANEL + (pc4 & 1) * 8;
        unsigned char *bb = bimg + buf * KR_BBUF + (pc4 >> 1) * KR_BPANEL + (pc4 & 1) * 8;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const SplitH v = split4h_u(RA[rs][j], h ? sa[3 + j] : sa[j]);
          unsigned char *p = ab + (prow + 32 * (3 * h + j)) * 16;
          *reinterpret_cast<uint2 *>(p) = v.hi;
          *reinterpret_cast<uint2 *>(p + KR_APLANE) = v.lo;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (BP) {                                                        // a 16-byte piece of a plane IS an LDS slot: [k panel][column][8 halves]
            const int q = pt + 256 * j, plane = q >> 9, col = 128 * h + ((q & 511) >> 2), pan = q & 3;
            *reinterpret_cast<float4 *>(bimg + buf * KR_BBUF + plane * KR_BPLANE + pan * KR_BPANEL + col * 16) = RB[rs][j];
          } else {
            const SplitH v = split4h_u(RB[rs][j], h ? sb[4 + j] : sb[j]);
            unsigned char *p = bb + (prow + 32 * (4 * h + j)) * 16;
            *reinterpret_cast<uint2 *>(p) = v.hi;
            *reinterpret_cast<uint2 *>(p + KR_BPLANE) = v.lo;
          }
        }
      };
      gload(0, 0);
      gload(1, 1);
      gload(2, 2);
      const int NH = 2 * NC;
      for (int u0 = 0; u0 < NH; u0 += 6) {
#pragma unroll
        for (int k6 = 0; k6 < 6; ++k6) {
          const int u = u0 + k6;
          if (u >= NH) break;
          const int cc = u >> 1, gc = g0 + cc, par = gc & 1, h = k6 & 1, rs = k6 % 3;
          const bool tr = (KABL & 16) && blockIdx.x == 0 && w == 8 && lane == 0 && seq == 0;
          if (h == 0) {
            if (tr) kp_stamp(1, cc, 0);
            kp_wait_ge(ctr + 2 + par, 8 * (gc >> 1));                    // the consumers are done with this image's previous chunk
            if (tr) kp_stamp(1, cc, 1);
          }
          if (KABL & 16) { asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); if (tr) kp_stamp(1, cc, 2 + 2 * h); }   // this half's loads have landed (14 younger ones may be out)
          if (!(KABL & 1)) split_store(rs, par, h);
          if (!(KABL & 8)) gload(rs, u + 3);
          if (KABL & 16) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (tr) kp_stamp(1, cc, 3 + 2 * h); }
          if (h == 1 && lane == 0) __hip_atomic_fetch_add(ctr + par, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    return;
  }
  // ---------------- consumers: 2 row groups (96 rows) x 4 column groups (64 columns)
  const int fr = lane & 31, fh = lane >> 5, rgp = w >> 2, cg = w & 3;
  int g0 = 0, seq = 0;
  for (int work = blockIdx.x; work < nblk * npanels; work += gridDim.x, g0 += NC, ++seq) {
    const int blk = work / npanels, panel = work - blk * npanels, row0 = blk * KR_RB, n0 = panel * 256 + cg * 64;
    const float *sinv_it = sinv + (seq & 1) * KR_RB;
    float ibv[2] = {1.f, 1.f};
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      float sc = 1.f;
      if (AM) row_scale(b_amax[n0 + 32 * cb + fr], sc, ibv[cb]);
    }
    f32x16 acc[3][2];
#pragma unroll
    for (int ti = 0; ti < 3; ++ti)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ti][cb][e] = 0.f;
    for (int c = 0; c < NC; c += 2) {
#pragma unroll
      for (int par0 = 0; par0 < 2; ++par0) {
        const int cc = c + par0, gc = g0 + cc, par = gc & 1;
        if (cc >= NC) break;
        const bool tr = (KABL & 16) && blockIdx.x == 0 && w == 0 && lane == 0 && seq == 0;
        if (tr) kp_stamp(0, cc, 0);
        kp_wait_ge(ctr + par, 4 * ((gc >> 1) + 1));                      // all four producers have stored this chunk
        if (tr) kp_stamp(0, cc, 1);
        const unsigned char *ab = aimg + par * KR_ABUF + fh * KR_APANEL + (96 * rgp + fr) * 16;
        const unsigned char *bb = bimg + par * KR_BBUF + fh * KR_BPANEL + (64 * cg + fr) * 16;
#pragma unroll
        for (int s = 0; s < KR_KC / 16; ++s) {
          h16x8 ah[3], al[3], bh[2], bl[2];
#pragma unroll
          for (int ti = 0; ti < 3; ++ti) {
            if (KABL & 4) {
              ah[ti] = __builtin_bit_cast(h16x8, u32x4{(unsigned)s, (unsigned)ti, 1u, (unsigned)cc});
              al[ti] = ah[ti];
            } else {
              ah[ti] = *reinterpret_cast<const h16x8 *>(ab + 2 * s * KR_APANEL + ti * 32 * 16);
              al[ti] = *reinterpret_cast<const h16x8 *>(ab + 2 * s * KR_APANEL + ti * 32 * 16 + KR_APLANE);
            }
          }
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            if (KABL & 4) {
              bh[cb] = __builtin_bit_cast(h16x8, u32x4{(unsigned)s, (unsigned)cb, 3u, (unsigned)cc});
              bl[cb] = bh[cb];
            } else {
              bh[cb] = *reinterpret_cast<const h16x8 *>(bb + 2 * s * KR_BPANEL + cb * 32 * 16);
              bl[cb] = *reinterpret_cast<const h16x8 *>(bb + 2 * s * KR_BPANEL + cb * 32 * 16 + KR_BPLANE);
            }
          }
          if (KABL & 2) {
#pragma unroll
            for (int ti = 0; ti < 3; ++ti)
#pragma unroll
              for (int cb = 0; cb < 2; ++cb) acc[ti][cb][0] += (float)ah[ti][0] * (float)bl[cb][1] + (float)al[ti][2] * (float)bh[cb][3];
            continue;
          }
          if (KABL & 16) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (tr) kp_stamp(0, cc, 3 + s); }   // this step's ten fragments are in
#pragma unroll
          for (int ti = 0; ti < 3; ++ti)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) mmah(acc[ti][cb], al[ti], bh[cb]);
#pragma unroll
          for (int ti = 0; ti < 3; ++ti)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) mmah(acc[ti][cb], ah[ti], bl[cb]);
#pragma unroll
          for (int ti = 0; ti < 3; ++ti)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) mmah(acc[ti][cb], ah[ti], bh[cb]);
        }
        // (the fragments are in registers: every LDS read of this image has completed) hand the image back
        if (tr) kp_stamp(0, cc, 2);
        if (lane == 0) __hip_atomic_fetch_add(ctr + 2 + par, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
#pragma unroll
    for (int ti = 0; ti < 3; ++ti) {
      const int rl0 = 96 * rgp + 32 * ti;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int col = n0 + 32 * cb + fr;
        const float bv = bias ? bias[col] : 0.f;
        float *cp = C + (int64_t)(row0 + rl0) * ldc + col;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int rl = (e & 3) + 8 * (e >> 2) + 4 * fh;
          if (row0 + rl0 + rl < M) cp[(int64_t)rl * ldc] = acc[ti][cb][e] * (sinv_it[rl0 + rl] * ibv[cb]) + bv;
        }
      }
    }
  }
}
}  // namespace

static const bool g_rows_relu = []() { const char *e = getenv("PD_H2_ROWS_RELU"); return !e || e[0] != '0'; }();   // A/B switch (default on)
int g_pd_dbg_f16x2 = 0;   // tools/ only (pd_debug_set "f16x2_tile"): see pd_gemm_tn_f16x2

// the producer / consumer kernel runs ONE 192-row work item per workgroup at a time: a problem with few items (the FPN's res4 input projection:
// 43 on 256 CUs: 54 us) leaves most of the chip idle.  PD_H2_KPC_MIN_ITEMS=128 sends such problems to the tiled kernel (44-48 us for that launch) —
// measured over the step on two boxes: 18.19 -> 18.12 ms and 18.54 -> 18.57 ms, i.e. nothing; the default (0) keeps the producer / consumer kernel
static bool kpc_items_ok(int M, int N)
{
  static const int min_items = []() { const char *e = getenv("PD_H2_KPC_MIN_ITEMS"); return e ? atoi(e) : 0; }();
  return ((M + KR_RB - 1) / KR_RB) * (N / 256) >= min_items;
}

template <int TM, int TN, int WN, int BKK, bool CONV = false, int NRS = 1, int FAST = 0>
static int launch_f16x2(const float *A, const float *B, const float *bias, float *C, int M, int N, int K, int lda, int ldb, int ldc, int mode,
                        uint32_t *bits, float *colsum, const float *a_amax, const float *b_amax, float *c_amax, hipStream_t st, int H = 0,
                        int W = 0)
{
  constexpr int NTH = (TM / 64) * (TN / WN) * 64;
  // stages (2) x planes (2) x (TM + TN) rows x BKK halves + scales; the row-maxima reduction reuses the staging space (NW x 64 x 33 floats)
  constexpr size_t stage = (size_t)2 * 2 * (TM + TN) * BKK * sizeof(h16_t), red = (size_t)(NTH / 64) * 64 * 33 * sizeof(float);
  constexpr size_t lds = (stage > red ? stage : red) + (size_t)2 * (TM + TN) * sizeof(float);
  const int tn = (N + TN - 1) / TN, tm = (M + TM - 1) / TM;
  typedef void (*kfn)(const float *, const float *, const float *, float *, int, int, int, int, int, int, int, uint32_t *, float *, const float *,
                      const float *, unsigned *, int, int, int);
  const kfn k = CONV ? (kfn)gemm_tn_f16x2<TM, TN, WN, BKK, 0, CONV, NRS, FAST>
                     : mode == 0 ? (kfn)gemm_tn_f16x2<TM, TN, WN, BKK, 0, false, NRS, FAST>
                                 : mode == 1 ? (kfn)gemm_tn_f16x2<TM, TN, WN, BKK, 1, false, NRS, FAST> : (kfn)gemm_tn_f16x2<TM, TN, WN, BKK, 2, false, NRS, FAST>;
  static bool attr[3] = {false, false, false};
  if (!attr[mode]) { (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr[mode] = true; }
  hipLaunchKernelGGL(k, dim3((unsigned)((int64_t)tm * tn)), dim3(NTH), lds, st, A, B, bias, C, M, N, K, lda, ldb, ldc, tn, bits, colsum, a_amax,
                     b_amax, reinterpret_cast<unsigned *>(c_amax), H, W, g_pd_dbg_f16x2 == 21 ? 0 : 1);
  return pd_check_launch("pd_gemm_tn_f16x2");
}

extern "C" int64_t pd_gemm_tn_f16x2_bits_words(int M, int N)
{
  if (M <= 0 || N <= 0 || (N % 256)) return 0;
  return (int64_t)((M + 255) / 256) * (N / 256) * 8 * 64 * 4;
}

static int gemm_tn_f16x2_impl(const float *A, const float *B, const float *bias, float *C, uint32_t *bits, float *colsum, const float *a_amax,
                             const float *b_amax, float *c_amax, int M, int N, int K, int lda, int ldb, int ldc, int mode, void *stream_, int flags)
{
  static const bool wstage_set = []() {                      // PD_H2_ROWS_WSTAGE=0: the row stream's weight fragments straight from memory (A/B)
    const char *e = getenv("PD_H2_ROWS_WSTAGE");
    const int v = e ? atoi(e) : 1;
    if (v != 1) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rows_wstage), &v, sizeof(int));
    return true;
  }();
  (void)wstage_set;
  if (M < 0 || N < 0 || K < 0 || mode < 0 || mode > 2) return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f16x2: negative size / bad mode");
  if (M == 0 || N == 0) return PD_OK;
  if (!A || !B || !C || (mode == 2 && (!bits || !colsum))) return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f16x2: null pointer");
  if ((K & 3) || (lda & 3) || (ldb & 3) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f16x2: K, lda, ldb must be multiples of 4 and A, B 16-byte aligned");
  hipStream_t st = (hipStream_t)stream_;
  const int dbg = g_pd_dbg_f16x2;
  // (round 5: the row stream is the product's choice for these shapes again — 22.06 -> 22.00 ms per step in two same-box A/B pairs, where round 3
  // measured it 0.2 ms slower; PD_H2_ROWS_K256=0 / pd_debug_set("f16x2_tile", 80) keep the tiled kernel)
  static const bool rows_k256 = []() { const char *e = getenv("PD_H2_ROWS_K256"); return !e || e[0] != '0'; }();
  if ((dbg == 61 || (rows_k256 && dbg == 0)) && !flags && mode == 0 && K == 256 && (N % 256) == 0 && M >= 8192 && (a_amax == nullptr) == (b_amax == nullptr)) {
    // row stream (gemm_rows_f16x2_k256, experimental: see the kernel's header): one persistent workgroup per CU, panels of a row group on one XCD
    static int ncu = 0;
    if (!ncu) { int dev = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 8) ncu = 256; }
    const int np = N / 256, G = (ncu / (8 * np)) * 8 * np;
    if (G > 0) {
      if (a_amax)
        hipLaunchKernelGGL(gemm_rows_f16x2_k256<true>, dim3((unsigned)G), dim3(512), 0, st, A, B, bias, C, M, np, lda, ldb, ldc, a_amax, b_amax,
                           reinterpret_cast<unsigned *>(c_amax));
      else
        hipLaunchKernelGGL(gemm_rows_f16x2_k256<false>, dim3((unsigned)G), dim3(512), 0, st, A, B, bias, C, M, np, lda, ldb, ldc, a_amax, b_amax,
                           reinterpret_cast<unsigned *>(c_amax));
      return pd_check_launch("pd_gemm_tn_f16x2 (row stream)");
    }
  }
#ifdef PD_PROBES
  // K = 256, plain epilogue: the register-operand kernel (gemm_ra_f16x2_k256) — EXPERIMENTAL, pd_debug_set("f16x2_tile", 90) (100 + bits: its
  // ablations).  Correct (tests/test_gemm_gpu.py) and SLOWER than the tiled kernel: 35.6 vs 30.4 us at 43 008 x 256 <- 256, 99.6 vs 77.5 at 131 072
  // rows, 123 vs 91 at N = 1024.  Its ablation (tools/debug/ra_ablate.py, 32 768 rows, one round of 172 workgroups): 8.9 us with every load,
  // product and store removed (launch + weight staging skeleton), 14.8 with only the C stores, +6 for the A loads, +5 for the weight staging,
  // +7 for fragments and products — the kernel's phases ADD UP: one 132 KB workgroup per CU, one round, every workgroup stages, loads,
  // multiplies and stores in step with all the others, so memory is either read or written, never both.  The per-CU share of these shapes
  // (168 rows x 256 columns) is ~10 us of work under ~20 us of fixed cost, whatever the operand path; see DESIGN.md (round 5).
  if ((dbg == 90 || (dbg >= 100 && dbg < 132)) && mode == 0 && K == 256 && M >= 4096 && (N % 32) == 0 && N >= 96 &&
      (a_amax == nullptr) == (b_amax == nullptr)) {
    const int ntl = (N % 128) == 0 ? 4 : (N % 96) == 0 ? 3 : 0;
    if (ntl) {
      constexpr int NWV = 12;
      const int nblk_n = N / (ntl * 32), nblk_m = (M + NWV * 32 - 1) / (NWV * 32);
      const size_t lds = (size_t)2 * 32 * (ntl * 32 * 16 + RA_PSTRIDE_PAD) + (size_t)(ntl * 32 + NWV * 32) * sizeof(float);
      typedef void (*rfn)(const float *, const float *, const float *, float *, int, int, int, int, int, int, const float *, const float *, unsigned *, int);
      rfn kf = ntl == 4 ? (a_amax ? (rfn)gemm_ra_f16x2_k256<4, NWV, true> : (rfn)gemm_ra_f16x2_k256<4, NWV, false>)
                        : (a_amax ? (rfn)gemm_ra_f16x2_k256<3, NWV, true> : (rfn)gemm_ra_f16x2_k256<3, NWV, false>);
      if (ntl == 4 && a_amax && dbg >= 100 && dbg < 132) {             // tools: ablations (bit 1 no A loads, 2 no fragment reads, 4 no products, 8 no stores, 16 no weight staging)
        switch (dbg - 100) {
          case 1: kf = (rfn)gemm_ra_f16x2_k256<4, NWV, true, 1>; break;
          case 2: kf = (rfn)gemm_ra_f16x2_k256<4, NWV, true, 2>; break;
          case 4: kf = (rfn)gemm_ra_f16x2_k256<4, NWV, true, 4>; break;
          case 6: kf = (rfn)gemm_ra_f16x2_k256<4, NWV, true, 6>; break;
          case 8: kf = (rfn)gemm_ra_f16x2_k256<4, NWV, true, 8>; break;
          case 16: kf = (rfn)gemm_ra_f16x2_k256<4, NWV, true, 16>; break;
          case 31: kf = (rfn)gemm_ra_f16x2_k256<4, NWV, true, 31>; break;
          case 23: kf = (rfn)gemm_ra_f16x2_k256<4, NWV, true, 23>; break;
          default: break;
        }
        (void)hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      }
      static bool rattr[4] = {false, false, false, false};
      const int ai = (ntl == 4 ? 0 : 2) + (a_amax ? 1 : 0);
      if (!rattr[ai]) { (void)hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); rattr[ai] = true; }
      hipLaunchKernelGGL(kf, dim3((unsigned)(nblk_m * nblk_n)), dim3(NWV * 64), lds, st, A, B, bias, C, M, N, lda, ldb, ldc, nblk_n, a_amax, b_amax,
                         reinterpret_cast<unsigned *>(c_amax), flags);
      return pd_check_launch("pd_gemm_tn_f16x2 (register operands)");
    }
  }
#endif
  // N = 1024 <- K = 256 with the ReLU epilogues (the encoder FFN's first Linear and the input gradient of its second: modes 1 / 2 with sign
  // bits): the row stream in its own bit order.  The choice depends on (M, N, K) and the switches only, so the mode-1 launch that writes a
  // layer's bits and the mode-2 launch that reads them agree.  pd_debug_set("f16x2_tile", 62) / PD_H2_ROWS_RELU=0 keep the tiled kernel.
  if (dbg != 62 && dbg != 3 && dbg != 13 && dbg != 70 && !flags && (mode == 1 || mode == 2) && bits && K == 256 && (N % 256) == 0 && N >= 512 && M >= 8192 &&
      a_amax && b_amax && g_rows_relu) {
    static int ncu2 = 0;
    if (!ncu2) { int dev = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&ncu2, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu2 < 8) ncu2 = 256; }
    const int np = N / 256, G = (ncu2 / (8 * np)) * 8 * np;
    if (G > 0) {
      if (mode == 1)
        hipLaunchKernelGGL((gemm_rows_f16x2_k256<true, 1>), dim3((unsigned)G), dim3(512), 0, st, A, B, bias, C, M, np, lda, ldb, ldc, a_amax, b_amax,
                           reinterpret_cast<unsigned *>(c_amax), reinterpret_cast<unsigned short *>(bits), colsum);
      else
        hipLaunchKernelGGL((gemm_rows_f16x2_k256<true, 2>), dim3((unsigned)G), dim3(512), 0, st, A, B, bias, C, M, np, lda, ldb, ldc, a_amax, b_amax,
                           reinterpret_cast<unsigned *>(c_amax), reinterpret_cast<unsigned short *>(bits), colsum);
      return pd_check_launch("pd_gemm_tn_f16x2 (row stream, ReLU epilogue)");
    }
  }
#ifdef PD_PROBES
  // deep K, 256-column panels, plain epilogue: K outside, a row block's accumulators resident (gemm_kres_f16x2) — EXPERIMENTAL, opt-in
  // (PD_H2_KRES=1 / pd_debug_set("f16x2_tile", 91); 200 + bits: its ablations).  Correct (tests/test_gemm_gpu.py) and NOT faster: 256 <- 1024 at
  // M = 43 008 103 us against the tiled kernel's 92 on the same box (82 vs 85 on another).  tools/debug/kres_ablate.py: 17 us with an empty
  // loop (launch, prologue, the C stores), +27 staging (loads, split, LDS stores, barrier), +17 fragment reads, +40 products = 101: one
  // 116 KB workgroup per CU, eight wavefronts in step, and the compiler keeps the split's VALU block out of the matrix instructions' shadow
  // whatever sched_group_barrier asks for, and a hand-ordered stream (sched_barrier between the pieces, as the kernel has it now) changes nothing:
  // 256 VGPRs leave no room for a second set of fragments, both wavefronts of a SIMD wait for their LDS reads together — the phases add up.
  static const bool kres_env = []() { const char *e = getenv("PD_H2_KRES"); return e && e[0] == '1'; }();
  if (((kres_env && dbg == 0) || dbg == 91 || (dbg >= 200 && dbg < 216)) && !flags && mode == 0 && !bits && !c_amax && K >= 512 && (K % KR_KC) == 0 && (N % 256) == 0 && N <= 512 && M >= 8192 &&
      (a_amax == nullptr) == (b_amax == nullptr)) {
    static int ncu3 = 0;
    if (!ncu3) { int dev = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&ncu3, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu3 < 8) ncu3 = 256; }
    const int np = N / 256, nwork = ((M + KR_RB - 1) / KR_RB) * np, G = nwork < ncu3 ? nwork : ncu3;
    static bool kattr = false;
    if (!kattr) {
      (void)hipFuncSetAttribute((const void *)gemm_kres_f16x2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KR_LDS);
      (void)hipFuncSetAttribute((const void *)gemm_kres_f16x2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KR_LDS);
      kattr = true;
    }
    if (a_amax && dbg >= 200 && dbg < 216) {                           // tools: ablations of the loop's phases (timing only)
      typedef void (*kfn2)(const float *, const float *, const float *, float *, int, int, int, int, int, int, const float *, const float *);
      kfn2 kf = nullptr;
      switch (dbg - 200) {
        case 1: kf = gemm_kres_f16x2<true, 1>; break;
        case 2: kf = gemm_kres_f16x2<true, 2>; break;
        case 4: kf = gemm_kres_f16x2<true, 4>; break;
        case 6: kf = gemm_kres_f16x2<true, 6>; break;
        case 8: kf = gemm_kres_f16x2<true, 8>; break;
        case 9: kf = gemm_kres_f16x2<true, 9>; break;
        case 15: kf = gemm_kres_f16x2<true, 15>; break;
        default: kf = gemm_kres_f16x2<true, 0>; break;
      }
      (void)hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KR_LDS);
      hipLaunchKernelGGL(kf, dim3((unsigned)G), dim3(512), KR_LDS, st, A, B, bias, C, M, K, lda, ldb, ldc, np, a_amax, b_amax);
      return pd_check_launch("pd_gemm_tn_f16x2 (resident accumulators, ablation)");
    }
    if (a_amax) hipLaunchKernelGGL(gemm_kres_f16x2<true>, dim3((unsigned)G), dim3(512), KR_LDS, st, A, B, bias, C, M, K, lda, ldb, ldc, np, a_amax, b_amax);
    else hipLaunchKernelGGL(gemm_kres_f16x2<false>, dim3((unsigned)G), dim3(512), KR_LDS, st, A, B, bias, C, M, K, lda, ldb, ldc, np, a_amax, b_amax);
    return pd_check_launch("pd_gemm_tn_f16x2 (resident accumulators)");
  }
#endif
  // the same shapes with producer / consumer wavefronts (gemm_kpc_f16x2) — the PRODUCT'S CHOICE since its split uses unpacked VALU instructions
  // (f16x2.h split4h_u): 256 <- 1024 at M = 43 008 76-78 us against the tiled kernel's 92-94, the step 21.82 -> 21.65 ms (three same-box A/B pairs).
  // PD_H2_KPC=0 / pd_debug_set("f16x2_tile", 80) keep the tiled kernel; 92 / 93 select it with fp32 weights / pre-split weight planes.
  static const bool kpc_env = []() { const char *e = getenv("PD_H2_KPC"); return !e || e[0] != '0'; }();
  if (((kpc_env && dbg == 0) || dbg == 92 || dbg == 93 || dbg == 94 || (dbg >= 220 && dbg < 236)) && (int64_t)M * lda * 4 < (1ll << 31) && (int64_t)N * ldb * 4 < (1ll << 31) && !flags && mode == 0 && !bits && !c_amax && (K >= 512 || (dbg == 92 && K >= 128)) && (K % KR_KC) == 0 && K / KR_KC >= 3 /* the double-buffered inverse-scale slot is reused two chunks later */ && (N % 256) == 0 && (N <= 512 || dbg == 92) && M >= 8192 && kpc_items_ok(M, N) &&
      (a_amax == nullptr) == (b_amax == nullptr)) {
    static int ncu4 = 0;
    if (!ncu4) { int dev = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&ncu4, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu4 < 8) ncu4 = 256; }
    const int np = N / 256, nwork = ((M + KR_RB - 1) / KR_RB) * np, G = nwork < ncu4 ? nwork : ncu4;
    static bool pattr = false;
    if (!pattr) {
      (void)hipFuncSetAttribute((const void *)gemm_kpc_f16x2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KP_LDS);
      (void)hipFuncSetAttribute((const void *)gemm_kpc_f16x2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KP_LDS);
      pattr = true;
    }
#ifdef PD_PROBES
    if (a_amax && dbg >= 220 && dbg < 236 && dbg != 234) {             // tools: ablations of the two roles (timing only); 234: stamps of the planes form, below
      typedef void (*kfn3)(const float *, const float *, const float *, float *, int, int, int, int, int, int, const float *, const float *, int, int);
      kfn3 kf = nullptr;
      switch (dbg - 220) {
        case 1: kf = gemm_kpc_f16x2<true, 1>; break;
        case 2: kf = gemm_kpc_f16x2<true, 2>; break;
        case 4: kf = gemm_kpc_f16x2<true, 4>; break;
        case 6: kf = gemm_kpc_f16x2<true, 6>; break;
        case 8: kf = gemm_kpc_f16x2<true, 8>; break;
        case 9: kf = gemm_kpc_f16x2<true, 9>; break;
        case 15: kf = gemm_kpc_f16x2<true, 15>; break;
        case 12: kf = gemm_kpc_f16x2<true, 16>; break;        // 232: stamps (pd_debug_read_kpc_trace)
        case 13: kf = gemm_kpc_f16x2<true, 18>; break;        // 233: stamps, no products
        default: kf = gemm_kpc_f16x2<true, 0>; break;
      }
      (void)hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KP_LDS);
      hipLaunchKernelGGL(kf, dim3((unsigned)G), dim3(768), KP_LDS, st, A, B, bias, C, M, K, lda, ldb, ldc, np, a_amax, b_amax, 0, 0);
      return pd_check_launch("pd_gemm_tn_f16x2 (producer / consumer wavefronts, ablation)");
    }
    static const bool kpc_bp = []() { const char *e = getenv("PD_H2_KPC_PLANES"); return e && e[0] == '1'; }();   // (not faster once the split is unpacked: 78.5 vs 77 us)
    if ((kpc_bp && dbg != 94 && dbg != 92) || dbg == 93 || dbg == 234) {
      // weights pre-split into planes by one small launch (a per-process scratch: launches of one stream are ordered)
      static unsigned short *planes = nullptr;
      static int64_t planes_cap = 0;
      const int64_t need = (int64_t)2 * N * K;
      if (need > planes_cap) {
        if (planes) (void)hipFree(planes);
        if (hipMalloc(reinterpret_cast<void **>(&planes), (size_t)need * 2) != hipSuccess) { planes = nullptr; planes_cap = 0; return pd_set_error(PD_ERR_LAUNCH, "pd_gemm_tn_f16x2: no memory for the weight planes"); }
        planes_cap = need;
      }
      hipLaunchKernelGGL(split_planes_f16x2, dim3((unsigned)(((int64_t)N * (K / 4) + 255) / 256)), dim3(256), 0, st, B, ldb, b_amax, planes, N, K);
      static bool battr = false;
      if (!battr) {
        (void)hipFuncSetAttribute((const void *)(gemm_kpc_f16x2<true, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)KP_LDS);
        (void)hipFuncSetAttribute((const void *)(gemm_kpc_f16x2<false, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)KP_LDS);
        (void)hipFuncSetAttribute((const void *)(gemm_kpc_f16x2<true, 16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)KP_LDS);
        battr = true;
      }
      const float *bpl = reinterpret_cast<const float *>(planes);
      if (dbg == 234) hipLaunchKernelGGL((gemm_kpc_f16x2<true, 16, true>), dim3((unsigned)G), dim3(768), KP_LDS, st, A, bpl, bias, C, M, K, lda, ldb, ldc, np, a_amax, b_amax, 0, 0);
      else if (a_amax) hipLaunchKernelGGL((gemm_kpc_f16x2<true, 0, true>), dim3((unsigned)G), dim3(768), KP_LDS, st, A, bpl, bias, C, M, K, lda, ldb, ldc, np, a_amax, b_amax, 0, 0);
      else hipLaunchKernelGGL((gemm_kpc_f16x2<false, 0, true>), dim3((unsigned)G), dim3(768), KP_LDS, st, A, bpl, bias, C, M, K, lda, ldb, ldc, np, a_amax, b_amax, 0, 0);
      return pd_check_launch("pd_gemm_tn_f16x2 (producer / consumer wavefronts, weight planes)");
    }
#endif
    if (a_amax) hipLaunchKernelGGL(gemm_kpc_f16x2<true>, dim3((unsigned)G), dim3(768), KP_LDS, st, A, B, bias, C, M, K, lda, ldb, ldc, np, a_amax, b_amax, 0, 0);
    else hipLaunchKernelGGL(gemm_kpc_f16x2<false>, dim3((unsigned)G), dim3(768), KP_LDS, st, A, B, bias, C, M, K, lda, ldb, ldc, np, a_amax, b_amax, 0, 0);
    return pd_check_launch("pd_gemm_tn_f16x2 (producer / consumer wavefronts)");
  }
  // the sign bits are laid out in the 256 x 256 kernel's accumulator order: bits / mask launches must take that kernel
  const bool need_wide = bits != nullptr;
  if (need_wide && ((N % 256) || M < 1024)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f16x2: sign bits need N %% 256 == 0 and M >= 1024");
  const bool wide_ok = (N % 256) == 0 && M >= 1024;
  const bool by_shape = dbg == 0 || dbg == 61 || dbg == 70 || dbg == 80 || dbg == 62;
  static const bool wide_deepk = []() { const char *e = getenv("PD_H2_WIDE_DEEPK"); return e && e[0] == '1'; }();   // 256 x 256 tiles for N < 1024, K >= 512: faster in isolation (256 <- 1024: 80.7 vs 96.8 us), 0.15 ms SLOWER over the step (round 5 A/B: 22.01 vs 21.86 ms): off; PD_H2_WIDE_DEEPK=1 restores
  const bool wide = need_wide || dbg == 3 || dbg == 13 || (by_shape && wide_ok && (int64_t)((M + 255) / 256) * (N / 256) >= 128 && (N >= 1024 || (K >= 512 && wide_deepk)));
  static const bool fast_env = []() { const char *e = getenv("PD_H2_FAST"); return !e || e[0] != '0'; }();   // A/B switch: the interleaved interior step
#define GO(TM, TN, WN, NRS, FAST) return launch_f16x2<TM, TN, WN, 16, false, NRS, FAST>(A, B, bias, C, M, N, K, lda, ldb, ldc, mode, bits, colsum, a_amax, b_amax, c_amax, st, flags)
  // Measured and dropped (tools/bench_gemm_h2.py, M = 43 008): 32-deep steps (1024 <- 256 103.7 vs 96.8 us, 256 <- 1024 90.9 vs 85.4,
  // 256 <- 256 32.9 vs 29.3: half the workgroups in flight), 128 x 256 tiles with two workgroups per CU (no gain).
  // Two register stages (loads two steps ahead of their split): 97.0 vs 99.9 us on 1024 <- 256, 85.3 vs 87.9 on 256 <- 1024.
  // Interior workgroups take the branch-free, interleaved step (FAST = 1): 1024 <- 256 109 -> 92 us, 256 <- 1024 91 -> 79 us (wide tiles),
  // 256 <- 1024 94 -> 87 us (128 x 128 tiles) at M = 43 520; the step 24.9 -> 24.65 ms.
  // pd_debug_set("f16x2_tile"): 3 / 13 wide tiles with one / two register stages and the guarded step, 4 / 14 the same for 128 x 128
  // tiles, 70 the guarded step by shape, 61 the row stream where it applies, 21 the (tap, channel) contraction order of the convolution
  if (wide) {
    if (!wide_ok) return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f16x2: 256 x 256 tiles need N %% 256 == 0 and M >= 1024");
    if (dbg == 3) GO(256, 256, 128, 1, 0);
    if (dbg == 13 || dbg == 70) GO(256, 256, 128, 2, 0);
    if (!fast_env) GO(256, 256, 128, 2, 0);
    GO(256, 256, 128, 2, 1);
  }
  // (measured and dropped, round 6: 32-deep steps for the few-tile, long-contraction problems — the FPN's res5 input projection, 32 tiles x 128
  //  steps of 16: 70 us — run 76.6 us)
  if (dbg == 4 || dbg == 70) GO(128, 128, 64, 1, 0);
  if (dbg == 14 || !fast_env) GO(128, 128, 64, 2, 0);
  GO(128, 128, 64, 2, 1);
#undef GO
}

// which kernel pd_gemm_tn_f16x2 launches for a problem (tools / bench.py labels): 0 = 128 x 128 tiles, 1 = 256 x 256 tiles, 2 = the row stream
// (gemm_rows_f16x2_k256), 3 = the register-operand kernel (opt-in), 4 = resident accumulators (gemm_kres_f16x2, opt-in), 5 = the same with
// producer / consumer wavefronts (gemm_kpc_f16x2; a launch that asks for output row maxima takes the tiled kernel).  Mirrors the dispatch of gemm_tn_f16x2_impl.
extern "C" int pd_gemm_tn_f16x2_which(int M, int N, int K, int mode, int has_bits, int has_amax)
{
  const int dbg = g_pd_dbg_f16x2;
  static const bool rows_k256 = []() { const char *e = getenv("PD_H2_ROWS_K256"); return !e || e[0] != '0'; }();
  if ((dbg == 61 || (rows_k256 && dbg == 0)) && mode == 0 && K == 256 && (N % 256) == 0 && M >= 8192) return 2;
#ifdef PD_PROBES
  if ((dbg == 90 || (dbg >= 100 && dbg < 132)) && mode == 0 && K == 256 && M >= 4096 && (N % 32) == 0 && N >= 96 && ((N % 128) == 0 || (N % 96) == 0)) return 3;
#endif
  if (dbg != 62 && dbg != 3 && dbg != 13 && dbg != 70 && (mode == 1 || mode == 2) && has_bits && K == 256 && (N % 256) == 0 && N >= 512 && M >= 8192 && has_amax && g_rows_relu) return 2;
  static const bool kpc_env = []() { const char *e = getenv("PD_H2_KPC"); return !e || e[0] != '0'; }();
  if (((kpc_env && dbg == 0) || dbg == 92 || dbg == 93) && mode == 0 && !has_bits && K >= 512 && (K % KR_KC) == 0 && (N % 256) == 0 && N <= 512 && M >= 8192 && kpc_items_ok(M, N)) return 5;
#ifdef PD_PROBES
  static const bool kres_env = []() { const char *e = getenv("PD_H2_KRES"); return e && e[0] == '1'; }();
  if (((kres_env && dbg == 0) || dbg == 91) && mode == 0 && !has_bits && K >= 512 && (K % KR_KC) == 0 && (N % 256) == 0 && N <= 512 && M >= 8192) return 4;
#endif
  const bool wide_ok = (N % 256) == 0 && M >= 1024;
  const bool by_shape = dbg == 0 || dbg == 61 || dbg == 70 || dbg == 80 || dbg == 62;
  static const bool wide_deepk = []() { const char *e = getenv("PD_H2_WIDE_DEEPK"); return e && e[0] == '1'; }();
  const bool wide = has_bits || dbg == 3 || dbg == 13 || (by_shape && wide_ok && (int64_t)((M + 255) / 256) * (N / 256) >= 128 && (N >= 1024 || (K >= 512 && wide_deepk)));
  return wide ? 1 : 0;
}

extern "C" int pd_gemm_tn_f16x2(const float *A, const float *B, const float *bias, float *C, uint32_t *bits, float *colsum, const float *a_amax,
                                const float *b_amax, float *c_amax, int M, int N, int K, int lda, int ldb, int ldc, int mode, void *stream_)
{
  return gemm_tn_f16x2_impl(A, B, bias, C, bits, colsum, a_amax, b_amax, c_amax, M, N, K, lda, ldb, ldc, mode, stream_, 0);
}

extern "C" int pd_gemm_tn_f16x2_bf16out(const float *A, const float *B, const float *bias, void *C_bf16, const float *a_amax, const float *b_amax,
                                        int M, int N, int K, int lda, int ldb, int ldc, void *stream_)
{
  return gemm_tn_f16x2_impl(A, B, bias, reinterpret_cast<float *>(C_bf16), nullptr, nullptr, a_amax, b_amax, nullptr, M, N, K, lda, ldb, ldc, 0, stream_, 1);
}

extern "C" int pd_cast_bf16_f32_amax(const void *X_bf16, int rows, int cols, float *Y, float *row_amax, void *stream_)
{
  if (rows < 0 || cols <= 0 || (cols & 7)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_cast_bf16_f32_amax: rows=%d cols=%d (a multiple of 8)", rows, cols);
  if (rows == 0) return PD_OK;
  if (!X_bf16 || !Y || !row_amax || ((uintptr_t)X_bf16 & 15) || ((uintptr_t)Y & 15)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_cast_bf16_f32_amax: null / misaligned pointer");
  hipLaunchKernelGGL(cast_bf16_f32_amax, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, (const unsigned short *)X_bf16, rows, cols, Y,
                     row_amax);
  return pd_check_launch("pd_cast_bf16_f32_amax");
}

extern "C" int pd_conv3x3_nhwc_f16x2(const float *X, const float *Wk, const float *bias, float *Y, const float *x_amax, const float *w_amax,
                                     float *y_amax, int B, int H, int W, int Ci, int Co, void *stream_)
{
  if (B < 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || (Ci & 15))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_conv3x3_nhwc_f16x2: B=%d H=%d W=%d Ci=%d (%% 16) Co=%d", B, H, W, Ci, Co);
  if (B == 0) return PD_OK;
  if (!X || !Wk || !Y || ((uintptr_t)X & 15) || ((uintptr_t)Wk & 15)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_conv3x3_nhwc_f16x2: null / misaligned pointer");
  const int64_t M = (int64_t)B * H * W;
  if (M > 0x7fffffffLL - 4096) return pd_set_error(PD_ERR_INVALID_ARG, "pd_conv3x3_nhwc_f16x2: too many pixels");
  hipStream_t st = (hipStream_t)stream_;
#ifdef PD_PROBES
  // producer / consumer wavefronts (gemm_kpc_f16x2<.., CONV>): opt-in (PD_H2_KPC_CONV=1 / pd_debug_set("f16x2_tile", 92)) — correct, and at
  // config 2's 2 x 256^2 x 256 slower than the tiled kernel (0.496 vs 0.452 ms: 683 row blocks on 256 CUs are 2.67 rounds of 72 chunks each);
  // 3 % faster at 2 x 320^2 (0.808 vs 0.831 ms)
  static const bool kpc_conv = []() { const char *e = getenv("PD_H2_KPC_CONV"); return e && e[0] == '1'; }();
  if (((kpc_conv && g_pd_dbg_f16x2 == 0) || g_pd_dbg_f16x2 == 92 || g_pd_dbg_f16x2 == 234) && x_amax && w_amax && !y_amax && (Co % 256) == 0 && (Ci % KR_KC) == 0 && ((Ci / KR_KC) & (Ci / KR_KC - 1)) == 0 && M >= 8192 &&
      H < 65536 && W < 65536 && M * Ci * 4 < (1ll << 31) && (int64_t)Co * 9 * Ci * 4 < (1ll << 31)) {
    static int ncu5 = 0;
    if (!ncu5) { int dev = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&ncu5, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu5 < 8) ncu5 = 256; }
    const int np = Co / 256, nwork = (int)((M + KR_RB - 1) / KR_RB) * np, G = nwork < ncu5 ? nwork : ncu5;
    static bool cattr = false;
    if (!cattr) {
      (void)hipFuncSetAttribute((const void *)(gemm_kpc_f16x2<true, 0, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)KP_LDS);
      cattr = true;
    }
    if (g_pd_dbg_f16x2 == 234) {                                       // tools: phase stamps (tools/debug/kpc_trace.py conv)
      (void)hipFuncSetAttribute((const void *)(gemm_kpc_f16x2<true, 16, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)KP_LDS);
      hipLaunchKernelGGL((gemm_kpc_f16x2<true, 16, false, true>), dim3((unsigned)G), dim3(768), KP_LDS, st, X, Wk, bias, Y, (int)M, 9 * Ci, Ci, 9 * Ci, Co, np, x_amax,
                         w_amax, H, W);
      return pd_check_launch("pd_conv3x3_nhwc_f16x2 (producer / consumer wavefronts, stamps)");
    }
    hipLaunchKernelGGL((gemm_kpc_f16x2<true, 0, false, true>), dim3((unsigned)G), dim3(768), KP_LDS, st, X, Wk, bias, Y, (int)M, 9 * Ci, Ci, 9 * Ci, Co, np, x_amax,
                       w_amax, H, W);
    return pd_check_launch("pd_conv3x3_nhwc_f16x2 (producer / consumer wavefronts)");
  }
#endif
  const bool wide = g_pd_dbg_f16x2 == 3 || (g_pd_dbg_f16x2 != 4 && (Co % 256) == 0 && M >= 65536);
  if (g_pd_dbg_f16x2 == 70 || g_pd_dbg_f16x2 == 21) {               // the guarded step
    if (wide) return launch_f16x2<256, 256, 128, 16, true>(X, Wk, bias, Y, (int)M, Co, 9 * Ci, Ci, 9 * Ci, Co, 0, nullptr, nullptr, x_amax, w_amax, y_amax, st, H, W);
    return launch_f16x2<128, 128, 64, 16, true>(X, Wk, bias, Y, (int)M, Co, 9 * Ci, Ci, 9 * Ci, Co, 0, nullptr, nullptr, x_amax, w_amax, y_amax, st, H, W);
  }
  if (wide) return launch_f16x2<256, 256, 128, 16, true, 1, 1>(X, Wk, bias, Y, (int)M, Co, 9 * Ci, Ci, 9 * Ci, Co, 0, nullptr, nullptr, x_amax, w_amax, y_amax, st, H, W);
  return launch_f16x2<128, 128, 64, 16, true, 1, 1>(X, Wk, bias, Y, (int)M, Co, 9 * Ci, Ci, 9 * Ci, Co, 0, nullptr, nullptr, x_amax, w_amax, y_amax, st, H, W);
}

extern "C" int pd_row_amax_f32(const float *X, int rows, int cols, int ld, float *out, void *stream_)
{
  if (rows < 0 || cols < 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_row_amax_f32: negative size");
  if (rows == 0) return PD_OK;
  if (!X || !out) return pd_set_error(PD_ERR_INVALID_ARG, "pd_row_amax_f32: null pointer");
  hipLaunchKernelGGL(row_amax_f32, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, X, rows, cols, ld, out);
  return pd_check_launch("pd_row_amax_f32");
}

extern "C" int pd_debug_read_kpc_trace(unsigned long long *dst)
{
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_kp_trace), sizeof(unsigned long long) * 2 * 64 * 8) == hipSuccess ? PD_OK : PD_ERR_LAUNCH;
}
