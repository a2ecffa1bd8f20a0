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
#include <type_traits>

#include "f16x2.h"
#include "pd_common.h"
#include "pd_gemm.h"
#include "pd_msda.h"

// tools/ablate_gemm_h2.sh builds diagnostic copies of this file with -DPD_ABL=<bits> (1: no MFMA, 2: no LDS fragment reads, 4: no split /
// LDS writes, 8: no global operand loads, 16: no C stores, 32: unit scales without reading the maxima); the product build has PD_ABL 0
#ifndef PD_ABL
#define PD_ABL 0
#endif

namespace {
using namespace pdh2;
__device__ __forceinline__ int xcd_chunk(int bid, int nb) { return (nb & 7) == 0 ? (bid & 7) * (nb >> 3) + (bid >> 3) : bid; }

// TM x TN tile, waves laid out (TM / 64) x (TN / WN), each wave 64 x WN = 2 x (WN / 32) MFMA tiles.
// LDS: [stage][plane][k panel of 8][row][8 halves]: an MFMA operand (8 consecutive k of row lane % 32, panel lane / 32 of the
// 16-wide sub-step) is ONE ds_read_b128 and the 32 lanes of a half read 512 contiguous bytes.
// MODE 0: C = A B^T + bias; 1: relu(...) and, when bits != NULL, its sign bits in accumulator order; 2: (A B^T) where the
// recorded bit is set, colsum += column sums.  c_amax != NULL: atomic max of |C| per row (as uint bits; caller zero-fills).
// CONV: A is an NHWC image [*, H, W, Ci] (lda = Ci) and row m of the GEMM is output pixel m of a 3 x 3, stride 1, pad 1
// convolution, K = 9 Ci ordered (tap, channel) like the channels-last filter [Co][3][3][Ci]: a 16-wide step lies inside one tap and
// its A tile is the input at pixel m + dy W + dx (zeros outside the image) — an implicit GEMM (gemm_x3.hip's CONV form).  The scale
// of output row m then has to cover the nine input pixels it reads: the largest of their maxima.
template <int TM, int TN, int WN, int BKK, int MODE, bool CONV = false, int NRS = 1>
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
  for (int kt = 0; kt < KT; kt += 2) {
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
          else if (ok) { C[(int64_t)row * ldc + col] = v; rm = fmaxf(rm, fabsf(v)); }
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

}  // namespace

int g_pd_dbg_f16x2 = 0;   // tools/ only (pd_debug_set "f16x2_tile"): 1 force 256x256x32, 2 force 128x128x32, 3 force 256x256x16, 4 force 128x128x16 (0: by shape, 16-deep)

template <int TM, int TN, int WN, int BKK, bool CONV = false, int NRS = 1>
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
  const kfn k = CONV ? (kfn)gemm_tn_f16x2<TM, TN, WN, BKK, 0, CONV, NRS>
                     : mode == 0 ? (kfn)gemm_tn_f16x2<TM, TN, WN, BKK, 0, false, NRS> : mode == 1 ? (kfn)gemm_tn_f16x2<TM, TN, WN, BKK, 1, false, NRS>
                                                                                                  : (kfn)gemm_tn_f16x2<TM, TN, WN, BKK, 2, false, NRS>;
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

extern "C" int pd_gemm_tn_f16x2(const float *A, const float *B, const float *bias, float *C, uint32_t *bits, float *colsum, const float *a_amax,
                                const float *b_amax, float *c_amax, int M, int N, int K, int lda, int ldb, int ldc, int mode, void *stream_)
{
  if (M < 0 || N < 0 || K < 0 || mode < 0 || mode > 2) return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f16x2: negative size / bad mode");
  if (M == 0 || N == 0) return PD_OK;
  if (!A || !B || !C || (mode == 2 && (!bits || !colsum))) return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f16x2: null pointer");
  if ((K & 3) || (lda & 3) || (ldb & 3) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f16x2: K, lda, ldb must be multiples of 4 and A, B 16-byte aligned");
  hipStream_t st = (hipStream_t)stream_;
  // the sign bits are laid out in the 256 x 256 kernel's accumulator order: bits / mask launches must take that kernel
  const bool need_wide = bits != nullptr;
  if (need_wide && ((N % 256) || M < 1024)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f16x2: sign bits need N %% 256 == 0 and M >= 1024");
  const bool wide_ok = (N % 256) == 0 && M >= 1024;
  const bool wide = need_wide || g_pd_dbg_f16x2 == 1 || g_pd_dbg_f16x2 == 3 || g_pd_dbg_f16x2 == 13 || g_pd_dbg_f16x2 == 5 || g_pd_dbg_f16x2 == 15 ||
                    (g_pd_dbg_f16x2 == 0 && wide_ok && (int64_t)((M + 255) / 256) * (N / 256) >= 128 && (N >= 1024 || K >= 512));
#define GO(TM, TN, WN, BKK) return launch_f16x2<TM, TN, WN, BKK>(A, B, bias, C, M, N, K, lda, ldb, ldc, mode, bits, colsum, a_amax, b_amax, c_amax, st)
  // 16-deep steps beat 32-deep ones on every encoder shape (tools/bench_gemm_h2.py, M = 43 008: 1024 <- 256 96.8 vs 103.7 us,
  // 256 <- 1024 85.4 vs 90.9, 256 <- 256 29.3 vs 32.9): half the LDS per workgroup, more workgroups in flight
  if (wide) {
    if (!wide_ok) return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f16x2: 256 x 256 tiles need N %% 256 == 0 and M >= 1024");
    if (g_pd_dbg_f16x2 == 5) GO(128, 256, 128, 16);
    if (g_pd_dbg_f16x2 == 15) return launch_f16x2<128, 256, 128, 16, false, 2>(A, B, bias, C, M, N, K, lda, ldb, ldc, mode, bits, colsum, a_amax, b_amax, c_amax, st);
    if (g_pd_dbg_f16x2 == 1) GO(256, 256, 128, 32);
    // two register stages (loads two steps ahead of their split): 97.0 vs 99.9 us on 1024 <- 256, 85.3 vs 87.9 on 256 <- 1024; the
    // 128 x 128 kernel does not gain (its three workgroups per CU already interleave); 128 x 256 tiles with two workgroups per CU: no gain
    if (g_pd_dbg_f16x2 == 3) GO(256, 256, 128, 16);
    return launch_f16x2<256, 256, 128, 16, false, 2>(A, B, bias, C, M, N, K, lda, ldb, ldc, mode, bits, colsum, a_amax, b_amax, c_amax, st);
  }
  if (g_pd_dbg_f16x2 == 2) GO(128, 128, 64, 32);
  if (g_pd_dbg_f16x2 == 14) return launch_f16x2<128, 128, 64, 16, false, 2>(A, B, bias, C, M, N, K, lda, ldb, ldc, mode, bits, colsum, a_amax, b_amax, c_amax, st);
  GO(128, 128, 64, 16);
#undef GO
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
  const bool wide = g_pd_dbg_f16x2 == 3 || (g_pd_dbg_f16x2 != 4 && (Co % 256) == 0 && M >= 65536);
  if (wide) return launch_f16x2<256, 256, 128, 16, true>(X, Wk, bias, Y, (int)M, Co, 9 * Ci, Ci, 9 * Ci, Co, 0, nullptr, nullptr, x_amax, w_amax, y_amax, st, H, W);
  return launch_f16x2<128, 128, 64, 16, true>(X, Wk, bias, Y, (int)M, Co, 9 * Ci, Ci, 9 * Ci, Co, 0, nullptr, nullptr, x_amax, w_amax, y_amax, st, H, W);
}

extern "C" int pd_row_amax_f32(const float *X, int rows, int cols, int ld, float *out, void *stream_)
{
  if (rows < 0 || cols < 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_row_amax_f32: negative size");
  if (rows == 0) return PD_OK;
  if (!X || !out) return pd_set_error(PD_ERR_INVALID_ARG, "pd_row_amax_f32: null pointer");
  hipLaunchKernelGGL(row_amax_f32, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, X, rows, cols, ld, out);
  return pd_check_launch("pd_row_amax_f32");
}
