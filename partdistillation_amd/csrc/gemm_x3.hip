// fp32 GEMM on the bf16 matrix cores: every fp32 operand element is split EXACTLY into three bf16 values
//     x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)     (3 x 8 significand bits = fp32's 24)
// when a tile is staged in LDS, and a product a*b is accumulated in fp32 as six bf16 MFMA terms
//     lo_a hi_b + hi_a lo_b + mid_a mid_b + mid_a hi_b + hi_a mid_b + hi_a hi_b
// (each bf16 x bf16 product is exact in fp32; the three dropped terms mid*lo, lo*mid, lo*lo are <= 2^-23 |a b| together —
// the size of ONE fp32 rounding of the product).  v_mfma_f32_32x32x16_bf16 runs 16 x the rate of v_mfma_f32_32x32x2_f32,
// so six of them per step are 2.7x the exact-fp32 matrix rate: the fp32 Linears of the deformable-attention encoder
// (reference msdeformattn.py:120-135, forced fp32 by :318) get matrix-core speed without giving up fp32 results
// (measured against fp64 in tests/test_gemm_gpu.py: error at the level of the library's fp32 GEMM).
//
//   gemm_tn_f32x3   C[M,N] = A[M,K] B[N,K]^T (+bias)(ReLU)   128 x 128 x 16 tiles, 4 waves (2x2) x (2x2) MFMA tiles
// Two LDS stages of three bf16 planes per operand; the next tile's global loads are issued before the MFMAs of the
// current one, split and written to the other stage after them (one barrier per 16-wide step).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "f16x2.h"
#include "mfma_bf16.h"
#include "pd_common.h"
#include "xcd.h"
#include "pd_gemm.h"
#include "pd_msda.h"

namespace {
using namespace pdmfma;

constexpr int BM = 128, BN = 128, BK = 16, PITCH = BK + 4;      // bf16 elements per LDS row (40 bytes: 8-byte reads of 32 rows conflict-free)

struct Split4 { uint2 hi, mid, lo; };                            // 4 consecutive k of one row, per plane

__device__ __forceinline__ void split2(float x0, float x1, unsigned &h, unsigned &m, unsigned &l)
{
  h = pk_bf16(x0, x1);
  const float r0 = x0 - bf_lo(h), r1 = x1 - bf_hi(h);           // exact (Sterbenz / leading bits cancel)
  m = pk_bf16(r0, r1);
  l = pk_bf16(r0 - bf_lo(m), r1 - bf_hi(m));                    // the remainder has <= 8 significant bits: exact
}
__device__ __forceinline__ Split4 split4(float4 v)
{
  Split4 s;
  split2(v.x, v.y, s.hi.x, s.mid.x, s.lo.x);
  split2(v.z, v.w, s.hi.y, s.mid.y, s.lo.y);
  return s;
}
typedef __bf16 hwbf16x8 __attribute__((ext_vector_type(8)));
// operand of v_mfma_f32_32x32x16_bf16 (the gfx950 double-K form: 32 cycles per SIMD, twice the rate of the 32x32x8 form):
// lane l holds 8 consecutive k of row l % 32, starting at 8 * (l / 32) of the 16-wide step.  Two 8-byte LDS reads (the
// 72-byte row pitch is conflict-free for those).
__device__ __forceinline__ hwbf16x8 frag(const bf16_t *p)
{
  union { uint2 h[2]; hwbf16x8 v; } u;
  u.h[0] = *reinterpret_cast<const uint2 *>(p);
  u.h[1] = *reinterpret_cast<const uint2 *>(p + 4);
  return u.v;
}
__device__ __forceinline__ void mma16k(f32x16 &c, hwbf16x8 x, hwbf16x8 y) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0); }

__device__ __forceinline__ int xcd_chunk(int bid, int nb) { return pd_xcd_chunk(bid, nb); }   // xcd.h: any workgroup count

__device__ int g_wgrad_xcd = 1;   // tools only (pd_debug_set "wgrad_xcd" 0: launch order = logical order)
__device__ int g_wgrad_fast = 1;  // tools only (pd_debug_set "wgrad_fast" 0: the guarded step everywhere)
__device__ __forceinline__ bool v_never(float x) { return x == 1.2345678e30f; }

// CONV: A is an NHWC image [*, H, W, Ci] and row m of the GEMM is output pixel m of a 3 x 3, stride 1, pad 1 convolution:
// K = 9 Ci ordered (tap, channel) like the channels-last filter [Co][3][3][Ci]; a 16-wide k-step lies inside one tap, so the
// A tile of a step is the input at pixel m + dy W + dx (zeros outside the image) — an implicit GEMM, nothing is unfolded.
template <bool RELU, int ABL, bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_tn_f32x3(const float *__restrict__ A, const float *__restrict__ B,
                                                         const float *__restrict__ bias, float *__restrict__ C, int M, int N,
                                                         int K, int lda, int ldb, int ldc, int ntiles_n, int H, int W)
{
  __shared__ __attribute__((aligned(16))) bf16_t As[2][3][BM][PITCH];      // two stages x three planes
  __shared__ __attribute__((aligned(16))) bf16_t Bs[2][3][BN][PITCH];
  const int lb = xcd_chunk(blockIdx.x, gridDim.x);
  const int m0 = (lb / ntiles_n) * BM, n0 = (lb % ntiles_n) * BN;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int lr = t >> 2, lk = (t & 3) * 4;                       // rows lr + 64 j, columns lk .. lk+3 of both tiles
  float4 ra[2][2], rb[2][2];                                     // two register stages: loads run two steps ahead of their use
  int py[2], px[2];                                              // CONV: image row / column of this thread's two A rows
  if (CONV) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + lr + 64 * j, pix = m % (H * W);
      py[j] = pix / W;
      px[j] = pix - py[j] * W;
    }
  }
  auto gload = [&](int s, int k0) {
    int tap = 0, dy = 0, dx = 0, kc = k0 + lk;
    if (CONV) { tap = (k0 + lk) / lda; kc = (k0 + lk) - tap * lda; dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }   // lda = Ci
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = lr + 64 * j, k = k0 + lk;
      if (CONV) {
        const int yy = py[j] + dy, xx = px[j] + dx;
        const bool ok = m0 + r < M && k < K && yy >= 0 && yy < H && xx >= 0 && xx < W;
        ra[s][j] = ok ? *reinterpret_cast<const float4 *>(A + ((int64_t)(m0 + r) + dy * W + dx) * lda + kc) : make_float4(0, 0, 0, 0);
      } else {
        ra[s][j] = (m0 + r < M && k < K) ? *reinterpret_cast<const float4 *>(A + (int64_t)(m0 + r) * lda + k) : make_float4(0, 0, 0, 0);
      }
      rb[s][j] = (n0 + r < N && k < K) ? *reinterpret_cast<const float4 *>(B + (int64_t)(n0 + r) * ldb + k) : make_float4(0, 0, 0, 0);
    }
  };
  // split + store one staged float4 (operand 0 = A, 1 = B; j = which of the thread's two rows) of register stage s into LDS
  // stage buf.  Issued in four pieces BETWEEN the MFMA groups of a step: the matrix pipe works ~128 cycles on a group of
  // four, the ~25 VALU operations of a piece run in that shadow.
  auto lstore1 = [&](int s, int buf, int op, int j) {
    const float4 v = op == 0 ? ra[s][j] : rb[s][j];
    Split4 x;
    if (ABL == 3) { x.hi = make_uint2(pk_bf16(v.x, v.y), pk_bf16(v.z, v.w)); x.mid = x.lo = make_uint2(0, 0); }
    else x = split4(v);
    const int r = lr + 64 * j;
    bf16_t(*P)[BM][PITCH] = op == 0 ? As[buf] : Bs[buf];
    *reinterpret_cast<uint2 *>(&P[0][r][lk]) = x.hi; *reinterpret_cast<uint2 *>(&P[1][r][lk]) = x.mid; *reinterpret_cast<uint2 *>(&P[2][r][lk]) = x.lo;
  };
  auto lstore = [&](int s, int buf) { lstore1(s, buf, 0, 0); lstore1(s, buf, 0, 1); lstore1(s, buf, 1, 0); lstore1(s, buf, 1, 1); };
  // two-level accumulation: `acc` collects FLUSH k-steps (256 contraction elements), then is added to `tot` and cleared — the
  // rounding error of a long contraction (K = 2304 in the 3 x 3 convolution) grows with the square root of the chain length
  constexpr int FLUSH = 16;
  f32x16 acc[2][2], tot[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; tot[i][j][e] = 0.f; }

  const int KT = (K + BK - 1) / BK;
  gload(0, 0);
  if (KT > 1) gload(1, BK);
  lstore(0, 0);
  __syncthreads();
  const int fr = lane & 31, fk = (lane >> 5) * 8;
  // step kt computes from LDS stage kt & 1, writes tile kt + 1 (register stage (kt + 1) & 1, loaded during step kt - 1) to the
  // other LDS stage and issues the loads of tile kt + 2 into the register stage it has just freed
  auto step = [&](int kt, int par) {
    if (kt + 2 < KT) gload(par, (kt + 2) * BK);        // first thing in the step: a whole step of latency cover (see the wide kernel)
    hwbf16x8 a[3][2], b[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[p][i] = frag(&As[par][p][wm + i * 32 + fr][fk]);
        b[p][i] = frag(&Bs[par][p][wn + i * 32 + fr][fk]);
      }
    // one partial product at a time over the four accumulators (consecutive MFMAs independent); smallest terms first
#define TERM(PA, PB)                                                         \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                              \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) mma16k(acc[i][j], a[PA][i], b[PB][j]);
    const bool more = kt + 1 < KT;
    if (ABL != 1 && ABL != 2) {
      TERM(2, 0) if (more) lstore1(par ^ 1, par ^ 1, 0, 0);
      TERM(0, 2) if (more) lstore1(par ^ 1, par ^ 1, 0, 1);
      TERM(1, 1) if (more) lstore1(par ^ 1, par ^ 1, 1, 0);
      TERM(1, 0) if (more) lstore1(par ^ 1, par ^ 1, 1, 1);
      TERM(0, 1)
    } else if (more) lstore(par ^ 1, par ^ 1);
    if (ABL != 1) { TERM(0, 0) }
#undef TERM
    if (ABL == 1) acc[0][0][0] += (float)a[0][0][0] + (float)b[2][1][1] + (float)a[1][1][2] + (float)b[1][0][3] + (float)a[2][0][5] + (float)b[0][0][7];
    if (KT > FLUSH && (kt + 1) % FLUSH == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) { tot[i][j][e] += acc[i][j][e]; acc[i][j][e] = 0.f; }
    }
    __syncthreads();
  };
  for (int kt = 0; kt < KT; kt += 2) {
    step(kt, 0);
    if (kt + 1 < KT) step(kt + 1, 1);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] += tot[i][j][e];
  // C layout of the 32x32 MFMA: col = lane & 31 (B row), row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) (A row); bias loaded before the
  // first store and a branch-free path for interior tiles (see the wide kernel's epilogue: guarded stores serialise on vmcnt(0))
  float bv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn + j * 32 + (lane & 31);
    bv[j] = (bias && col < N) ? bias[col] : 0.f;
  }
  const bool full = m0 + BM <= M && n0 + BN <= N;
  auto store_tile = [&](auto guard) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn + j * 32 + (lane & 31);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          float v = acc[i][j][e] + bv[j];
          if (RELU) v = fmaxf(v, 0.f);
          if ((!decltype(guard)::value || (row < M && col < N)) && (ABL != 4 || v_never(acc[i][j][e]))) C[(int64_t)row * ldc + col] = v;
        }
    }
  };
  if (full) store_tile(std::false_type{});
  else store_tile(std::true_type{});
}

// ---------------------------------------------------------------------------------------------- wide tiles
// The 128 x 128 kernel above is bound by the L2 -> LDS operand stream, not by the matrix pipe: every A element is fetched
// N / 128 times and every B element M / 128 times — 4 M N K (1/BM + 1/BN) bytes, 704 MB for the encoder's 1024 <- 256 Linear at
// M = 43 008, which the kernel moves in ~100 us with the MFMAs switched off (x3_ablate 1), against 54 us of matrix work.
// 256 x 256 tiles halve that stream (352 MB): 8 wavefronts (4 x 2), each 64 x 128 of the tile = 2 x 4 MFMA tiles (128
// accumulator registers), one workgroup per CU, two 120 KB LDS stages of three bf16 planes.  Per 16-wide step a wave issues
// 48 MFMAs (1 536 cycles) against 18 operand fragments; the B fragments are fetched for two column tiles at a time so that
// 12 fragments, not 18, are live next to the accumulators.
// Measured (with the schedule of the step below): 161-169 vs 202-216 us on the 1024 <- 256 Linear, 143 vs 167 on 256 <- 1024, a tie
// at K = 256 with N = 256 / 512.  Ablations of THIS kernel on that shape: no MFMA
// 117 us, no output stores 129, no operand split 148 — operand staging (~42) + split (28) + matrix work (59, the full bf16
// rate) + stores (47, 3.7 TB/s) simply ADD UP: one barrier per 16-wide step keeps the 8 wavefronts in lock-step, so the pipes
// take turns instead of overlapping.  A 128 x 256 variant with two independent workgroups per CU (template parameter WVM = 2,
// `x3_narrow` 3) measures the same (172 vs 169 us) — so it is not the common barrier; moving the global loads to the top of
// the step and a conflict-free LDS layout changed nothing either; the guide's register-staging order (write tile t + 1 right
// after the barrier, re-issue the loads, then compute) is worth 6 % and frees 26 VGPRs.  What is left to try is a deeper software pipeline (fragments of step k + 1 fetched during step k, i.e.
// three LDS stages) — the structure, not any single pipe, is the limit.
constexpr int WBM = 256, WBN = 256;

// ReLU backward of the FFN in the epilogue.  The forward (RELU) launch can leave the sign pattern of its output as a BIT per
// element — one 32-bit word per (workgroup tile, lane, column tile j) holding that lane's 2 x 16 accumulator elements, i.e.
// in this kernel's own C layout (`bits`, 1/32 of the tensor) — and the MASKSUM launch of the transposed product over the
// same [M, N] (same tiling, same layout) consumes it: C = (A B^T) where the forward output was > 0, else 0, and colsum[N] +=
// the column sums of that C (the bias gradient of the Linear in front of the ReLU).  That replaces a separate pass over the
// [M, N] gradient (pd_relu_bwd_colsum: read 2, write 1 such tensor) by 4 words per lane.  (Reading the fp32 activations
// themselves in the epilogue instead was measured: +140 us per launch, a net loss.)
// WVM = wavefronts along M: 4 -> 256 x 256 tile, 512 threads, one workgroup per CU; 2 -> 128 x 256 tile, 256 threads, TWO
// independent workgroups per CU (each SIMD then hosts one wave of each: they are not tied by a common barrier and can sit in
// different phases of their steps).
// BPRE: B is not fp32 but its three bf16 planes, split once per optimizer step by pd_split3_bf16 ([3][N][K], row stride K) — B
// is a weight matrix, re-split here by every one of the M / 256 row tiles otherwise.  Bit-identical to the in-kernel split; measured
// (tools/bench_gemm_x3_wide.py) it does NOT pay: 174.6 vs 178.6 us on 1024 <- 256, 157 vs 151 on 256 <- 1024, plus the split launch
// — the bf16 planes are 6 bytes per element to fetch instead of 4, which costs what the saved vector work gained.  Kept as an
// entry point (off by default in functions/encoder_core.py).
template <bool RELU, int ABL = 0, bool MASKSUM = false, int WVM = 4, bool BPRE = false>   // ABL (tools only): 1 no MFMA, 2 no output stores, 3 no operand split
__global__ __launch_bounds__(128 * WVM, WVM == 4 ? 1 : 2) void gemm_tn_f32x3_wide(const float *__restrict__ A, const float *__restrict__ B,
                                                              const float *__restrict__ bias, float *__restrict__ C, int M, int N,
                                                              int K, int lda, int ldb, int ldc, int ntiles_n,
                                                              uint32_t *__restrict__ bits, float *__restrict__ colsum)
{
  // LDS tile layout: [stage][plane][k half][row][8 bf16] — a 16-wide step is two PANELS of 16-byte row slots.  An MFMA
  // operand (8 consecutive k of row lane % 32, half lane / 32) is then ONE ds_read_b128 and the 32 lanes of a half read 512
  // contiguous bytes (conflict-free, full LDS rate); the 40-byte row pitch of the 128 x 128 kernel needs two 8-byte reads
  // per operand and puts rows r and r + 16 on the same banks.
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];              // As[2][3][2][WBM][8] | Bs[2][3][2][WBN][8]
  constexpr int TBM = 64 * WVM, NTH = 128 * WVM, RPP = NTH / 4, BPASS = WBN / RPP;     // tile rows, threads, rows per load pass, B passes
  auto As = [&](int buf, int pl, int h, int r) -> bf16_t * { return smem + ((((buf * 3 + pl) * 2 + h) * TBM + r) << 3); };
  auto Bs = [&](int buf, int pl, int h, int r) -> bf16_t * { return smem + 12 * TBM * 8 + ((((buf * 3 + pl) * 2 + h) * WBN + r) << 3); };
  const int lb = xcd_chunk(blockIdx.x, gridDim.x);
  const int m0 = (lb / ntiles_n) * TBM, n0 = (lb % ntiles_n) * WBN;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 128;
  const int lr = t >> 2, lk = (t & 3) * 4;                       // rows lr + RPP j, columns lk .. lk+3 of both tiles
  float4 ra[2], rb[BPASS];                                       // ONE register stage (see the step)
  uint4 rbp[3];                                                  // BPRE: one (row, k half) unit of each plane per thread (512 threads x 3 = the tile)
  const bf16_t *Bp = reinterpret_cast<const bf16_t *>(B);
  const int64_t PS = (int64_t)N * ldb;                           // BPRE: plane stride (ldb = K)
  const int br = t >> 1, bh = t & 1;
  auto gload = [&](int k0) {
    const int k = k0 + lk;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = lr + RPP * j;
      ra[j] = (m0 + r < M && k < K) ? *reinterpret_cast<const float4 *>(A + (int64_t)(m0 + r) * lda + k) : make_float4(0, 0, 0, 0);
    }
    if (BPRE) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        rbp[pl] = (n0 + br < N && k0 + bh * 8 < K) ? *reinterpret_cast<const uint4 *>(Bp + pl * PS + (int64_t)(n0 + br) * ldb + k0 + bh * 8)
                                                   : make_uint4(0, 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < BPASS; ++j) {
        const int r = lr + RPP * j;
        rb[j] = (n0 + r < N && k < K) ? *reinterpret_cast<const float4 *>(B + (int64_t)(n0 + r) * ldb + k) : make_float4(0, 0, 0, 0);
      }
    }
  };
  auto lstore1 = [&](int buf, int op, int j) {
    const float4 v4 = op == 0 ? ra[j] : rb[j];
    Split4 x;
    if (ABL == 3) { x.hi = make_uint2(pk_bf16(v4.x, v4.y), pk_bf16(v4.z, v4.w)); x.mid = x.lo = make_uint2(0, 0); }
    else x = split4(v4);
    const int r = lr + RPP * j;
    bf16_t *p0 = (op == 0 ? As(buf, 0, lk >> 3, r) : Bs(buf, 0, lk >> 3, r)) + (lk & 7);
    const int PL = 2 * (op == 0 ? TBM : WBN) * 8;                // plane stride
    *reinterpret_cast<uint2 *>(p0) = x.hi; *reinterpret_cast<uint2 *>(p0 + PL) = x.mid; *reinterpret_cast<uint2 *>(p0 + 2 * PL) = x.lo;
  };
  auto lstore = [&](int buf) {
    lstore1(buf, 0, 0); lstore1(buf, 0, 1);
    if (BPRE) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint4 *>(Bs(buf, pl, bh, br)) = rbp[pl];
    } else {
#pragma unroll
      for (int j = 0; j < BPASS; ++j) lstore1(buf, 1, j);
    }
  };
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int KT = (K + BK - 1) / BK;
  gload(0);
  lstore(0);
  if (KT > 1) gload(BK);
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5;
  auto step = [&](int kt, int par) {
    // the guide's register-staging schedule (T14): right after the barrier, tile kt + 1 (in the registers since the top of step
    // kt - 1) is split and written to the other LDS buffer, the loads of tile kt + 2 are re-issued into the same registers at
    // once (a whole step to land), and only then does the wave turn to tile kt's fragments and MFMAs — no barrier between the
    // staging and the compute, so the waves drift apart inside a step and one's MFMAs cover another's staging
    if (kt + 1 < KT) {
      lstore(par ^ 1);
      if (kt + 2 < KT) gload((kt + 2) * BK);
    }
    hwbf16x8 a[3][2];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int i = 0; i < 2; ++i) a[pl][i] = *reinterpret_cast<const hwbf16x8 *>(As(par, pl, fh, wm + i * 32 + fr));
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {                             // two column tiles at a time
      hwbf16x8 b[3][2];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int j = 0; j < 2; ++j) b[pl][j] = *reinterpret_cast<const hwbf16x8 *>(Bs(par, pl, fh, wn + (jp * 2 + j) * 32 + fr));
      // one partial product at a time over the four accumulators (consecutive MFMAs independent); smallest terms first
#define WTERM(PA, PB)                                                        \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                              \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) { if (ABL == 1) acc[i][jp * 2 + j][0] += (float)a[PA][i][0] * (float)b[PB][j][1]; else mma16k(acc[i][jp * 2 + j], a[PA][i], b[PB][j]); }
      WTERM(2, 0)
      WTERM(0, 2)
      WTERM(1, 1)
      WTERM(1, 0)
      WTERM(0, 1)
      WTERM(0, 0)
#undef WTERM
    }
    __syncthreads();
  };
  for (int kt = 0; kt < KT; kt += 2) {
    step(kt, 0);
    if (kt + 1 < KT) step(kt + 1, 1);
  }
  // C layout of the 32x32 MFMA: col = lane & 31 (B row), row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) (A row).
  // Everything a store depends on is loaded BEFORE the first store, and interior tiles take a branch-free path: with a per-element
  // `if (row < M)` around the store hipcc puts an s_waitcnt vmcnt(0) into every guarded block (it cannot prove the bias / bits load
  // has been waited for on that path), and vmcnt counts stores too — every store then waited for the previous one's
  // acknowledgement, 128 serialised round trips per wave (round 3: ~35 of this kernel's ~185 us at 1024 <- 256).
  float bv[4];
  uint32_t word[4];
  int64_t widx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = n0 + wn + j * 32 + (lane & 31);
    bv[j] = (bias && col < N) ? bias[col] : 0.f;
    widx[j] = (((int64_t)lb * (2 * WVM) + wave) * 64 + lane) * 4 + j;
    word[j] = MASKSUM ? bits[widx[j]] : 0u;
  }
  const bool full = m0 + TBM <= M && n0 + WBN <= N;
  auto store_tile = [&](auto guard) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wn + j * 32 + (lane & 31);
      float csum = 0.f;
      uint32_t w = word[j];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          const bool ok = !decltype(guard)::value || (row < M && col < N);
          float v = acc[i][j][e] + bv[j];
          if (RELU) {
            v = fmaxf(v, 0.f);
            if (ok) w |= (v > 0.f ? 1u : 0u) << (i * 16 + e);
          }
          if (MASKSUM) {
            v = ((w >> (i * 16 + e)) & 1u) ? v : 0.f;
            if (ok) csum += v;
          }
          if (ok && (ABL != 2 || v_never(acc[i][j][e]))) C[(int64_t)row * ldc + col] = v;
        }
      if (RELU && bits && col < N) bits[widx[j]] = w;
      if (MASKSUM) {
        csum += __shfl_xor(csum, 32, 64);                          // the two row halves of the wavefront hold the same column
        if (lane < 32 && csum != 0.f && col < N) unsafeAtomicAdd(colsum + col, csum);
      }
    }
  };
  if (full) store_tile(std::false_type{});
  else store_tile(std::true_type{});
}

// ---------------------------------------------------------------------------------------------- weight gradient
// dW[N,K] += dY[M,N]^T X[M,K] contracts over the ROWS, so an MFMA operand (8 consecutive contraction elements per lane) is
// 8 consecutive rows of one column: the tiles are staged TRANSPOSED, [column][row pair] with the two rows of a pair packed
// in one dword (they are the low / high bf16 of the pair's split).  A lane stages one column (scalar 4-byte loads, 256
// contiguous bytes per wave instruction) so the 32 lanes of a write hit 32 different banks (row pitch 9 dwords, odd).
constexpr int WS = 16, WPD = WS / 2 + 1;                         // rows per stage, dwords per LDS row

__global__ __launch_bounds__(256, 2) void gemm_wgrad_f32x3(const float *__restrict__ dY, const float *__restrict__ X,
                                                            float *__restrict__ dW, float *__restrict__ dB, int M, int N, int K,
                                                            int ldy, int ldx, int ldw, int tiles_k, int tiles, int m_chunk)
{
  __shared__ unsigned Yt[2][3][BN][WPD];
  __shared__ unsigned Xt[2][3][BM][WPD];
  const int tile = blockIdx.x % tiles, split = blockIdx.x / tiles;
  const int n0 = (tile / tiles_k) * BN, k0 = (tile % tiles_k) * BM;
  const int mb = split * m_chunk, me = min(M, mb + m_chunk);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;
  const int col = t & 127, pg = (t >> 7) * 4;                    // this thread: column `col`, row pairs pg .. pg+3 of the stage
  const bool ycol_ok = n0 + col < N, xcol_ok = k0 + col < K;
  float ry[2][8], rx[2][8];                                      // two register stages
  auto gload = [&](int s, int m) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = m + 2 * pg + q;
      ry[s][q] = (r < me && ycol_ok) ? dY[(int64_t)r * ldy + n0 + col] : 0.f;
      rx[s][q] = (r < me && xcol_ok) ? X[(int64_t)r * ldx + k0 + col] : 0.f;
    }
  };
  const bool do_bias = dB != nullptr && k0 == 0;
  float bsum = 0.f;
  auto lstore = [&](int s, int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned h, m_, l;
      split2(ry[s][2 * q], ry[s][2 * q + 1], h, m_, l);
      Yt[buf][0][col][pg + q] = h; Yt[buf][1][col][pg + q] = m_; Yt[buf][2][col][pg + q] = l;
      split2(rx[s][2 * q], rx[s][2 * q + 1], h, m_, l);
      Xt[buf][0][col][pg + q] = h; Xt[buf][1][col][pg + q] = m_; Xt[buf][2][col][pg + q] = l;
      if (do_bias) bsum += ry[s][2 * q] + ry[s][2 * q + 1];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int steps = (me - mb + WS - 1) / WS;
  if (steps > 0) {
    gload(0, mb);
    if (steps > 1) gload(1, mb + WS);
    lstore(0, 0);
  }
  __syncthreads();
  const int fr = lane & 31, fd = (lane >> 5) * 4;                // operand: column fr of a 32-block, row pairs fd .. fd+3 (8 rows)
  auto frag4 = [&](const unsigned *p) {
    union { unsigned d[4]; hwbf16x8 v; } u;
    u.d[0] = p[0]; u.d[1] = p[1]; u.d[2] = p[2]; u.d[3] = p[3];
    return u.v;
  };
  auto step = [&](int st, int par) {
    hwbf16x8 a[3][2], b[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[p][i] = frag4(&Yt[par][p][wn + i * 32 + fr][fd]);
        b[p][i] = frag4(&Xt[par][p][wk + i * 32 + fr][fd]);
      }
#define TERM(PA, PB)                                                         \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                              \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) mma16k(acc[i][j], a[PA][i], b[PB][j]);
    TERM(2, 0) TERM(0, 2) TERM(1, 1) TERM(1, 0) TERM(0, 1) TERM(0, 0)
#undef TERM
    if (st + 1 < steps) lstore(par ^ 1, par ^ 1);
    if (st + 2 < steps) gload(par, mb + (st + 2) * WS);
    __syncthreads();
  };
  for (int st = 0; st < steps; st += 2) {
    step(st, 0);
    if (st + 1 < steps) step(st + 1, 1);
  }
  if (do_bias && ycol_ok) unsafeAtomicAdd(dB + n0 + col, bsum);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = k0 + wk + j * 32 + (lane & 31);
    if (c >= K) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = n0 + wn + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < N) unsafeAtomicAdd(dW + (int64_t)row * ldw + c, acc[i][j][e]);
      }
  }
}

// ---------------------------------------------------------------------------------------------- weight gradient, transpose-read form
// The same product with the tiles staged the way they lie in memory — [contraction row][column], 16-byte global loads, one
// 8-byte LDS write per plane — and TRANSPOSED ON THE WAY OUT of LDS by ds_read_b64_tr_b16 (gfx950): within a group of 16
// lanes, lane i receives element i % 4 of the 8-byte segments addressed by lanes i / 4 + 4 e (e = 0..3 -> the 4 result
// elements; mapping read off the hardware with tools/probes/tr_read_probe.hip).  Lane s of a group therefore points at
// row s / 4, columns 4 (s % 4) .. +3 of a [4 rows][16 columns] block and lane i gets column i of the 4 rows: two such reads are the
// 8 consecutive contraction elements of one column that v_mfma_f32_32x32x16_bf16 wants from lane (column, k half).
// Row pitch 160 bf16 = 320 bytes: the 4 rows of a read start 16 banks apart, its 32 lanes cover 64 distinct banks.
constexpr int TWS = 16, TP = 160;
typedef short v4s16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ hwbf16x8 frag_tr(const bf16_t *p)      // p: this lane's segment of rows 0..3; rows 4..7 are 4 * TP further
{
  typedef __attribute__((address_space(3))) v4s16 *lp;
  union { v4s16 h[2]; hwbf16x8 v; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p));
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p + 4 * TP));
  return u.v;
}

// CONV: X is an NHWC image [*, H, W, Ci] (ldx = Ci), K = 9 Ci ordered (tap, channel) like the channels-last filter gradient
// [Co][3][3][Ci]; Ci % 128 == 0, so a tile's 128 columns lie inside one tap and its X rows are the pixels m + dy W + dx (zeros
// outside the image): the weight gradient of a 3 x 3, stride 1, pad 1 convolution, nothing unfolded.
// H2: the fp16 two-plane form (f16x2.h; three products per term instead of six).  The contraction runs over the rows, so the two
// operands are scaled per WORKGROUP: by powers of two from the largest row maximum (y_amax / x_amax, the absolute row maxima of dY
// / X as their producers emit them; NULL = O(1) operand) among the rows [mb, me) this workgroup walks, undone on its partial tile.
// Rows far below their slab's maximum lose relative, not absolute, accuracy — the sum is dominated by the large rows.
template <int ABL, bool CONV, bool H2 = false>
__device__ __forceinline__ void wgrad_tr_body(const float *__restrict__ dY, const float *__restrict__ X, float *__restrict__ dW,
                                              float *__restrict__ dB, float *__restrict__ ws, int M, int N, int K, int ldy, int ldx,
                                              int ldw, int tiles_k, int tiles, int m_chunk, int H, int W, int bid, int64_t ws_tile0,
                                              const float *__restrict__ y_amax = nullptr, const float *__restrict__ x_amax = nullptr)
{
  constexpr int NP = H2 ? 2 : 3;
  __shared__ __attribute__((aligned(16))) bf16_t S[2][2][NP][TWS][TP];     // stage, operand (dY, X), plane, row, column
  const int tile = bid % tiles, split = bid / tiles;
  const int n0 = (tile / tiles_k) * BN, k0 = (tile % tiles_k) * BM;
  const int mb = split * m_chunk, me = min(M, mb + m_chunk);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;
  const int sr = t >> 5, sc = (t & 31) * 4;                      // staging: rows sr, sr + 8; columns sc .. sc+3
  const bool ycol_ok = n0 + sc < N, xcol_ok = k0 + sc < K;
  float sy = 1.f, sx = 1.f, inv_yx = 1.f;
  if (H2 && (y_amax || x_amax)) {
    float my = 0.f, mx = 0.f;
    if (y_amax) for (int r = mb + t; r < me; r += 256) my = fmaxf(my, y_amax[r]);
    // CONV: the X rows of this slab are the pixels r + dy W + dx — the range widened by one image row + 1 bounds them
    const int xlo = CONV ? max(0, mb - W - 1) : mb, xhi = CONV ? min(M, me + W + 1) : me;
    if (x_amax) for (int r = xlo + t; r < xhi; r += 256) mx = fmaxf(mx, x_amax[r]);
#pragma unroll
    for (int o = 32; o; o >>= 1) { my = fmaxf(my, __shfl_xor(my, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
    float *red = reinterpret_cast<float *>(&S[0][0][0][0][0]);
    if (lane == 0) { red[wave] = my; red[4 + wave] = mx; }
    __syncthreads();
    my = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    mx = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    __syncthreads();
    float iy = 1.f, ix = 1.f;
    if (y_amax) pdh2::row_scale(my, sy, iy);
    if (x_amax) pdh2::row_scale(mx, sx, ix);
    inv_yx = iy * ix;
  }
  float4 ry[2][2], rx[2][2];                                     // two register stages
  int cpy[2] = {0, 0}, cpx[2] = {0, 0}, tdy = 0, tdx = 0, xoff = k0 + sc;   // CONV: image coordinates of the rows the NEXT gload stages
  if (CONV) {
    const int tap = k0 / ldx;
    tdy = tap / 3 - 1; tdx = tap - (tap / 3) * 3 - 1;
    xoff = k0 - tap * ldx + sc;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pix = (mb + sr + 8 * j) % (H * W);
      cpy[j] = pix / W; cpx[j] = pix - cpy[j] * W;
    }
  }
  auto gload = [&](int s, int m) {                               // called with m = mb, mb + 16, mb + 32, ... in this order
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = m + sr + 8 * j;
      ry[s][j] = (r < me && ycol_ok) ? *reinterpret_cast<const float4 *>(dY + (int64_t)r * ldy + n0 + sc) : make_float4(0, 0, 0, 0);
      if (CONV) {
        const int yy = cpy[j] + tdy, xx = cpx[j] + tdx;
        const bool ok = r < me && xcol_ok && yy >= 0 && yy < H && xx >= 0 && xx < W;
        rx[s][j] = ok ? *reinterpret_cast<const float4 *>(X + ((int64_t)r + tdy * W + tdx) * ldx + xoff) : make_float4(0, 0, 0, 0);
        cpx[j] += TWS;
        while (cpx[j] >= W) { cpx[j] -= W; if (++cpy[j] == H) cpy[j] = 0; }
      } else {
        rx[s][j] = (r < me && xcol_ok) ? *reinterpret_cast<const float4 *>(X + (int64_t)r * ldx + k0 + sc) : make_float4(0, 0, 0, 0);
      }
    }
  };
  const bool do_bias = dB != nullptr && k0 == 0;
  float4 bsum = make_float4(0, 0, 0, 0);
  auto lstore = [&](int s, int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = sr + 8 * j;
      if constexpr (H2) {
        const pdh2::SplitH y = pdh2::split4h(ry[s][j], sy), x = pdh2::split4h(rx[s][j], sx);
        *reinterpret_cast<uint2 *>(&S[buf][0][0][r][sc]) = y.hi; *reinterpret_cast<uint2 *>(&S[buf][0][1][r][sc]) = y.lo;
        *reinterpret_cast<uint2 *>(&S[buf][1][0][r][sc]) = x.hi; *reinterpret_cast<uint2 *>(&S[buf][1][1][r][sc]) = x.lo;
      } else {
        const Split4 y = split4(ry[s][j]), x = split4(rx[s][j]);
        *reinterpret_cast<uint2 *>(&S[buf][0][0][r][sc]) = y.hi; *reinterpret_cast<uint2 *>(&S[buf][0][1][r][sc]) = y.mid;
        *reinterpret_cast<uint2 *>(&S[buf][0][NP - 1][r][sc]) = y.lo;
        *reinterpret_cast<uint2 *>(&S[buf][1][0][r][sc]) = x.hi; *reinterpret_cast<uint2 *>(&S[buf][1][1][r][sc]) = x.mid;
        *reinterpret_cast<uint2 *>(&S[buf][1][NP - 1][r][sc]) = x.lo;
      }
      if (do_bias) { bsum.x += ry[s][j].x; bsum.y += ry[s][j].y; bsum.z += ry[s][j].z; bsum.w += ry[s][j].w; }
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int steps = (me - mb + TWS - 1) / TWS;
  if (steps > 0) {
    gload(0, mb);
    if (steps > 1) gload(1, mb + TWS);
    lstore(0, 0);
  }
  __syncthreads();
  const int grp = lane >> 4, sl = lane & 15;
  const int frow = 8 * (grp >> 1) + (sl >> 2), fcol = 16 * (grp & 1) + 4 * (sl & 3);   // this lane's segment inside a 32-column block
  auto step = [&](int st, int par) {
    if (st + 2 < steps) gload(par, mb + (st + 2) * TWS);
    hwbf16x8 a[NP][2], b[NP][2];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[p][i] = frag_tr(&S[par][0][p][frow][wn + i * 32 + fcol]);
        b[p][i] = frag_tr(&S[par][1][p][frow][wk + i * 32 + fcol]);
      }
#define TERM(PA, PB)                                                         \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                              \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                        \
        if constexpr (H2) pdh2::mmah(acc[i][j], __builtin_bit_cast(pdh2::h16x8, a[PA][i]), __builtin_bit_cast(pdh2::h16x8, b[PB][j])); \
        else mma16k(acc[i][j], a[PA][i], b[PB][j]);                          \
      }
    if constexpr (H2) { TERM(1, 0) TERM(0, 1) TERM(0, 0) }
    else if (ABL != 2) { TERM(NP - 1, 0) TERM(0, NP - 1) TERM(1, 1) TERM(1, 0) TERM(0, 1) TERM(0, 0) }
    else { _Pragma("unroll") for (int p = 0; p < NP; ++p) _Pragma("unroll") for (int i = 0; i < 2; ++i) { acc[i][0][p] += (float)a[p][i][0]; acc[i][1][p] += (float)b[p][i][0]; } }
#undef TERM
    if (st + 1 < steps) lstore(par ^ 1, par ^ 1);
    __syncthreads();
  };
  for (int st = 0; st < steps; st += 2) {
    step(st, 0);
    if (st + 1 < steps) step(st + 1, 1);
  }
  if (do_bias) {
    // column sums: the 8 staging rows of a column are summed through LDS first — ONE atomic per column and workgroup (8 x splits
    // same-address atomics per column serialise: 1 024 of them cost ~100 us at 256 x 256)
    float *red = reinterpret_cast<float *>(&S[0][0][0][0][0]);   // the last step ended with a barrier: the stages are free
    *reinterpret_cast<float4 *>(red + sr * BN + sc) = bsum;
    __syncthreads();
    if (t < BN && n0 + t < N) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) v += red[r * BN + t];
      unsafeAtomicAdd(dB + n0 + t, v);
    }
  }
  if (ws) {
    // partial tile in REGISTER order (element (ij, e) of thread t at ((ij * 16 + e) * 256 + t): 1 KB per store instruction);
    // wgrad_tr_reduce sums the splits and owns the read-modify-write of dW — 8.4 M fp32 atomics per launch cost 30-110 us
    float *w = ws + (ws_tile0 + (int64_t)split * tiles + tile) * (BN * BM) + t;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) w[((i * 2 + j) * 16 + e) * 256] = H2 ? acc[i][j][e] * inv_yx : acc[i][j][e];
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = k0 + wk + j * 32 + (lane & 31);
    if (c >= K) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = n0 + wn + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < N) unsafeAtomicAdd(dW + (int64_t)row * ldw + c, H2 ? acc[i][j][e] * inv_yx : acc[i][j][e]);
      }
  }
}

template <int ABL, bool CONV, bool H2 = false>
__global__ __launch_bounds__(256, 2) void gemm_wgrad_f32x3_tr(const float *__restrict__ dY, const float *__restrict__ X,
                                                               float *__restrict__ dW, float *__restrict__ dB, float *__restrict__ ws,
                                                               int M, int N, int K, int ldy, int ldx, int ldw, int tiles_k, int tiles,
                                                               int m_chunk, int H, int W, const float *__restrict__ y_amax,
                                                               const float *__restrict__ x_amax)
{
  // XCD-aware order (block i runs on XCD i % 8): the output tiles of one slab of rows are consecutive LOGICAL blocks, i.e. land on
  // the same XCD at about the same time and share the slab's dY / X columns in its L2 instead of fetching them into all eight
  const int lb = g_wgrad_xcd ? xcd_chunk((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  wgrad_tr_body<ABL, CONV, H2>(dY, X, dW, dB, ws, M, N, K, ldy, ldx, ldw, tiles_k, tiles, m_chunk, H, W, lb, 0, y_amax, x_amax);
}

// ---------------------------------------------------------------------------------------------- weight gradient, 256 x 256 tiles (f16x2)
// The 128 x 128 kernel above re-reads every slab of rows once per output tile that needs its columns: 2.8 x the algorithmic bytes
// at the encoder's five weight gradients (rocprofv3 FETCH_SIZE / WRITE_SIZE: 1.97 GB per layer against 0.71 GB, profiles/r03_gemm_pmc.json) —
// 11.8 GB per step in 1.6 ms, i.e. the kernel runs at the memory system's limit ON ITS OWN RE-READS.  256 x 256 tiles halve them:
// 8 wavefronts (4 x 2), each 64 x 128 of the tile (2 x 4 MFMA tiles, 128 accumulator registers), one workgroup per CU, 16-row stages
// of two fp16 planes per operand (74 KB), the same transpose reads on a 576-byte row pitch (= 16 banks mod 64 like the 320-byte one).
constexpr int WTD = 256, WTP = 288, WNT = 512;
__device__ __forceinline__ hwbf16x8 frag_trw(const bf16_t *p)      // frag_tr on the wide tile's row pitch
{
  typedef __attribute__((address_space(3))) v4s16 *lp;
  union { v4s16 h[2]; hwbf16x8 v; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p));
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p + 4 * WTP));
  return u.v;
}

template <bool CONV>
__device__ __forceinline__ void wgrad_h2w_body(const float *__restrict__ dY, const float *__restrict__ X, float *__restrict__ dW,
                                               float *__restrict__ dB, float *__restrict__ ws, int M, int N, int K, int ldy, int ldx,
                                               int ldw, int tiles_k, int tiles, int m_chunk, int H, int W, int bid, int64_t ws_tile0,
                                               const float *__restrict__ y_amax, const float *__restrict__ x_amax)
{
  extern __shared__ __attribute__((aligned(16))) bf16_t Sw[];                // [stage 2][operand 2][plane 2][TWS][WTP]
  auto S = [&](int buf, int op, int pl, int r) -> bf16_t * { return Sw + ((((buf * 2 + op) * 2 + pl) * TWS + r) * WTP); };
  const int tile = bid % tiles, split = bid / tiles;
  const int n0 = (tile / tiles_k) * WTD, k0 = (tile % tiles_k) * WTD;
  const int mb = split * m_chunk, me = min(M, mb + m_chunk);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wn = (wave >> 1) * 64, wk = (wave & 1) * 128;
  const int sr = t >> 6, sc = (t & 63) * 4;                      // staging: rows sr, sr + 8; columns sc .. sc+3
  const bool ycol_ok = n0 + sc < N, xcol_ok = k0 + sc < K;
  float sy = 1.f, sx = 1.f, inv_yx = 1.f;
  if (y_amax || x_amax) {                                        // slab-wise power-of-two scales (see wgrad_tr_body)
    float my = 0.f, mx = 0.f;
    if (y_amax) for (int r = mb + t; r < me; r += WNT) my = fmaxf(my, y_amax[r]);
    const int xlo = CONV ? max(0, mb - W - 1) : mb, xhi = CONV ? min(M, me + W + 1) : me;
    if (x_amax) for (int r = xlo + t; r < xhi; r += WNT) mx = fmaxf(mx, x_amax[r]);
#pragma unroll
    for (int o = 32; o; o >>= 1) { my = fmaxf(my, __shfl_xor(my, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
    float *red = reinterpret_cast<float *>(Sw);
    if (lane == 0) { red[wave] = my; red[8 + wave] = mx; }
    __syncthreads();
    my = mx = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { my = fmaxf(my, red[i]); mx = fmaxf(mx, red[8 + i]); }
    __syncthreads();
    float iy = 1.f, ix = 1.f;
    if (y_amax) pdh2::row_scale(my, sy, iy);
    if (x_amax) pdh2::row_scale(mx, sx, ix);
    inv_yx = iy * ix;
  }
  float4 ry[2][2], rx[2][2];                                     // two register stages
  int cpy[2] = {0, 0}, cpx[2] = {0, 0}, tdy = 0, tdx = 0, xoff = k0 + sc;
  if (CONV) {
    const int tap = k0 / ldx;
    tdy = tap / 3 - 1; tdx = tap - (tap / 3) * 3 - 1;
    xoff = k0 - tap * ldx + sc;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pix = (mb + sr + 8 * j) % (H * W);
      cpy[j] = pix / W; cpx[j] = pix - cpy[j] * W;
    }
  }
  auto gload = [&](int s, int m) {                               // called with m = mb, mb + 16, mb + 32, ... in this order
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = m + sr + 8 * j;
      ry[s][j] = (r < me && ycol_ok) ? *reinterpret_cast<const float4 *>(dY + (int64_t)r * ldy + n0 + sc) : make_float4(0, 0, 0, 0);
      if (CONV) {
        const int yy = cpy[j] + tdy, xx = cpx[j] + tdx;
        const bool ok = r < me && xcol_ok && yy >= 0 && yy < H && xx >= 0 && xx < W;
        rx[s][j] = ok ? *reinterpret_cast<const float4 *>(X + ((int64_t)r + tdy * W + tdx) * ldx + xoff) : make_float4(0, 0, 0, 0);
        cpx[j] += TWS;
        while (cpx[j] >= W) { cpx[j] -= W; if (++cpy[j] == H) cpy[j] = 0; }
      } else {
        rx[s][j] = (r < me && xcol_ok) ? *reinterpret_cast<const float4 *>(X + (int64_t)r * ldx + k0 + sc) : make_float4(0, 0, 0, 0);
      }
    }
  };
  const bool do_bias = dB != nullptr && k0 == 0;
  float4 bsum = make_float4(0, 0, 0, 0);
  auto lstore = [&](int s, int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = sr + 8 * j;
      const pdh2::SplitH y = pdh2::split4h(ry[s][j], sy), x = pdh2::split4h(rx[s][j], sx);
      *reinterpret_cast<uint2 *>(S(buf, 0, 0, r) + sc) = y.hi; *reinterpret_cast<uint2 *>(S(buf, 0, 1, r) + sc) = y.lo;
      *reinterpret_cast<uint2 *>(S(buf, 1, 0, r) + sc) = x.hi; *reinterpret_cast<uint2 *>(S(buf, 1, 1, r) + sc) = x.lo;
      if (do_bias) { bsum.x += ry[s][j].x; bsum.y += ry[s][j].y; bsum.z += ry[s][j].z; bsum.w += ry[s][j].w; }
    }
  };
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int steps = (me - mb + TWS - 1) / TWS;
  if (steps > 0) {
    gload(0, mb);
    if (steps > 1) gload(1, mb + TWS);
    lstore(0, 0);
  }
  __syncthreads();
  const int grp = lane >> 4, sl = lane & 15;
  const int frow = 8 * (grp >> 1) + (sl >> 2), fcol = 16 * (grp & 1) + 4 * (sl & 3);
  auto step = [&](int st, int par) {
    if (st + 2 < steps) gload(par, mb + (st + 2) * TWS);
    hwbf16x8 a[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) a[p][i] = frag_trw(S(par, 0, p, frow) + wn + i * 32 + fcol);
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {                             // two column tiles at a time (8, not 16, X fragments live)
      hwbf16x8 b[2][2];
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int j = 0; j < 2; ++j) b[p][j] = frag_trw(S(par, 1, p, frow) + wk + (jp * 2 + j) * 32 + fcol);
#define WTERM(PA, PB)                                                        \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                              \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                          \
        pdh2::mmah(acc[i][jp * 2 + j], __builtin_bit_cast(pdh2::h16x8, a[PA][i]), __builtin_bit_cast(pdh2::h16x8, b[PB][j]));
      WTERM(1, 0) WTERM(0, 1) WTERM(0, 0)
#undef WTERM
    }
    if (st + 1 < steps) lstore(par ^ 1, par ^ 1);
    __syncthreads();
  };
  int st0 = 0;
  // Interior tiles (all 256 columns of both operands inside, whole 16-row stages): the same step as ONE basic block — no bounds
  // branches around the loads, the convolution's halo test a select on a load from the centre pixel's (always valid) address — with
  // the split, the LDS stores and the next loads laid between the matrix instructions (gemm_f16x2.hip's interleaved step).
  // The last two steps run through the guarded form.
  if (g_wgrad_fast && n0 + WTD <= N && k0 + WTD <= K && ((me - mb) % TWS) == 0 && steps >= 4 && (!CONV || W >= TWS)) {
    const float *py0 = dY + (int64_t)(mb + sr) * ldy + n0 + sc, *px0 = X + (int64_t)(mb + sr) * ldx + (CONV ? xoff : k0 + sc);
    auto fstep = [&](int st, int par) {
      const int64_t m = (int64_t)(st + 2) * TWS;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        ry[par][j] = *reinterpret_cast<const float4 *>(py0 + (m + 8 * j) * ldy);
        if (CONV) {
          const int yy = cpy[j] + tdy, xx = cpx[j] + tdx;
          const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
          const float4 v = *reinterpret_cast<const float4 *>(px0 + (m + 8 * j + (ok ? tdy * W + tdx : 0)) * ldx);
          rx[par][j] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
          cpx[j] += TWS;
          const bool wrap = cpx[j] >= W;
          cpx[j] -= wrap ? W : 0;
          cpy[j] = wrap ? (cpy[j] + 1 == H ? 0 : cpy[j] + 1) : cpy[j];
        } else {
          rx[par][j] = *reinterpret_cast<const float4 *>(px0 + (m + 8 * j) * ldx);
        }
      }
      hwbf16x8 a[2][2], b[2][4];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) a[p][i] = frag_trw(S(par, 0, p, frow) + wn + i * 32 + fcol);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[p][j] = frag_trw(S(par, 1, p, frow) + wk + j * 32 + fcol);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = sr + 8 * j;
        const pdh2::SplitH y = pdh2::split4h(ry[par ^ 1][j], sy), x = pdh2::split4h(rx[par ^ 1][j], sx);
        *reinterpret_cast<uint2 *>(S(par ^ 1, 0, 0, r) + sc) = y.hi; *reinterpret_cast<uint2 *>(S(par ^ 1, 0, 1, r) + sc) = y.lo;
        *reinterpret_cast<uint2 *>(S(par ^ 1, 1, 0, r) + sc) = x.hi; *reinterpret_cast<uint2 *>(S(par ^ 1, 1, 1, r) + sc) = x.lo;
        bsum.x += ry[par ^ 1][j].x; bsum.y += ry[par ^ 1][j].y; bsum.z += ry[par ^ 1][j].z; bsum.w += ry[par ^ 1][j].w;   // read only when do_bias
      }
#define WFTERM(PA, PB)                                                       \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                              \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                          \
        pdh2::mmah(acc[i][j], __builtin_bit_cast(pdh2::h16x8, a[PA][i]), __builtin_bit_cast(pdh2::h16x8, b[PB][j]));
      WFTERM(1, 0) WFTERM(0, 1) WFTERM(0, 0)
#undef WFTERM
      __builtin_amdgcn_sched_group_barrier(0x100, 24, 0);
#pragma unroll
      for (int g = 0; g < 24; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        if (g % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        if (g % 6 == 5) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __syncthreads();
    };
    for (; st0 + 3 < steps; st0 += 2) {
      fstep(st0, 0);
      fstep(st0 + 1, 1);
    }
  }
  for (int st = st0; st < steps; st += 2) {
    step(st, 0);
    if (st + 1 < steps) step(st + 1, 1);
  }
  if (do_bias) {
    float *red = reinterpret_cast<float *>(Sw);                  // the last step ended with a barrier: the stages are free
    *reinterpret_cast<float4 *>(red + sr * WTD + sc) = bsum;
    __syncthreads();
    if (t < WTD && n0 + t < N) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) v += red[r * WTD + t];
      unsafeAtomicAdd(dB + n0 + t, v);
    }
  }
  if (ws) {
    // partial tile in REGISTER order (element (ij, e) of thread t at ((ij * 16 + e) * 512 + t), ij = 4 i + j)
    float *w = ws + (ws_tile0 + (int64_t)split * tiles + tile) * (WTD * WTD) + t;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) w[((i * 4 + j) * 16 + e) * WNT] = acc[i][j][e] * inv_yx;
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = k0 + wk + j * 32 + (lane & 31);
    if (c >= K) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = n0 + wn + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < N) unsafeAtomicAdd(dW + (int64_t)row * ldw + c, acc[i][j][e] * inv_yx);
      }
  }
}

template <bool CONV>
__global__ __launch_bounds__(512, 1) void gemm_wgrad_f16x2_wide(const float *__restrict__ dY, const float *__restrict__ X, float *__restrict__ dW,
                                                                float *__restrict__ dB, float *__restrict__ ws, int M, int N, int K, int ldy, int ldx,
                                                                int ldw, int tiles_k, int tiles, int m_chunk, int H, int W,
                                                                const float *__restrict__ y_amax, const float *__restrict__ x_amax)
{
  wgrad_h2w_body<CONV>(dY, X, dW, dB, ws, M, N, K, ldy, ldx, ldw, tiles_k, tiles, m_chunk, H, W, xcd_chunk((int)blockIdx.x, (int)gridDim.x), 0,
                       y_amax, x_amax);
}

// element (q = ij * 16 + e, thread t) of a wide partial tile -> (output row, column) of the tile
__device__ __forceinline__ void h2w_coord(int q, int t, int &row, int &col)
{
  const int lane = t & 63, wave = t >> 6, ij = q >> 4, e = q & 15, i = ij >> 2, j = ij & 3;
  row = (wave >> 1) * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
  col = (wave & 1) * 128 + j * 32 + (lane & 31);
}

// dW tile += sum over splits of the partial tiles (128 blocks of 512 threads per tile, one (ij, e) each; blockIdx.y takes every
// gridDim.y-th split and finishes with an atomic, gridDim.y == 1 owns its element)
__global__ __launch_bounds__(512) void wgrad_h2w_reduce(const float *__restrict__ ws, float *__restrict__ dW, int N, int K, int ldw, int tiles_k,
                                                        int tiles, int splits)
{
  const int tile = blockIdx.x >> 7, q = blockIdx.x & 127, t = threadIdx.x;
  const int n0 = (tile / tiles_k) * WTD, k0 = (tile % tiles_k) * WTD;
  const float *p = ws + (int64_t)tile * (WTD * WTD) + q * WNT + t;
  const int64_t stride = (int64_t)tiles * (WTD * WTD);
  float s0 = 0.f, s1 = 0.f;
  int sp = blockIdx.y;
  const int G = gridDim.y;
  for (; sp + 7 * G < splits; sp += 8 * G) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(sp + u * G) * stride];
#pragma unroll
    for (int u = 0; u < 8; u += 2) { s0 += v[u]; s1 += v[u + 1]; }
  }
  for (; sp < splits; sp += G) s0 += p[(int64_t)sp * stride];
  int row, col;
  h2w_coord(q, t, row, col);
  if (k0 + col < K && n0 + row < N) {
    float *d = dW + (int64_t)(n0 + row) * ldw + k0 + col;
    if (G == 1) *d += s0 + s1; else unsafeAtomicAdd(d, s0 + s1);
  }
}

// Several weight gradients in ONE launch (pd_gemm_wgrad_f32x3_grouped): a table of problems in device memory, workgroup b belongs
// to the problem whose [block0, block0 + tiles * splits) contains b.  The encoder's 30 weight gradients per step are independent of
// everything until the optimizer runs, so they are queued during its backward pass and run here together: every (tile, split)
// workgroup of the launch walks the same number of rows, the machine is filled in whole rounds once instead of 30 times, and
// the partial-tile traffic falls from 512 tiles per GEMM to ~17 splits per output tile.
struct WgradX3Problem {
  const float *dY, *X;
  float *dW, *dB;
  int M, N, K, ldy, ldx, ldw, tiles_k, tiles, m_chunk, splits, block0, pad;
  int64_t ws_tile0;
  const float *y_amax, *x_amax;
};

template <bool H2>
__global__ __launch_bounds__(256, 2) void gemm_wgrad_f32x3_tr_grouped(const WgradX3Problem *__restrict__ tab, int count, float *__restrict__ ws)
{
  const int lb = g_wgrad_xcd ? xcd_chunk((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;     // see gemm_wgrad_f32x3_tr
  int p = 0;
  while (p + 1 < count && lb >= tab[p + 1].block0) ++p;                   // <= a few dozen problems: a scalar scan
  const WgradX3Problem q = tab[p];
  wgrad_tr_body<0, false, H2>(pd_as_global(q.dY), pd_as_global(q.X), pd_as_global(q.dW), pd_as_global(q.dB), ws, q.M, q.N, q.K, q.ldy, q.ldx, q.ldw, q.tiles_k, q.tiles, q.m_chunk, 0, 0,
                              lb - q.block0, q.ws_tile0, pd_as_global(q.y_amax), pd_as_global(q.x_amax));   // (pd_common.h: table pointers would be FLAT)
}

// grouped form of wgrad_tr_reduce: blockIdx.x = 64 x (global output tile index); the problem is found from its first tile
__global__ __launch_bounds__(256) void wgrad_tr_reduce_grouped(const WgradX3Problem *__restrict__ tab, int count, const float *__restrict__ ws)
{
  const int gt = blockIdx.x >> 6, q = blockIdx.x & 63;
  int p = 0, t0 = 0;
  while (p + 1 < count && gt >= t0 + tab[p].tiles) { t0 += tab[p].tiles; ++p; }
  const WgradX3Problem pr = tab[p];
  const int tile = gt - t0;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ij = q >> 4, e = q & 15, i = ij >> 1, j = ij & 1;
  const int n0 = (tile / pr.tiles_k) * BN, k0 = (tile % pr.tiles_k) * BM;
  const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;
  const float *src = ws + (pr.ws_tile0 + tile) * (BN * BM) + q * 256 + t;
  const int64_t stride = (int64_t)pr.tiles * (BN * BM);
  float s0 = 0.f, s1 = 0.f;
  int sp = 0;
  for (; sp + 7 < pr.splits; sp += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(sp + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; u += 2) { s0 += v[u]; s1 += v[u + 1]; }
  }
  for (; sp < pr.splits; ++sp) s0 += src[(int64_t)sp * stride];
  const int c = k0 + wk + j * 32 + (lane & 31);
  const int row = n0 + wn + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
  if (c < pr.K && row < pr.N) pd_as_global(pr.dW)[(int64_t)row * pr.ldw + c] += s0 + s1;      // this block owns the element: plain read-modify-write
}

__global__ __launch_bounds__(512, 1) void gemm_wgrad_f16x2_wide_grouped(const WgradX3Problem *__restrict__ tab, int count, float *__restrict__ ws)
{
  const int lb = xcd_chunk((int)blockIdx.x, (int)gridDim.x);
  int p = 0;
  while (p + 1 < count && lb >= tab[p + 1].block0) ++p;
  const WgradX3Problem q = tab[p];
  wgrad_h2w_body<false>(pd_as_global(q.dY), pd_as_global(q.X), pd_as_global(q.dW), pd_as_global(q.dB), ws, q.M, q.N, q.K, q.ldy, q.ldx, q.ldw, q.tiles_k, q.tiles, q.m_chunk, 0, 0, lb - q.block0, q.ws_tile0,
                        pd_as_global(q.y_amax), pd_as_global(q.x_amax));
}

__global__ __launch_bounds__(512) void wgrad_h2w_reduce_grouped(const WgradX3Problem *__restrict__ tab, int count, const float *__restrict__ ws)
{
  const int gt = blockIdx.x >> 7, q = blockIdx.x & 127, t = threadIdx.x;
  int p = 0, t0 = 0;
  while (p + 1 < count && gt >= t0 + tab[p].tiles) { t0 += tab[p].tiles; ++p; }
  const WgradX3Problem pr = tab[p];
  const int tile = gt - t0;
  const int n0 = (tile / pr.tiles_k) * WTD, k0 = (tile % pr.tiles_k) * WTD;
  const float *src = ws + (pr.ws_tile0 + tile) * (WTD * WTD) + q * WNT + t;
  const int64_t stride = (int64_t)pr.tiles * (WTD * WTD);
  float s0 = 0.f, s1 = 0.f;
  int sp = 0;
  for (; sp + 7 < pr.splits; sp += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(sp + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; u += 2) { s0 += v[u]; s1 += v[u + 1]; }
  }
  for (; sp < pr.splits; ++sp) s0 += src[(int64_t)sp * stride];
  int row, col;
  h2w_coord(q, t, row, col);
  if (k0 + col < pr.K && n0 + row < pr.N) pd_as_global(pr.dW)[(int64_t)(n0 + row) * pr.ldw + k0 + col] += s0 + s1;
}

// dW tile += sum over splits of the partial tiles gemm_wgrad_f32x3_tr left in the workspace (same register-order indexing)
__global__ __launch_bounds__(256) void wgrad_tr_reduce(const float *__restrict__ ws, float *__restrict__ dW, int N, int K, int ldw, int tiles_k,
                                                       int tiles, int splits)
{
  const int tile = blockIdx.x >> 6, q = blockIdx.x & 63;         // 64 blocks per tile, one (ij, e) each
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ij = q >> 4, e = q & 15, i = ij >> 1, j = ij & 1;
  const int n0 = (tile / tiles_k) * BN, k0 = (tile % tiles_k) * BM;
  const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;
  // blockIdx.y takes every gridDim.y-th split: 8 independent loads in flight per thread, a few atomics per element at the end
  const float *p = ws + (int64_t)tile * (BN * BM) + q * 256 + t;
  const int64_t stride = (int64_t)tiles * (BN * BM);
  float s0 = 0.f, s1 = 0.f;
  int sp = blockIdx.y;
  const int G = gridDim.y;
  for (; sp + 7 * G < splits; sp += 8 * G) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(sp + u * G) * stride];
#pragma unroll
    for (int u = 0; u < 8; u += 2) { s0 += v[u]; s1 += v[u + 1]; }
  }
  for (; sp < splits; sp += G) s0 += p[(int64_t)sp * stride];
  const int c = k0 + wk + j * 32 + (lane & 31);
  const int row = n0 + wn + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
  if (c < K && row < N) unsafeAtomicAdd(dW + (int64_t)row * ldw + c, s0 + s1);
}
}  // namespace

extern "C" void pd_dbg_set_wgrad_xcd(int v) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wgrad_xcd), &v, sizeof(int)); }
extern "C" void pd_dbg_set_wgrad_fast(int v) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wgrad_fast), &v, sizeof(int)); }
static uint32_t *const g_relu_bits = nullptr;   // plain pd_gemm_tn_f32x3 launches do not record the sign bits
int g_pd_dbg_x3_narrow = 0;   // tools/ only (pd_debug_set "x3_narrow"): 1 = never use the 256 x 256 kernel
int g_pd_dbg_x3 = 0;   // tools/ only (pd_debug_set "x3_ablate"): 1 no MFMA, 2 only hi*hi, 3 no operand split, 4 no output stores

extern "C" int pd_gemm_tn_f32x3(const float *A, const float *B, const float *bias, float *C, int M, int N, int K, int lda,
                                int ldb, int ldc, int relu, void *stream_)
{
  if (M < 0 || N < 0 || K < 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f32x3: negative size");
  if (M == 0 || N == 0) return PD_OK;
  if (!A || !B || !C) return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f32x3: null pointer");
  if ((K & 3) || (lda & 3) || (ldb & 3) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f32x3: K, lda, ldb must be multiples of 4 and A, B 16-byte aligned");
  hipStream_t st = (hipStream_t)stream_;
  // wide tiles where they were measured to pay (tools/bench_gemm_x3_wide.py, M = 43 008): 1024 <- 256: 161-169 vs 202-216 us; 256 <- 1024:
  // 143 vs 167; at K = 256 with N = 256 / 512 the two kernels tie or the narrow one wins; few tiles (< 128) leave CUs idle
  if ((!g_pd_dbg_x3 || g_pd_dbg_x3 > 10) && g_pd_dbg_x3_narrow != 1 && (N % WBN) == 0 && M >= 4 * WBM &&
      ((int64_t)((M + WBM - 1) / WBM) * (N / WBN) >= 128 || g_pd_dbg_x3_narrow >= 2) && (N >= 1024 || K >= 512 || g_pd_dbg_x3_narrow >= 2)) {
    const int wtn = N / WBN, wtm = (M + WBM - 1) / WBM;
    const size_t lds = (size_t)2 * 3 * 2 * (WBM + WBN) * 8 * sizeof(bf16_t);
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void *)gemm_tn_f32x3_wide<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void *)gemm_tn_f32x3_wide<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr = true;
    }
    if (g_pd_dbg_x3_narrow == 3) {                             // tools: 128 x 256 tiles, two workgroups per CU
      const int htm = (M + 127) / 128;
      const size_t hlds = (size_t)2 * 3 * 2 * (128 + WBN) * 8 * sizeof(bf16_t);
      (void)hipFuncSetAttribute((const void *)gemm_tn_f32x3_wide<false, 0, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hlds);
      hipLaunchKernelGGL((gemm_tn_f32x3_wide<false, 0, false, 2>), dim3((unsigned)((int64_t)htm * wtn)), dim3(256), hlds, st, A, B, bias, C, M, N, K,
                         lda, ldb, ldc, wtn, (uint32_t *)nullptr, (float *)nullptr);
      return pd_check_launch("pd_gemm_tn_f32x3");
    }
    const dim3 wg((unsigned)((int64_t)wtm * wtn)), wb(512);
    if (g_pd_dbg_x3 > 10) {
      auto kf = g_pd_dbg_x3 == 11 ? gemm_tn_f32x3_wide<false, 1> : g_pd_dbg_x3 == 12 ? gemm_tn_f32x3_wide<false, 2> : gemm_tn_f32x3_wide<false, 3>;
      (void)hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kf, wg, wb, lds, st, A, B, bias, C, M, N, K, lda, ldb, ldc, wtn, (uint32_t *)nullptr, (float *)nullptr);
      return pd_check_launch("pd_gemm_tn_f32x3");
    }
    if (relu) hipLaunchKernelGGL(gemm_tn_f32x3_wide<true>, wg, wb, lds, st, A, B, bias, C, M, N, K, lda, ldb, ldc, wtn, g_relu_bits, (float *)nullptr);
    else hipLaunchKernelGGL(gemm_tn_f32x3_wide<false>, wg, wb, lds, st, A, B, bias, C, M, N, K, lda, ldb, ldc, wtn, (uint32_t *)nullptr, (float *)nullptr);
    return pd_check_launch("pd_gemm_tn_f32x3");
  }
  const int tn = (N + BN - 1) / BN, tm = (M + BM - 1) / BM;
  const dim3 g((unsigned)((int64_t)tm * tn)), b(256);
#define LAUNCH(R, AB) hipLaunchKernelGGL((gemm_tn_f32x3<R, AB, false>), g, b, 0, st, A, B, bias, C, M, N, K, lda, ldb, ldc, tn, 0, 0)
  if (g_pd_dbg_x3 == 1) LAUNCH(false, 1);
  else if (g_pd_dbg_x3 == 2) LAUNCH(false, 2);
  else if (g_pd_dbg_x3 == 3) LAUNCH(false, 3);
  else if (g_pd_dbg_x3 == 4) LAUNCH(false, 4);
  else if (relu) LAUNCH(true, 0);
  else LAUNCH(false, 0);
#undef LAUNCH
  return pd_check_launch("pd_gemm_tn_f32x3");
}

namespace {
// W fp32 [N, K] (row stride ldw) -> three bf16 planes [3][R][C], (R, C) = (N, K) or, TR, (K, N) (the planes of W^T)
template <bool TR>
__global__ __launch_bounds__(256) void split3_planes(const float *__restrict__ W, int N, int K, int ldw, bf16_t *__restrict__ out)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, total = (int64_t)N * K;
  if (i >= total) return;
  const int r = (int)(i / (TR ? N : K)), c = (int)(i - (int64_t)r * (TR ? N : K));     // output row / column
  const float x = TR ? W[(int64_t)c * ldw + r] : W[(int64_t)r * ldw + c];
  unsigned h, m, l;
  split2(x, 0.f, h, m, l);
  out[i] = (bf16_t)(h & 0xffffu); out[total + i] = (bf16_t)(m & 0xffffu); out[2 * total + i] = (bf16_t)(l & 0xffffu);
}
}  // namespace

extern "C" int pd_split3_bf16(const float *W, int N, int K, int ldw, int transpose, void *planes, void *stream_)
{
  if (N <= 0 || K <= 0 || !W || !planes) return pd_set_error(PD_ERR_INVALID_ARG, "pd_split3_bf16: N=%d K=%d or null pointer", N, K);
  const int64_t total = (int64_t)N * K;
  const dim3 g((unsigned)((total + 255) / 256)), b(256);
  if (transpose) hipLaunchKernelGGL(split3_planes<true>, g, b, 0, (hipStream_t)stream_, W, N, K, ldw, (bf16_t *)planes);
  else hipLaunchKernelGGL(split3_planes<false>, g, b, 0, (hipStream_t)stream_, W, N, K, ldw, (bf16_t *)planes);
  return pd_check_launch("pd_split3_bf16");
}

// mode 0: C = A B^T + bias; 1: relu(...) (+ sign bits when `bits`); 2: (A B^T) masked by `bits`, colsum += column sums
extern "C" int pd_gemm_tn_f32x3_pre(const float *A, const void *Bplanes, const float *bias, float *C, uint32_t *bits, float *colsum, int M,
                                    int N, int K, int lda, int ldc, int mode, void *stream_)
{
  if (M <= 0 || N <= 0 || K <= 0 || (N % WBN) || (K & 15) || M < 4 * WBM || (lda & 3) || ((uintptr_t)A & 15) || ((uintptr_t)Bplanes & 15))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f32x3_pre: needs N %% 256 == 0, K %% 16 == 0, M >= 1024, aligned operands");
  if (!A || !Bplanes || !C || (mode == 2 && (!bits || !colsum)) || mode < 0 || mode > 2)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f32x3_pre: null pointer / bad mode");
  const int wtn = N / WBN, wtm = (M + WBM - 1) / WBM;
  const size_t lds = (size_t)2 * 3 * 2 * (WBM + WBN) * 8 * sizeof(bf16_t);
  typedef void (*kfn)(const float *, const float *, const float *, float *, int, int, int, int, int, int, int, uint32_t *, float *);
  const kfn k = mode == 0 ? (kfn)gemm_tn_f32x3_wide<false, 0, false, 4, true> : mode == 1 ? (kfn)gemm_tn_f32x3_wide<true, 0, false, 4, true>
                                                                                            : (kfn)gemm_tn_f32x3_wide<false, 0, true, 4, true>;
  static bool attr[3] = {false, false, false};
  if (!attr[mode]) { (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr[mode] = true; }
  hipLaunchKernelGGL(k, dim3((unsigned)((int64_t)wtm * wtn)), dim3(512), lds, (hipStream_t)stream_, A, (const float *)Bplanes, bias, C, M, N, K, lda, K,
                     ldc, wtn, bits, colsum);
  return pd_check_launch("pd_gemm_tn_f32x3_pre");
}

extern "C" int64_t pd_gemm_tn_f32x3_relu_bits_words(int M, int N)
{
  if (M <= 0 || N <= 0 || (N % WBN)) return 0;
  return (int64_t)((M + WBM - 1) / WBM) * (N / WBN) * 8 * 64 * 4;
}

extern "C" int pd_gemm_tn_f32x3_relu_bits(const float *A, const float *B, const float *bias, float *C, uint32_t *bits, int M, int N, int K,
                                          int lda, int ldb, int ldc, void *stream_)
{
  if (M <= 0 || N <= 0 || (N % WBN) || M < 4 * WBM || !bits)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f32x3_relu_bits: needs N %% 256 == 0, M >= 1024 and a bits buffer");
  if (!A || !B || !C) return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f32x3_relu_bits: null pointer");
  if ((K & 3) || (lda & 3) || (ldb & 3) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f32x3_relu_bits: K, lda, ldb must be multiples of 4 and A, B 16-byte aligned");
  const int wtn = N / WBN, wtm = (M + WBM - 1) / WBM;
  const size_t lds = (size_t)2 * 3 * 2 * (WBM + WBN) * 8 * sizeof(bf16_t);
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void *)gemm_tn_f32x3_wide<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
  hipLaunchKernelGGL(gemm_tn_f32x3_wide<true>, dim3((unsigned)((int64_t)wtm * wtn)), dim3(512), lds, (hipStream_t)stream_, A, B, bias, C, M, N, K,
                     lda, ldb, ldc, wtn, bits, (float *)nullptr);
  return pd_check_launch("pd_gemm_tn_f32x3_relu_bits");
}

extern "C" int pd_gemm_tn_f32x3_relumask(const float *A, const float *B, const uint32_t *bits, float *C, float *colsum, int M, int N, int K,
                                         int lda, int ldb, int ldc, void *stream_)
{
  if (M < 0 || N < 0 || K < 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f32x3_relumask: negative size");
  if (M == 0 || N == 0) return PD_OK;
  if (!A || !B || !C || !bits || !colsum) return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f32x3_relumask: null pointer");
  if ((K & 3) || (lda & 3) || (ldb & 3) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || (N % WBN) || M < 4 * WBM)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f32x3_relumask: K, lda, ldb multiples of 4, N a multiple of 256, M >= 1024, A, B 16-byte aligned");
  const int wtn = N / WBN, wtm = (M + WBM - 1) / WBM;
  const size_t lds = (size_t)2 * 3 * 2 * (WBM + WBN) * 8 * sizeof(bf16_t);
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void *)gemm_tn_f32x3_wide<false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
  hipLaunchKernelGGL((gemm_tn_f32x3_wide<false, 0, true>), dim3((unsigned)((int64_t)wtm * wtn)), dim3(512), lds, (hipStream_t)stream_, A, B,
                     (const float *)nullptr, C, M, N, K, lda, ldb, ldc, wtn, const_cast<uint32_t *>(bits), colsum);
  return pd_check_launch("pd_gemm_tn_f32x3_relumask");
}

int g_pd_dbg_wgrad_wide = 1;   // tools only (pd_debug_set "wgrad_wide" 0: the 128 x 128 kernel for every f16x2 weight gradient)
// the f16x2 weight gradient takes 256 x 256 tiles when the output has several of them (measured at M = 43 008, tools/bench_wgrad_x3.py:
// 1024 x 256 104 vs 126 us, 256 x 2304 195 vs 257; a single 256 x 256 tile LOSES, 48 vs 42 us — 256 slices of rows each leave a 256 KB
// partial tile) and, for the 3 x 3 convolution, a tile's 256 X columns lie inside one tap
static bool wgrad_wide_ok(int N, int K, int conv_ci)
{
  return g_pd_dbg_wgrad_wide && N > 128 && K > 128 && (N >= 1024 || K >= 1024) && (conv_ci == 0 || conv_ci % WTD == 0);
}
extern "C" int pd_gemm_wgrad_f16x2_takes_wide_tiles(int N, int K) { return wgrad_wide_ok(N, K, 0) ? 1 : 0; }

static int wgrad_x3_launch(const float *dY, const float *X, float *dW, float *dB, float *ws, int64_t ws_floats, int M, int N, int K, int ldy,
                           int ldx, int ldw, hipStream_t st, const char *who, int convH = 0, int convW = 0, bool h2 = false,
                           const float *y_amax = nullptr, const float *x_amax = nullptr)
{
  if (M < 0 || N < 0 || K < 0) return pd_set_error(PD_ERR_INVALID_ARG, "%s: negative size", who);
  if (N == 0 || K == 0 || M == 0) return PD_OK;
  if (!dW || !dY || !X) return pd_set_error(PD_ERR_INVALID_ARG, "%s: null pointer", who);
  const int tk = (K + BM - 1) / BM, tn = (N + BN - 1) / BN, tiles = tk * tn;
  // the contraction is split over workgroups (each ends in a tile of atomics, or of plain stores into the workspace).  Two
  // workgroups are resident per CU, so the launch is sized to ONE round of 512: 1 044 workgroups (the old "~4 per CU" rule at
  // 256 x 2304) ran as 3 rounds, the last one almost empty — 387 us instead of 260.  g_pd_dbg_x3 22: the old rule.
  int splits;
  if (g_pd_dbg_x3 == 22) { const int target = tiles <= 4 ? 256 : 1024; splits = (target + tiles - 1) / tiles; }
  else splits = tiles >= 512 ? 1 : (512 + tiles / 2) / tiles;
  int m_chunk = ((M + splits - 1) / splits + WS - 1) / WS * WS;
  if (m_chunk < 4 * WS) m_chunk = 4 * WS;
  splits = (M + m_chunk - 1) / m_chunk;
  // transpose-read form when the operands allow 16-byte row loads (always, in this repo); x3_ablate 21 forces the scalar-staged one
  const bool vec = !(N & 3) && !(K & 3) && !(ldy & 3) && !(ldx & 3) && !((uintptr_t)dY & 15) && !((uintptr_t)X & 15) && g_pd_dbg_x3 != 21;
  if (convH && !vec) return pd_set_error(PD_ERR_INVALID_ARG, "%s: channel counts must be multiples of 4 and the tensors 16-byte aligned", who);
  if (h2 && !vec) return pd_set_error(PD_ERR_INVALID_ARG, "%s: N, K, ldy, ldx must be multiples of 4 and dY, X 16-byte aligned", who);
  if (!vec) {
    hipLaunchKernelGGL(gemm_wgrad_f32x3, dim3((unsigned)(tiles * splits)), dim3(256), 0, st, dY, X, dW, dB, M, N, K, ldy, ldx, ldw, tk, tiles,
                       m_chunk);
    return pd_check_launch(who);
  }
  if (h2 && wgrad_wide_ok(N, K, convH ? ldx : 0)) {
    // 256 x 256 tiles, one workgroup per CU: one round of 256
    const int wtk = (K + WTD - 1) / WTD, wtn = (N + WTD - 1) / WTD, wtiles = wtk * wtn;
    int wsplits = wtiles >= 256 ? 1 : (256 + wtiles / 2) / wtiles;
    int mc = ((M + wsplits - 1) / wsplits + TWS - 1) / TWS * TWS;
    if (mc < 4 * TWS) mc = 4 * TWS;
    wsplits = (M + mc - 1) / mc;
    if (ws && (ws_floats < (int64_t)wtiles * wsplits * WTD * WTD || wsplits < 2)) ws = nullptr;
    constexpr size_t lds = (size_t)2 * 2 * 2 * TWS * WTP * sizeof(bf16_t);
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void *)gemm_wgrad_f16x2_wide<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void *)gemm_wgrad_f16x2_wide<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr = true;
    }
    auto kw = convH ? gemm_wgrad_f16x2_wide<true> : gemm_wgrad_f16x2_wide<false>;
    hipLaunchKernelGGL(kw, dim3((unsigned)(wtiles * wsplits)), dim3(WNT), lds, st, dY, X, dW, dB, ws, M, N, K, ldy, ldx, ldw, wtk, wtiles, mc, convH, convW,
                       y_amax, x_amax);
    const int wgroups = wtiles * 128 >= 2048 ? 1 : wsplits >= 64 ? 8 : wsplits >= 16 ? 4 : 1;
    if (ws) hipLaunchKernelGGL(wgrad_h2w_reduce, dim3((unsigned)(wtiles * 128), wgroups), dim3(WNT), 0, st, (const float *)ws, dW, N, K, ldw, wtk, wtiles, wsplits);
    return pd_check_launch(who);
  }
  if (ws && (ws_floats < (int64_t)tiles * splits * BN * BM || splits < 2 || g_pd_dbg_x3 == 25)) ws = nullptr;   // not worth / does not fit: atomics
  auto kfn = h2 ? (convH ? gemm_wgrad_f32x3_tr<0, true, true> : gemm_wgrad_f32x3_tr<0, false, true>)
                 : convH ? gemm_wgrad_f32x3_tr<0, true> : g_pd_dbg_x3 == 24 ? gemm_wgrad_f32x3_tr<2, false> : gemm_wgrad_f32x3_tr<0, false>;
  hipLaunchKernelGGL(kfn, dim3((unsigned)(tiles * splits)), dim3(256), 0, st, dY, X, dW, dB, ws, M, N, K, ldy, ldx, ldw, tk, tiles, m_chunk,
                     convH, convW, y_amax, x_amax);
  const int groups = tiles * 64 >= 2048 ? 1 : splits >= 64 ? 8 : splits >= 16 ? 4 : 1;
  if (ws) hipLaunchKernelGGL(wgrad_tr_reduce, dim3((unsigned)(tiles * 64), groups), dim3(256), 0, st, (const float *)ws, dW, N, K, ldw, tk, tiles,
                             splits);
  return pd_check_launch(who);
}


// ---- grouped weight gradients (include/pd_gemm.h: pd_gemm_wgrad_f32x3_grouped)
extern "C" int64_t pd_gemm_wgrad_f32x3_grouped_table_bytes(int max_count) { return (int64_t)(max_count > 0 ? max_count : 0) * (int64_t)sizeof(WgradX3Problem); }

static bool grouped_wide(const PdGemmWgradDesc *d, int count)
{
  for (int i = 0; i < count; ++i) if (!wgrad_wide_ok(d[i].N, d[i].K, 0)) return false;
  return count > 0;
}

// TD: tile edge (128, or 256 for the wide f16x2 kernel: one workgroup per CU, rounds of 256)
static int grouped_plan(const PdGemmWgradDesc *d, int count, WgradX3Problem *tab, int64_t *ws_tiles, int *blocks, int TD = BM)
{
  int64_t total_tiles = 0;
  for (int i = 0; i < count; ++i) {
    if (d[i].M <= 0 || d[i].N <= 0 || d[i].K <= 0 || !d[i].dY || !d[i].X || !d[i].dW) return 1;
    if ((d[i].N & 3) || (d[i].K & 3) || (d[i].ldy & 3) || (d[i].ldx & 3) || ((uintptr_t)d[i].dY & 15) || ((uintptr_t)d[i].X & 15)) return 1;
    total_tiles += (int64_t)((d[i].K + TD - 1) / TD) * ((d[i].N + TD - 1) / TD);
  }
  // one split count for all problems, chosen so that the launch is a whole number of rounds of 512 resident workgroups (two per
  // CU): tiles x splits just below a multiple of 512, at least ~6 rounds deep so that unequal M do not leave a ragged tail
  int splits = (int)((6 * (TD == BM ? 512 : 256)) / (total_tiles > 0 ? total_tiles : 1));
  if (splits < 1) splits = 1;
  int64_t wt = 0; int b0 = 0;
  for (int i = 0; i < count; ++i) {
    WgradX3Problem &q = tab[i];
    q.dY = d[i].dY; q.X = d[i].X; q.dW = d[i].dW; q.dB = d[i].dB;
    q.M = d[i].M; q.N = d[i].N; q.K = d[i].K; q.ldy = d[i].ldy; q.ldx = d[i].ldx; q.ldw = d[i].ldw;
    q.tiles_k = (q.K + TD - 1) / TD; q.tiles = q.tiles_k * ((q.N + TD - 1) / TD);
    int mc = ((q.M + splits - 1) / splits + TWS - 1) / TWS * TWS;
    if (mc < 4 * TWS) mc = 4 * TWS;
    q.m_chunk = mc; q.splits = (q.M + mc - 1) / mc;
    q.block0 = b0; q.pad = 0; q.ws_tile0 = wt;
    q.y_amax = d[i].y_amax; q.x_amax = d[i].x_amax;
    b0 += q.tiles * q.splits; wt += (int64_t)q.tiles * q.splits;
  }
  *ws_tiles = wt; *blocks = b0;
  return 0;
}

extern "C" int64_t pd_gemm_wgrad_f32x3_grouped_ws_floats(const PdGemmWgradDesc *descs, int count)
{
  if (count <= 0 || count > 256 || !descs) return 0;
  WgradX3Problem tab[256];
  int64_t wt = 0; int blocks = 0;
  if (grouped_plan(descs, count, tab, &wt, &blocks)) return -1;
  return wt * BN * BM;
}

static int wgrad_grouped(const PdGemmWgradDesc *descs, int count, void *table_host_pinned, void *table_device, float *workspace,
                         int64_t workspace_floats, void *stream_, bool h2)
{
  if (count == 0) return PD_OK;
  if (count < 0 || count > 256 || !descs || !table_host_pinned || !table_device || !workspace)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_wgrad_f32x3_grouped: count=%d (1..256) or null pointer", count);
  WgradX3Problem *tab = reinterpret_cast<WgradX3Problem *>(table_host_pinned);
  int64_t wt = 0; int blocks = 0;
  const bool wide = h2 && grouped_wide(descs, count);
  const int TD = wide ? WTD : BM;
  if (grouped_plan(descs, count, tab, &wt, &blocks, TD))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_wgrad_f32x3_grouped: every problem needs M, N, K > 0, N, K, ldy, ldx multiples of 4, 16-byte aligned operands");
  if (workspace_floats < wt * TD * TD) return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_wgrad_f32x3_grouped: workspace too small");
  hipStream_t st = (hipStream_t)stream_;
  if (hipMemcpyAsync(table_device, table_host_pinned, (size_t)count * sizeof(WgradX3Problem), hipMemcpyHostToDevice, st) != hipSuccess)
    return pd_set_error(PD_ERR_LAUNCH, "pd_gemm_wgrad_f32x3_grouped: table upload failed");
  const WgradX3Problem *dt = reinterpret_cast<const WgradX3Problem *>(table_device);
  int64_t tiles = 0;
  for (int i = 0; i < count; ++i) tiles += tab[i].tiles;
  if (wide) {
    constexpr size_t lds = (size_t)2 * 2 * 2 * TWS * WTP * sizeof(bf16_t);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)gemm_wgrad_f16x2_wide_grouped, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    hipLaunchKernelGGL(gemm_wgrad_f16x2_wide_grouped, dim3((unsigned)blocks), dim3(WNT), lds, st, dt, count, workspace);
    hipLaunchKernelGGL(wgrad_h2w_reduce_grouped, dim3((unsigned)(tiles * 128)), dim3(WNT), 0, st, dt, count, (const float *)workspace);
    return pd_check_launch("pd_gemm_wgrad_f16x2_grouped");
  }
  if (h2) hipLaunchKernelGGL(gemm_wgrad_f32x3_tr_grouped<true>, dim3((unsigned)blocks), dim3(256), 0, st, dt, count, workspace);
  else hipLaunchKernelGGL(gemm_wgrad_f32x3_tr_grouped<false>, dim3((unsigned)blocks), dim3(256), 0, st, dt, count, workspace);
  hipLaunchKernelGGL(wgrad_tr_reduce_grouped, dim3((unsigned)(tiles * 64)), dim3(256), 0, st, dt, count, (const float *)workspace);
  return pd_check_launch("pd_gemm_wgrad_f32x3_grouped");
}

extern "C" int pd_gemm_wgrad_f32x3_grouped(const PdGemmWgradDesc *descs, int count, void *table_host_pinned, void *table_device,
                                           float *workspace, int64_t workspace_floats, void *stream_)
{
  return wgrad_grouped(descs, count, table_host_pinned, table_device, workspace, workspace_floats, stream_, false);
}

extern "C" int pd_gemm_wgrad_f16x2_grouped(const PdGemmWgradDesc *descs, int count, void *table_host_pinned, void *table_device,
                                           float *workspace, int64_t workspace_floats, void *stream_)
{
  return wgrad_grouped(descs, count, table_host_pinned, table_device, workspace, workspace_floats, stream_, true);
}

extern "C" int64_t pd_gemm_wgrad_f16x2_ws_floats(int N, int K)
{
  const int64_t narrow = pd_gemm_wgrad_f32x3_ws_floats(N, K);
  if (N <= 0 || K <= 0) return narrow;
  const int64_t wt = (int64_t)((K + WTD - 1) / WTD) * ((N + WTD - 1) / WTD);
  const int64_t wide = wt >= 256 ? 0 : (256 + wt / 2) / wt * wt * WTD * WTD;
  return wide > narrow ? wide : narrow;                          // whichever tile shape the launch takes
}

extern "C" int64_t pd_gemm_wgrad_f16x2_grouped_ws_floats(const PdGemmWgradDesc *descs, int count)
{
  if (count <= 0 || count > 256 || !descs) return 0;
  WgradX3Problem tab[256];
  int64_t wt = 0; int blocks = 0;
  const int TD = grouped_wide(descs, count) ? WTD : BM;
  if (grouped_plan(descs, count, tab, &wt, &blocks, TD)) return -1;
  return wt * TD * TD;
}

extern "C" int pd_gemm_wgrad_acc_f16x2_ws(const float *dY, const float *X, float *dW, float *dB, const float *y_amax, const float *x_amax,
                                          float *workspace, int64_t workspace_floats, int M, int N, int K, int ldy, int ldx, int ldw, void *stream_)
{
  return wgrad_x3_launch(dY, X, dW, dB, workspace, workspace_floats, M, N, K, ldy, ldx, ldw, (hipStream_t)stream_, "pd_gemm_wgrad_acc_f16x2_ws", 0, 0,
                         true, y_amax, x_amax);
}

extern "C" int pd_conv3x3_wgrad_nhwc_f16x2(const float *dY, const float *X, float *dWk, float *dB, const float *y_amax, const float *x_amax,
                                           float *workspace, int64_t workspace_floats, int B, int H, int W, int Ci, int Co, void *stream_)
{
  if (B < 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || (Ci % BM) || (Co & 3))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_conv3x3_wgrad_nhwc_f16x2: B=%d H=%d W=%d Ci=%d (%% 128) Co=%d (%% 4)", B, H, W, Ci, Co);
  const int64_t M = (int64_t)B * H * W;
  if (M > 0x7fffffffLL - 4096) return pd_set_error(PD_ERR_INVALID_ARG, "pd_conv3x3_wgrad_nhwc_f16x2: too many pixels");
  return wgrad_x3_launch(dY, X, dWk, dB, workspace, workspace_floats, (int)M, Co, 9 * Ci, Co, Ci, 9 * Ci, (hipStream_t)stream_,
                         "pd_conv3x3_wgrad_nhwc_f16x2", H, W, true, y_amax, x_amax);
}

extern "C" int pd_gemm_wgrad_acc_f32x3(const float *dY, const float *X, float *dW, float *dB, int M, int N, int K, int ldy, int ldx,
                                       int ldw, void *stream_)
{
  return wgrad_x3_launch(dY, X, dW, dB, nullptr, 0, M, N, K, ldy, ldx, ldw, (hipStream_t)stream_, "pd_gemm_wgrad_acc_f32x3");
}

extern "C" int64_t pd_gemm_wgrad_f32x3_ws_floats(int N, int K)
{
  const int64_t tiles = (int64_t)((K + BM - 1) / BM) * ((N + BN - 1) / BN);
  return tiles >= 512 ? 0 : (512 + tiles / 2) / tiles * tiles * BN * BM;     // <= 8.5 M floats (34 MB) whatever the shape
}

extern "C" int pd_gemm_wgrad_acc_f32x3_ws(const float *dY, const float *X, float *dW, float *dB, float *workspace, int64_t workspace_floats,
                                          int M, int N, int K, int ldy, int ldx, int ldw, void *stream_)
{
  return wgrad_x3_launch(dY, X, dW, dB, workspace, workspace_floats, M, N, K, ldy, ldx, ldw, (hipStream_t)stream_, "pd_gemm_wgrad_acc_f32x3_ws");
}

extern "C" int pd_conv3x3_wgrad_nhwc_f32x3(const float *dY, const float *X, float *dWk, float *dB, float *workspace, int64_t workspace_floats,
                                           int B, int H, int W, int Ci, int Co, void *stream_)
{
  if (B < 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || (Ci % BM) || (Co & 3))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_conv3x3_wgrad_nhwc_f32x3: B=%d H=%d W=%d Ci=%d (%% 128) Co=%d (%% 4)", B, H, W, Ci, Co);
  const int64_t M = (int64_t)B * H * W;
  if (M > 0x7fffffffLL - 4096) return pd_set_error(PD_ERR_INVALID_ARG, "pd_conv3x3_wgrad_nhwc_f32x3: too many pixels");
  return wgrad_x3_launch(dY, X, dWk, dB, workspace, workspace_floats, (int)M, Co, 9 * Ci, Co, Ci, 9 * Ci, (hipStream_t)stream_,
                         "pd_conv3x3_wgrad_nhwc_f32x3", H, W);
}

extern "C" int pd_conv3x3_nhwc_f32x3(const float *X, const float *Wk, const float *bias, float *Y, int B, int H, int W, int Ci, int Co,
                                     void *stream_)
{
  if (B < 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || (Ci & 15))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_conv3x3_nhwc_f32x3: B=%d H=%d W=%d Ci=%d (%% 16) Co=%d", B, H, W, Ci, Co);
  if (B == 0) return PD_OK;
  if (!X || !Wk || !Y || ((uintptr_t)X & 15) || ((uintptr_t)Wk & 15))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_conv3x3_nhwc_f32x3: null / misaligned pointer");
  const int64_t M = (int64_t)B * H * W;
  if (M > 0x7fffffffLL - 4096) return pd_set_error(PD_ERR_INVALID_ARG, "pd_conv3x3_nhwc_f32x3: too many pixels");
  const int tn = (Co + BN - 1) / BN, tm = (int)((M + BM - 1) / BM);
  hipLaunchKernelGGL((gemm_tn_f32x3<false, 0, true>), dim3((unsigned)((int64_t)tm * tn)), dim3(256), 0, (hipStream_t)stream_, X, Wk, bias, Y,
                     (int)M, Co, 9 * Ci, Ci, 9 * Ci, Co, tn, H, W);
  return pd_check_launch("pd_conv3x3_nhwc_f32x3");
}

