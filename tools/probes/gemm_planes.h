// fp32 GEMMs on the bf16 matrix cores from PRE-SPLIT operands (round 3).
//
// gemm_x3.hip splits every fp32 operand element into three bf16 values (hi + mid + lo == x exactly) while it stages a tile:
// global load -> registers -> ~25 VALU operations per float4 -> three ds_write_b64 — and its ablations show that staging and
// split are paid on top of the matrix time (the pipes take turns).  Here the operands arrive ALREADY split, as three bf16
// planes [3][rows][cols] written once by whoever produced the tensor (a LayerNorm / ReLU / sampling epilogue, or
// pd_split3_rows), and a tile is staged by direct-to-LDS loads (global_load_lds_dwordx4: no VGPRs, no VALU, no ds_write) into
// an NS-stage ring with counted vmcnt waits across raw barriers, so the only per-step vector work left is 18 ds_read_b128 for
// 48 MFMAs.  Six of the nine partial products are accumulated in fp32, smallest first (see gemm_x3.hip for the error bound).
//
//   gemm_tn_planes     C[M,N]  = A[M,K] . B[N,K]^T (+ bias)(ReLU)   tile (64 WVM) x 256 x 16, wave tile 64 x 128
//   gemm_wgrad_planes  dW[N,K] (+)= dY[M,N]^T . X[M,K]              tile 128 x 128, 16 contraction rows per step, transposed on the
//                                                                   way out of LDS by ds_read_b64_tr_b16
// LDS images are lane-linear (the DMA writes base + 16 lane), so bank-conflict freedom comes from permuting the SOURCE
// address per lane and applying the same permutation to the read address (CDNA4 guide, rule 21).
#ifndef PD_GEMM_PLANES_H
#define PD_GEMM_PLANES_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace pdplanes {

typedef unsigned short bf16_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 hwbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hwbf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short v4s16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr;
typedef __attribute__((address_space(1))) const void *glb_ptr;

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi)
{
  const f32x2 x = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(x, hwbf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
// x0, x1 -> packed (hi, mid, lo): hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid (exact: <= 8 significant bits left)
__device__ __forceinline__ void split2(float x0, float x1, unsigned &h, unsigned &m, unsigned &l)
{
  h = pk_bf16(x0, x1);
  const float r0 = x0 - bf_lo(h), r1 = x1 - bf_hi(h);
  m = pk_bf16(r0, r1);
  l = pk_bf16(r0 - bf_lo(m), r1 - bf_hi(m));
}

// PANEL layout of a split tensor X[R, C] (C % 16 == 0): planes[3][C/16][R][16] — for every 16-column chunk the rows lie back to
// back, 32 bytes each.  A GEMM stage wants, per operand row, the 16 contraction elements of one chunk: with row-major planes
// that is 32 bytes out of every 128-byte line (the L2 -> L1 path then carries 4x the useful bytes: measured 14 B/clk/CU, the
// whole kernel bound by it); in panel layout the 32 rows one DMA instruction stages are ONE contiguous kilobyte.
__global__ __launch_bounds__(256) void split3_panels(const float *__restrict__ x, int ld, bf16_t *__restrict__ out, int R, int C)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (row, 4-column group)
  const int cg = C >> 2;
  if (i >= (int64_t)R * cg) return;
  const int r = (int)(i / cg), c = (int)(i - (int64_t)r * cg) * 4;
  const float4 v = *reinterpret_cast<const float4 *>(x + (int64_t)r * ld + c);
  uint2 h, m, l;
  split2(v.x, v.y, h.x, m.x, l.x);
  split2(v.z, v.w, h.y, m.y, l.y);
  const int64_t ps = (int64_t)R * C, o = ((int64_t)(c >> 4) * R + r) * 16 + (c & 15);
  *reinterpret_cast<uint2 *>(out + o) = h;
  *reinterpret_cast<uint2 *>(out + ps + o) = m;
  *reinterpret_cast<uint2 *>(out + 2 * ps + o) = l;
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void mma16k(f32x16 &c, hwbf16x8 x, hwbf16x8 y) { c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0); }
__device__ __forceinline__ int xcd_chunk(int bid, int nb) { return (nb & 7) == 0 ? (bid & 7) * (nb >> 3) + (bid >> 3) : bid; }

// ------------------------------------------------------------------------------------------------------------------ tn
// Stage image: A planes [3][TBM rows][32 B] then B planes [3][256 rows][32 B]; a row's two 16-byte halves are swapped when
// bit 3 of the row is set, so that the 16 lanes ds_read_b128 serves together (rows {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}
// of an MFMA operand) hit 16 different 16-byte slots of the 256-byte bank row.
// ORIENT 0: acc = mfma(a, b): lane <-> output column (B row), register e <-> output row: dword stores, 128 contiguous bytes
//           per row and instruction.  ORIENT 1: acc = mfma(b, a): lane <-> output row, e <-> 4 consecutive columns: 16-byte stores.
// SCHED 0: every wave [barrier, DMA issue, LDS reads, MFMAs] in lock-step; 1: the DMA instructions spread between the MFMA groups;
//       2: two wave groups (one wave per SIMD each) half a step apart: while one group runs its 48 MFMAs the other issues its DMA
//          and LDS reads for the same step (WVM 4, NS 3).
// EPI 0: + bias; 1: + bias, ReLU; 10 + a (tools only): ablation a = 1 no MFMA, 2 no DMA in the loop, 3 no output stores, 4 no LDS
// reads, 5 MFMA + barriers only.
template <int WVM, int NS, int ORIENT, int EPI, int SCHED = 0>
__global__ __launch_bounds__(128 * WVM, 2) void gemm_tn_planes(const bf16_t *__restrict__ A, int64_t psA,
                                                               const bf16_t *__restrict__ B, int64_t psB,
                                                               const float *__restrict__ bias, float *__restrict__ C, int ldc,
                                                               int M, int N, int K, int ntn, int delay, unsigned long long *trace = nullptr)
{
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int TBM = 64 * WVM, TBN = 256, NW = 2 * WVM;
  // the second workgroup of every CU starts `delay` (x 64 clocks) late: co-resident workgroups that start together stay in phase —
  // both in their MFMA-bound contraction loops, then both in their HBM-bound epilogues — and the two resources take turns
  if (delay > 0 && blockIdx.x >= 256 && blockIdx.x < 512)
    for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(1);
  // tools: lane 0 of every wave of every 64th workgroup leaves time stamps (shader clocks): start, contraction loop entered, loop
  // done, stores issued
  auto stamp = [&](int slot) {
    if (trace && (blockIdx.x & 63) == 0 && (threadIdx.x & 63) == 0)
      trace[((blockIdx.x >> 6) * 8 + (threadIdx.x >> 6)) * 4 + slot] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  constexpr int ABL = EPI >= 10 ? EPI - 10 : 0;
  constexpr bool RELU = EPI == 1;
  constexpr int ABYTES = 3 * TBM * 32, STAGE = ABYTES + 3 * TBN * 32;
  constexpr int NIA = 3 * TBM / 32, NI = NIA + 3 * TBN / 32, G = NI / NW;      // 1 KB DMA instructions per stage / per wave
  static_assert(NI % NW == 0, "stage must divide over the waves");
  const int lb = xcd_chunk(blockIdx.x, gridDim.x);
  const int m0 = (lb / ntn) * TBM, n0 = (lb % ntn) * TBN;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 128;

  const char *src[G];
  int64_t adv[G];                                               // bytes from one 16-wide contraction chunk to the next: the panel height x 32
  int ldsoff[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int q = wave + NW * g;
    const bool isA = q < NIA;
    const int qq = isA ? q : q - NIA, rpb = isA ? TBM / 32 : TBN / 32;
    const int pl = qq / rpb, rb = qq - pl * rpb;
    const int rl = rb * 32 + (lane >> 1);                         // tile-local row
    const int c = (lane & 1) ^ ((rl >> 3) & 1);                   // source half of the 32-byte row segment
    if (isA) { src[g] = reinterpret_cast<const char *>(A + pl * psA + (int64_t)min(m0 + rl, M - 1) * 16 + c * 8); adv[g] = (int64_t)M * 32; }
    else { src[g] = reinterpret_cast<const char *>(B + pl * psB + (int64_t)min(n0 + rl, N - 1) * 16 + c * 8); adv[g] = (int64_t)N * 32; }
    ldsoff[g] = q * 1024;
  }
  auto issue = [&](int stage, int kt) {
#pragma unroll
    for (int g = 0; g < G; ++g)
      __builtin_amdgcn_global_load_lds((glb_ptr)(src[g] + kt * adv[g]), (lds_ptr)(smem + stage * STAGE + ldsoff[g]), 16, 0, 0);
  };
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int KT = K / 16;
  const int fr = lane & 31, fh = lane >> 5;
  const int foff = fr * 32 + ((fh ^ ((fr >> 3) & 1)) * 16);
  hwbf16x8 cfrag;
#pragma unroll
  for (int e = 0; e < 8; ++e) cfrag[e] = (__bf16)(float)(lane + e);
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < KT) issue(s, s);

  auto issue1 = [&](int stage, int kt, int g) {
    __builtin_amdgcn_global_load_lds((glb_ptr)(src[g] + kt * adv[g]), (lds_ptr)(smem + stage * STAGE + ldsoff[g]), 16, 0, 0);
  };
#define PD_MMA(I, J, PA, PB)                                                   \
  do {                                                                         \
    if (ABL == 1) { asm volatile("" ::"v"(a[PA][I]), "v"(b[PB][J])); }         \
    else if (ORIENT == 0) mma16k(acc[I][J], a[PA][I], b[PB][J]);               \
    else mma16k(acc[I][J], b[PB][J], a[PA][I]);                                \
  } while (0)
  if constexpr (SCHED == 2) {
    static_assert(SCHED != 2 || (WVM == 4 && NS == 3), "staggered schedule: 8 waves, 3 stages");
    const int grp = wave >> 2;                                    // waves 0-3 / 4-7: one per SIMD each
    auto waitv = [&](int kt) {                                    // tile kt + 1 landed (this wave's share); tile kt + 2 may stay in flight
      if (kt + 2 < KT) wait_vm<G>();
      else wait_vm<0>();
    };
    if (KT > 1) wait_vm<G>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();                   // group 1 runs one half-step behind
    for (int kt = 0; kt < KT; ++kt) {
      const int st = kt % 3;
      // ---- half-step A: DMA issue for tile kt + 2 (into the stage tile kt - 1 was read from by everyone two half-steps ago) and the
      // fragments of tile kt
      if (kt + 2 < KT) {
#pragma unroll
        for (int g = 0; g < G; ++g) issue1((st + 2) % 3, kt + 2, g);
      }
      const unsigned char *sb = smem + st * STAGE;
      hwbf16x8 a[3][2], b[3][4];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) a[p][i] = *reinterpret_cast<const hwbf16x8 *>(sb + p * (TBM * 32) + (wm + i * 32) * 32 + foff);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[p][j] = *reinterpret_cast<const hwbf16x8 *>(sb + ABYTES + p * (TBN * 32) + (wn + j * 32) * 32 + foff);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the reads are DONE before the barrier that lets the other group overwrite
      if (grp == 1) waitv(kt);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- half-step B: 48 MFMAs, nothing else
      __builtin_amdgcn_s_setprio(1);
#define PD_TERM(PA, PB)                                                      \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                              \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) PD_MMA(i, j, PA, PB);
      PD_TERM(2, 0) PD_TERM(0, 2) PD_TERM(1, 1) PD_TERM(1, 0) PD_TERM(0, 1) PD_TERM(0, 0)
#undef PD_TERM
      __builtin_amdgcn_s_setprio(0);
      if (grp == 0) waitv(kt);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
  } else {
  auto step = [&](int kt, int st) {
    // tile kt has landed once this wave's own loads for it are done (at most one later tile stays in flight) and every wave
    // has passed the barrier; the same barrier proves that everyone finished reading the stage the next issue overwrites
    if (NS == 2 || kt + 1 >= KT) wait_vm<0>();
    else wait_vm<G>();
    __builtin_amdgcn_s_barrier();
    const bool more = kt + NS - 1 < KT && ABL != 2 && ABL != 5;
    if (SCHED == 0 && more) issue((st + NS - 1) % NS, kt + NS - 1);
    const unsigned char *sb = (ABL == 4 || ABL == 5) ? smem + foff : smem + st * STAGE;
    hwbf16x8 a[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) a[p][i] = (ABL == 4 || ABL == 5) ? cfrag : *reinterpret_cast<const hwbf16x8 *>(sb + p * (TBM * 32) + (wm + i * 32) * 32 + foff);
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      hwbf16x8 b[3][4];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int j = 0; j < 2; ++j) b[p][jp * 2 + j] = (ABL == 4 || ABL == 5) ? cfrag : *reinterpret_cast<const hwbf16x8 *>(sb + ABYTES + p * (TBN * 32) + (wn + (jp * 2 + j) * 32) * 32 + foff);
      // SCHED 1: the G DMA instructions of the next tile go out one at a time between the 12 groups of 4 MFMAs
#define PD_TERM(IDX, PA, PB)                                                 \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                              \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) PD_MMA(i, jp * 2 + j, PA, PB); \
  if (SCHED == 1 && more) {                                                  \
    _Pragma("unroll") for (int g = 0; g < G; ++g)                            \
        if (g * 12 / G == jp * 6 + IDX) issue1((st + NS - 1) % NS, kt + NS - 1, g); \
  }
      PD_TERM(0, 2, 0) PD_TERM(1, 0, 2) PD_TERM(2, 1, 1) PD_TERM(3, 1, 0) PD_TERM(4, 0, 1) PD_TERM(5, 0, 0)
#undef PD_TERM
    }
  };
  stamp(1);
  for (int kt = 0; kt < KT; kt += NS) {
    step(kt, 0);
    if (kt + 1 < KT) step(kt + 1, 1);
    if (NS == 3 && kt + 2 < KT) step(kt + 2, 2);
  }
  }
#undef PD_MMA
  stamp(2);

  // Epilogue.  The interior tiles take a branch-free path: with a per-element `if (row < M)` hipcc waits vmcnt(0) inside every
  // guarded block (it cannot prove the bias load has been waited for on every path), and vmcnt counts stores too — each store
  // then waits for the previous one to be acknowledged: 128 serialised round trips per wave.
  const bool full = m0 + TBM <= M && n0 + TBN <= N;
  if ((ABL == 3 || ABL == 5) && acc[0][0][0] != 1.2345678e30f) return;
  if (ORIENT == 0) {
    // col = lane & 31 (B row), row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) (A row)
    float bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = bias ? bias[min(n0 + wn + j * 32 + (lane & 31), N - 1)] : 0.f;
    float *cp = C + (int64_t)(m0 + wm + 4 * (lane >> 5)) * ldc + n0 + wn + (lane & 31);
    if (full) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v = acc[i][j][e] + bv[j];
            if (RELU) v = fmaxf(v, 0.f);
            cp[(int64_t)(i * 32 + (e & 3) + 8 * (e >> 2)) * ldc + j * 32] = v;
          }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn + j * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            float v = acc[i][j][e] + bv[j];
            if (RELU) v = fmaxf(v, 0.f);
            if (row < M && col < N) C[(int64_t)row * ldc + col] = v;
          }
      }
    }
  } else {
    // row = lane & 31 (A row), cols = 8 g4 + 4 (lane >> 5) + 0..3 of the 32-column tile (B rows)
    float4 bv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int col = min(n0 + wn + j * 32 + 8 * g4 + 4 * (lane >> 5), N - 4);
        bv[j][g4] = bias ? *reinterpret_cast<const float4 *>(bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    auto store = [&](auto guard) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = m0 + wm + i * 32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int col = n0 + wn + j * 32 + 8 * g4 + 4 * (lane >> 5);
            float4 v = make_float4(acc[i][j][4 * g4] + bv[j][g4].x, acc[i][j][4 * g4 + 1] + bv[j][g4].y, acc[i][j][4 * g4 + 2] + bv[j][g4].z,
                                   acc[i][j][4 * g4 + 3] + bv[j][g4].w);
            if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (!decltype(guard)::value || (row < M && col < N)) *reinterpret_cast<float4 *>(C + (int64_t)row * ldc + col) = v;
          }
      }
    };
    if (full) store(std::false_type{});
    else store(std::true_type{});
  }
  stamp(3);
}

template <int WVM, int NS, int ORIENT, int EPI, int SCHED = 0>
inline int launch_tn_planes_t(const bf16_t *A, int64_t psA, const bf16_t *B, int64_t psB, const float *bias, float *C, int ldc,
                              int M, int N, int K, hipStream_t st, int delay = 0, unsigned long long *trace = nullptr)
{
  constexpr int TBM = 64 * WVM, STAGE = 3 * 32 * (TBM + 256);
  const int ntn = (N + 255) / 256, ntm = (M + TBM - 1) / TBM;
  auto k = gemm_tn_planes<WVM, NS, ORIENT, EPI, SCHED>;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, NS * STAGE); attr = true; }
  hipLaunchKernelGGL(k, dim3((unsigned)((int64_t)ntm * ntn)), dim3(128 * WVM), (size_t)NS * STAGE, st, A, psA, B, psB, bias, C, ldc, M, N, K, ntn, delay, trace);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// A, B: panel-layout planes of [M, K] and [N, K] (plane strides psA, psB elements).  K % 16 == 0, N % 4 == 0.  Returns 0, 1 (unsupported combination) or 2 (launch error).
inline int launch_tn_planes(int wvm, int ns, int orient, int epi, int sched, const bf16_t *A, int64_t psA, const bf16_t *B, int64_t psB,
                            const float *bias, float *C, int ldc, int M, int N, int K, hipStream_t st, int delay = 0, unsigned long long *trace = nullptr)
{
  if (M <= 0 || N <= 0 || K <= 0 || (K & 15) || (N & 3)) return 1;
#define PD_L(W, S, O, E) if (wvm == W && ns == S && orient == O && epi == E && sched == 0) return launch_tn_planes_t<W, S, O, E>(A, psA, B, psB, bias, C, ldc, M, N, K, st, delay, trace);
#define PD_LS(W, S, E, SC) if (wvm == W && ns == S && orient == 0 && epi == E && sched == SC) return launch_tn_planes_t<W, S, 0, E, SC>(A, psA, B, psB, bias, C, ldc, M, N, K, st, delay, trace);
#define PD_LE(W, S, O) PD_L(W, S, O, 0) PD_L(W, S, O, 1)
  PD_LE(4, 2, 0) PD_LE(4, 3, 0) PD_LE(4, 3, 1) PD_LE(2, 2, 0) PD_LE(2, 2, 1)
  PD_L(4, 2, 0, 11) PD_L(4, 2, 0, 12) PD_L(4, 2, 0, 13) PD_L(4, 2, 0, 14) PD_L(4, 2, 0, 15)
  PD_L(2, 2, 0, 11) PD_L(2, 2, 0, 12) PD_L(2, 2, 0, 13) PD_L(2, 2, 0, 14) PD_L(2, 2, 0, 15)
  PD_LS(4, 2, 0, 1) PD_LS(4, 2, 1, 1) PD_LS(4, 3, 0, 1) PD_LS(4, 3, 1, 1) PD_LS(2, 2, 0, 1) PD_LS(2, 2, 1, 1) PD_LS(4, 3, 0, 2) PD_LS(4, 3, 1, 2) PD_LS(4, 3, 13, 2) PD_LS(4, 3, 11, 2)
#undef PD_LS
#undef PD_LE
#undef PD_L
  return 1;
}

// --------------------------------------------------------------------------------------------------------------- wgrad
// Stage image: [operand (dY, X)][plane][row group of 4][4 rows x 256 B] — one DMA instruction per row group.  A transpose
// read serves 32 lanes together: 4 rows x two 32-byte column blocks; with the 32-byte blocks of row r XOR-permuted by 2 r the
// four rows of a read sit in four different block pairs, i.e. its 256 bytes cover all 64 banks.
__device__ __forceinline__ hwbf16x8 frag_tr(const unsigned char *p, const unsigned char *p2)   // this lane's 8-byte segments in rows 0..3 / 4..7
{
  typedef __attribute__((address_space(3))) v4s16 *lp;
  union { v4s16 h[2]; hwbf16x8 v; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p));
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p2));
  return u.v;
}

template <int NS>
__global__ __launch_bounds__(256, 2) void gemm_wgrad_planes(const bf16_t *__restrict__ Y, int64_t psY, const bf16_t *__restrict__ X,
                                                            int64_t psX, float *__restrict__ dW, int ldw, float *__restrict__ ws,
                                                            int M, int N, int K, int tiles_k, int tiles, int m_chunk)
{
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int STAGE = 2 * 3 * 4 * 1024, G = 6;
  const int tile = blockIdx.x % tiles, split = blockIdx.x / tiles;
  const int n0 = (tile / tiles_k) * 128, k0 = (tile % tiles_k) * 128;
  const int mb = split * m_chunk, me = min(M, mb + m_chunk);
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;
  const char *src[G];
  int ldsoff[G];
  {
    // one DMA instruction = two 16-column chunks x 16 contraction rows x 32 B, each chunk's 512 bytes contiguous in the panel; the odd
    // chunk is staged with its row groups 0-3 / 4-7 (and 8-11 / 12-15) swapped so that the two chunks a transpose read touches
    // together (4 rows x 32 B each) fall into different bank halves
    const int ch = lane >> 5, rw = ((lane >> 1) & 15) ^ (4 * ch), half = lane & 1;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int q = wave + 4 * g, op = q / 12, pl = (q % 12) / 4, cp = q & 3;          // cp: chunk pair 0..3 of the 128 columns
      if (op == 0) src[g] = reinterpret_cast<const char *>(Y + pl * psY + ((int64_t)((n0 >> 4) + cp * 2 + ch) * M + mb + rw) * 16 + half * 8);
      else src[g] = reinterpret_cast<const char *>(X + pl * psX + ((int64_t)((k0 >> 4) + cp * 2 + ch) * M + mb + rw) * 16 + half * 8);
      ldsoff[g] = q * 1024;
    }
  }
  auto issue = [&](int stage, int stp) {
#pragma unroll
    for (int g = 0; g < G; ++g)
      __builtin_amdgcn_global_load_lds((glb_ptr)(src[g] + (int64_t)stp * 512), (lds_ptr)(smem + stage * STAGE + ldsoff[g]), 16, 0, 0);
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int steps = (me - mb) / 16;
  const int grp = lane >> 4, sl = lane & 15;
  // this lane's 8-byte segment: rows 8 (grp >> 1) + (sl >> 2) (first read; + 4 for the second), chunk (grp & 1) of the 32-column block
  // starting at chunk pair `cp`: image [chunk pair][chunk][16 rows][32 B], odd chunks with row bit 2 flipped
  auto seg = [&](int cp, int second) {
    const int ch = grp & 1, row = (8 * (grp >> 1) + (sl >> 2) + 4 * second) ^ (4 * ch);
    return cp * 1024 + ch * 512 + row * 32 + 8 * (sl & 3);
  };
  int offY[2][2], offX[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) { offY[i][h] = seg((wn >> 5) + i, h); offX[i][h] = 12288 + seg((wk >> 5) + i, h); }
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < steps) issue(s, s);
  auto step = [&](int stp, int st) {
    if (NS == 2 || stp + 1 >= steps) wait_vm<0>();
    else wait_vm<G>();
    __builtin_amdgcn_s_barrier();
    if (stp + NS - 1 < steps) issue((st + NS - 1) % NS, stp + NS - 1);
    const unsigned char *sb = smem + st * STAGE;
    hwbf16x8 a[3][2], b[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[p][i] = frag_tr(sb + p * 4096 + offY[i][0], sb + p * 4096 + offY[i][1]);
        b[p][i] = frag_tr(sb + p * 4096 + offX[i][0], sb + p * 4096 + offX[i][1]);
      }
#define PD_TERM(PA, PB)                                                      \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                              \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) mma16k(acc[i][j], a[PA][i], b[PB][j]);
    PD_TERM(2, 0) PD_TERM(0, 2) PD_TERM(1, 1) PD_TERM(1, 0) PD_TERM(0, 1) PD_TERM(0, 0)
#undef PD_TERM
  };
  for (int stp = 0; stp < steps; stp += NS) {
    step(stp, 0);
    if (stp + 1 < steps) step(stp + 1, 1);
    if (NS == 3 && stp + 2 < steps) step(stp + 2, 2);
  }
  if (ws) {
    // partial tile in REGISTER order (element (ij, e) of thread t at ((ij * 16 + e) * 256 + t): 1 KB per store instruction)
    float *w = ws + ((int64_t)split * tiles + tile) * (128 * 128) + t;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) w[((i * 2 + j) * 16 + e) * 256] = acc[i][j][e];
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
        if (row < N) unsafeAtomicAdd(dW + (int64_t)row * ldw + c, acc[i][j][e]);
      }
  }
}

// dW tile += sum over splits of the partial tiles in the workspace (same register-order indexing)
__global__ __launch_bounds__(256) void wgrad_planes_reduce(const float *__restrict__ ws, float *__restrict__ dW, int N, int K, int ldw, int tiles_k,
                                                           int tiles, int splits)
{
  const int tile = blockIdx.x >> 6, q = blockIdx.x & 63;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ij = q >> 4, e = q & 15, i = ij >> 1, j = ij & 1;
  const int n0 = (tile / tiles_k) * 128, k0 = (tile % tiles_k) * 128;
  const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;
  const float *p = ws + (int64_t)tile * (128 * 128) + q * 256 + t;
  const int64_t stride = (int64_t)tiles * (128 * 128);
  float s0 = 0.f, s1 = 0.f;
  int sp = blockIdx.y;
  const int Gy = gridDim.y;
  for (; sp + 7 * Gy < splits; sp += 8 * Gy) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(sp + u * Gy) * stride];
#pragma unroll
    for (int u = 0; u < 8; u += 2) { s0 += v[u]; s1 += v[u + 1]; }
  }
  for (; sp < splits; sp += Gy) s0 += p[(int64_t)sp * stride];
  const int c = k0 + wk + j * 32 + (lane & 31);
  const int row = n0 + wn + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
  if (c < K && row < N) unsafeAtomicAdd(dW + (int64_t)row * ldw + c, s0 + s1);
}

// Y, X: panel-layout planes of [M, N] and [M, K].  M % 16 == 0, N % 128 == 0, K % 128 == 0.  dW is accumulated into.
inline int launch_wgrad_planes(int ns, const bf16_t *Y, int64_t psY, const bf16_t *X, int64_t psX, float *dW, int ldw, float *ws,
                               int64_t ws_floats, int M, int N, int K, hipStream_t st)
{
  if (M <= 0 || (M & 15) || (N & 127) || (K & 127) || (ns != 2 && ns != 3)) return 1;
  const int tk = K / 128, tn = N / 128, tiles = tk * tn;
  int splits = tiles >= 512 ? 1 : (512 + tiles / 2) / tiles;
  int m_chunk = ((M + splits - 1) / splits + 15) / 16 * 16;
  if (m_chunk < 64) m_chunk = 64;
  splits = (M + m_chunk - 1) / m_chunk;
  if (ws && (ws_floats < (int64_t)tiles * splits * 16384 || splits < 2)) ws = nullptr;
  auto k = ns == 2 ? gemm_wgrad_planes<2> : gemm_wgrad_planes<3>;
  static bool attr[2] = {false, false};
  if (!attr[ns - 2]) { (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, ns * 24576); attr[ns - 2] = true; }
  hipLaunchKernelGGL(k, dim3((unsigned)(tiles * splits)), dim3(256), (size_t)ns * 24576, st, Y, psY, X, psX, dW, ldw, ws, M, N, K, tk, tiles, m_chunk);
  const int groups = tiles * 64 >= 2048 ? 1 : splits >= 64 ? 8 : splits >= 16 ? 4 : 1;
  if (ws) hipLaunchKernelGGL(wgrad_planes_reduce, dim3((unsigned)(tiles * 64), groups), dim3(256), 0, st, (const float *)ws, dW, N, K, ldw, tk, tiles, splits);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // namespace pdplanes
#endif
