// bf16 implicit GEMM with fused epilogues on the gfx950 matrix cores (C-ABI: include/pd_igemm.h): the R50 bottleneck
// convolutions (forward + input gradient), the Swin Linears and the decoder's key / value projections.
//
//   rows    = result pixels (all images), 128 per workgroup              columns = output channels, 128 or 64 per workgroup
//   K       = taps x source channels, 64 per step (source channels % 64 == 0: a step lies inside ONE tap, its A tile is 128
//             gathered 128-byte pixel rows; nothing is unfolded, a 1 x 1 convolution / Linear is the plain GEMM on the rows)
//
// What differs from csrc/conv_bf16.hip's conv_igemm_bf16 (round 2: parity-green, 1.3-4 x slower than MIOpen, never wired):
//   * tiles go global -> LDS DIRECTLY (global_load_lds_dwordx4, 1 KB per wavefront instruction): no staging registers, no
//     ds_write pass, the next step's loads are in flight while the matrix instructions of the current one run.  The LDS image
//     of such a load is lane-linear, so rows are 128 bytes with no padding and the bank-conflict fix is an XOR swizzle applied to
//     the SOURCE address (16-byte chunk c of row r holds global chunk c ^ ((r >> 1) & 7): the permutation stays inside one
//     128-byte line, coalescing is untouched) and again on the fragment read — ds_read_b128 is served in 16-lane groups of rows
//     {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (MI355X_MICROARCH.md, LDS), which this map spreads over all 16 slots of the
//     256-byte bank row;
//   * padded / out-of-image / beyond-M rows load from a zero line instead of branching around the load;
//   * 32 KB (one stage, four workgroups per CU) or 64 KB (two stages, two per CU) of LDS per workgroup: a workgroup's load wait,
//     barrier and epilogue are the other workgroups' compute;
//   * split-K inside the launch for the deep layers (K = 1024 .. 4608 over <= 8 192 pixels: 64-128 tiles would leave the chip
//     idle): fp32 partial tiles to a workspace, agent-scope release / ticket / acquire, the last workgroup of a tile to arrive
//     sums the slabs IN SPLIT ORDER (deterministic) and runs the epilogue;
//   * epilogue: acc * scale + bias (+ residual, dense or 2 x upsampled-by-zeros) -> optional copy of the pre-activation -> ReLU /
//     exact-erf GELU -> gate (ReLU mask or GELU' of a saved tensor), straight from the accumulators: the 32 x 32 result layout
//     leaves a lane with 4 consecutive channels of one pixel per register quad and lanes l / l + 32 with the two halves of an
//     8-channel piece, so one v_permlane32_swap per register pair gives every lane 8 consecutive channels — 16-byte stores (and
//     16-byte residual / gate loads) without a trip through LDS and without its two barriers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gelu.h"
#include "mfma_bf16.h"
#include "pd_common.h"
#include "xcd.h"
#include "pd_igemm.h"
#include "pd_msda.h"

namespace {
using namespace pdmfma;

typedef __bf16 hwbf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void *lds_ptr;
typedef __attribute__((address_space(1))) const void *glb_ptr;

constexpr int BM = 128, BKB = 128;                       // rows per tile, BYTES per row of a K-step (64 bf16)
__device__ __attribute__((aligned(128))) unsigned char g_zero_line[128];   // zero-initialised: the source of every masked row

}  // namespace
int g_ig_bn = 0, g_ig_nst = 0, g_ig_splits = 0;
int g_ig_pcls = []() { const char *e = getenv("PD_IG_PCLS"); return e ? atoi(e) : 1; }();
int g_ig_patch = []() { const char *e = getenv("PD_IG_PATCH"); return e ? atoi(e) : 1; }();   // 0: the gathered kernel for the 3 x 3 convolutions too (A/B)              // pd_debug_set "ig_bn" / "ig_nst" / "ig_splits" (tools/ only; 0 = automatic)
namespace {

struct IgArgs {
  const bf16_t *S, *Wf;
  const float *scale, *bias;
  const bf16_t *res, *res2, *gate;
  bf16_t *Y, *Ypre;
  float *slabs;
  unsigned *tickets;
  int M, N;
  int Cs, kw, cch, KT;                                   // source channels, filter width, 64-channel chunks per tap, K-steps
  int Hs, Ws, Ho, Wo;
  int stride, pad, dgrad;
  int act, gate_mode, res_mode, bias_bf16;
  int splits, kt_per;
  int ntn, ntiles;
  int pcls, tpc;                                         // stride-2 input gradient by parity class: tiles per class (see make_plan)
  int slabN;                                             // PdIgemm.out_col_slab
};

__device__ __forceinline__ int xcd_chunk(int bid, int nb) { return pd_xcd_chunk(bid, nb); }   // xcd.h: any workgroup count
__device__ __forceinline__ float gelu_f(float x) { return pdgelu::gelu(x); }
__device__ __forceinline__ float gelu_grad_f(float x) { return pdgelu::gelu_grad(x); }

// lanes l < 32 and l + 32 hold the channel quads [8q .. 8q+3] and [8q+4 .. 8q+7] of pixel l for q = 0..3.  After swapping the upper
// half of quad 2p with the lower half of quad 2p + 1, lane l < 32 holds channels 16p .. 16p+7 and lane l + 32 channels 16p+8 .. 16p+15.
__device__ __forceinline__ void pin_regs(uint4 &r) { asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w)); }
__device__ __forceinline__ void swap_halves(float &a, float &b)
{
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}

// split-K seam: this workgroup's partial tile -> its slab; the LAST workgroup of the tile to arrive (agent-scope release / ticket / acquire) sums
// all slabs in split order (deterministic) and goes on to the epilogue (-> true); the others are done (-> false)
template <int MI>
__device__ __forceinline__ bool splitk_seam(const IgArgs &a, f32x16 (&acc)[2][MI], int tile, int split, int t, unsigned char *smem)
{
  float4 *slab = reinterpret_cast<float4 *>(a.slabs) + ((int64_t)tile * a.splits + split) * (2 * MI * 4 * 256);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        slab[((i * MI + j) * 4 + q) * 256 + t] = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int *flag = reinterpret_cast<int *>(smem);
  if (t == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned old = __hip_atomic_fetch_add(a.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == (unsigned)(a.splits - 1);
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(a.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // left zero for the next launch
    }
    *flag = last;
  }
  __syncthreads();
  if (*flag == 0) return false;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const float4 *s0 = reinterpret_cast<const float4 *>(a.slabs) + (int64_t)tile * a.splits * (2 * MI * 4 * 256);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  for (int sp = 0; sp < a.splits; ++sp) {
    const float4 *sl = s0 + (int64_t)sp * (2 * MI * 4 * 256);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = sl[((i * MI + j) * 4 + q) * 256 + t];
          acc[i][j][4 * q] += v.x; acc[i][j][4 * q + 1] += v.y; acc[i][j][4 * q + 2] += v.z; acc[i][j][4 * q + 3] += v.w;
        }
  }
  return true;
}

template <int BN, int NST, bool P1>
__global__ __launch_bounds__(256, NST == 1 ? 4 : (NST == 2 || BN == 64) ? 2 : 1) void igemm_bf16(IgArgs a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = (BM + BN) * BKB;                 // bytes per stage: A rows then B rows
  constexpr int SB_OFF = NST * STAGE;                    // behind the stages: scale[BN], bias[BN] of this tile's columns (fp32)
  constexpr int MI = BN == 128 ? 2 : 1;                  // 32-pixel tiles per wavefront (wave grid 2 x 2 | 4 x 1)
  constexpr int NBJ = BN / 32;                           // B row groups (8 rows = 1 KB) per wavefront
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lb = xcd_chunk(blockIdx.x, gridDim.x);
  const int tile = lb / a.splits, split = lb - tile * a.splits;
  // parity classes (stride-2 input gradient): the row tiles of result pixels (2 py + cy, 2 px + cx) of ONE class (cy, cx) — a class only meets the
  // taps of its own parity (1, 2, 2 or 4 of the 9), so its contraction has ntap x cch steps instead of 9 x cch of which 3 / 4 multiply zeros
  const int mtile = tile / a.ntn, cls = a.pcls ? mtile / a.tpc : 0, cy = cls >> 1, cx = cls & 1;
  const int m0 = (a.pcls ? mtile - cls * a.tpc : mtile) * BM, n0 = (tile % a.ntn) * BN;
  const int ntx = cx ? 2 : 1, KTc = a.pcls ? (cy ? 2 : 1) * ntx * a.cch : a.KT;
  const int kt0 = split * a.kt_per, kt1 = min(KTc, kt0 + a.kt_per);
  const int wm = BN == 128 ? (wave >> 1) * 64 : wave * 32, wn = BN == 128 ? (wave & 1) * 64 : 0;

  // ---- the four A rows this thread fetches (16-byte chunk lane & 7 of rows (wave * 4 + j) * 8 + lane / 8)
  int bb[4], y0[4], x0[4], akc[4];
  const bf16_t *ap[4];                                   // P1: the row's address at K-step 0 (or the zero line)
  const int lrow = lane >> 3, lch = lane & 7;
  const bf16_t *zline = reinterpret_cast<const bf16_t *>(g_zero_line);
  // frozen-BN scale / bias of the tile's columns -> LDS (read back in the epilogue: one ds_read instead of a dependent global load
  // per 8-channel piece); issued first so that their latency overlaps the first tile's
  float sbv = 0.f;
  if (t < 2 * BN) {
    const float *src = t < BN ? a.scale : a.bias;
    if (t >= BN && a.bias && a.bias_bf16) sbv = __uint_as_float((unsigned)reinterpret_cast<const bf16_t *>(a.bias)[n0 + t - BN] << 16);
    else sbv = src ? src[n0 + (t < BN ? t : t - BN)] : (t < BN ? 1.f : 0.f);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (wave * 4 + j) * 8 + lrow, m = m0 + row;
    akc[j] = (lch ^ ((row >> 1) & 7)) * 8;
    if (P1) {
      ap[j] = m < a.M ? a.S + (int64_t)m * a.Cs + akc[j] : zline + lch * 8;
      bb[j] = m < a.M ? 0 : -1; y0[j] = 0; x0[j] = 0;
      continue;
    }
    if (a.pcls) {
      const int hh = a.Ho >> 1, wh = a.Wo >> 1, hw = hh * wh;
      if (m < a.M >> 2) {
        const int b = m / hw, rem = m - b * hw, py = rem / wh, px = rem - py * wh;
        bb[j] = b * a.Hs * a.Ws; y0[j] = 2 * py + cy + a.pad; x0[j] = 2 * px + cx + a.pad;
      } else {
        bb[j] = -1; y0[j] = 0; x0[j] = 0;
      }
    } else if (m < a.M) {
      const int hw = a.Ho * a.Wo, b = m / hw, rem = m - b * hw, oy = rem / a.Wo, ox = rem - oy * a.Wo;
      bb[j] = b * a.Hs * a.Ws;
      y0[j] = a.dgrad ? oy + a.pad : oy * a.stride - a.pad;
      x0[j] = a.dgrad ? ox + a.pad : ox * a.stride - a.pad;
    } else {
      bb[j] = -1; y0[j] = 0; x0[j] = 0;
    }
  }
  const bf16_t *wb[NBJ];
  const int ldw = a.KT * 64;
#pragma unroll
  for (int j = 0; j < NBJ; ++j) {
    const int row = (wave * NBJ + j) * 8 + lrow;
    wb[j] = a.Wf + (int64_t)(n0 + row) * ldw + (lch ^ ((row >> 1) & 7)) * 8;
  }

  int cur_tap = -1;                                        // (gathered form) the tap whose row offsets roff[] hold
  int64_t cur_kw = 0, roff[4] = {-1, -1, -1, -1};
  auto issue = [&](int kt, int buf) {
    unsigned char *As = smem + buf * STAGE, *Bs = As + BM * BKB;
    if (P1) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((glb_ptr)(bb[j] >= 0 ? ap[j] + (int64_t)kt * 64 : ap[j]), (lds_ptr)(As + (wave * 4 + j) * 1024), 16, 0, 0);
#pragma unroll
      for (int j = 0; j < NBJ; ++j)
        __builtin_amdgcn_global_load_lds((glb_ptr)(wb[j] + (int64_t)kt * 64), (lds_ptr)(Bs + (wave * NBJ + j) * 1024), 16, 0, 0);
      return;
    }
    const int tap = kt / a.cch, c0 = (kt - tap * a.cch) * 64;
    // the four rows' source pixels under a tap — bounds, (y, x) -> element offset with its 64-bit multiply — only when the TAP changes (every
    // cch steps; K-steps arrive in order): recomputed per step they made the step ~410 instructions around 8-16 MFMAs (round 6, the same
    // finding as the patch kernel's)
    if (tap != cur_tap) {
      cur_tap = tap;
      int dy = tap / a.kw, dx = tap - dy * a.kw;
      cur_kw = (int64_t)tap * a.cch;                       // the filter's first K-step of this tap
      if (a.pcls) {                                        // tap = index into the class's tap list: dy in {1} | {0, 2}, dx likewise
        const int ty_ = tap / ntx, tx_ = tap - ty_ * ntx;
        dy = cy ? 2 * ty_ : 1; dx = cx ? 2 * tx_ : 1;
        cur_kw = (int64_t)(dy * a.kw + dx) * a.cch;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int sy, sx;
        bool ok = bb[j] >= 0;
        if (a.dgrad) {
          const int ty = y0[j] - dy, tx = x0[j] - dx;
          ok = ok && ty >= 0 && tx >= 0;
          if (a.stride == 2) { ok = ok && ((ty | tx) & 1) == 0; sy = ty >> 1; sx = tx >> 1; }
          else { sy = ty; sx = tx; }
          ok = ok && sy < a.Hs && sx < a.Ws;
        } else {
          sy = y0[j] + dy; sx = x0[j] + dx;
          ok = ok && (unsigned)sy < (unsigned)a.Hs && (unsigned)sx < (unsigned)a.Ws;
        }
        roff[j] = ok ? ((int64_t)(bb[j] + sy * a.Ws + sx)) * a.Cs + akc[j] : -1;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16_t *p = roff[j] >= 0 ? a.S + roff[j] + c0 : zline + lch * 8;
      __builtin_amdgcn_global_load_lds((glb_ptr)p, (lds_ptr)(As + (wave * 4 + j) * 1024), 16, 0, 0);
    }
    const int64_t kw_ = cur_kw + (kt - tap * a.cch);
#pragma unroll
    for (int j = 0; j < NBJ; ++j)
      __builtin_amdgcn_global_load_lds((glb_ptr)(wb[j] + kw_ * 64), (lds_ptr)(Bs + (wave * NBJ + j) * 1024), 16, 0, 0);
  };

  f32x16 acc[2][MI];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int fr = lane & 31, kh = lane >> 5, sw = (fr >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = ((ks * 2 + kh) ^ sw) * 16;

  auto compute = [&](int buf) {
    const unsigned char *As = smem + buf * STAGE, *Bs = As + BM * BKB;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      hwbf16x8 wf[2], af[MI];
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[i] = *reinterpret_cast<const hwbf16x8 *>(Bs + (wn + i * 32 + fr) * BKB + foff[ks]);
#pragma unroll
      for (int j = 0; j < MI; ++j) af[j] = *reinterpret_cast<const hwbf16x8 *>(As + (wm + j * 32 + fr) * BKB + foff[ks]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
    }
  };

  if (t < 2 * BN) reinterpret_cast<float *>(smem + SB_OFF)[t] = sbv;
  if (NST == 3) {
    // three-stage ring, loads TWO steps ahead: the wait before the barrier leaves the next step's pieces in flight (counted vmcnt:
    // 4 + NBJ pieces per wavefront and step), and the barrier is the raw instruction — __syncthreads() would drain them
    if (kt0 < kt1) issue(kt0, 0);
    if (kt0 + 1 < kt1) issue(kt0 + 1, 1);
    int buf = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      if (kt + 1 < kt1) {
        if (NBJ == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                       // step kt landed for everyone; everyone is done reading step kt - 1's stage
      if (kt + 2 < kt1) issue(kt + 2, buf == 0 ? 2 : buf - 1);
      compute(buf);
      buf = buf == 2 ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
  } else if (NST == 2) {
    if (kt0 < kt1) issue(kt0, 0);
    for (int kt = kt0; kt < kt1; ++kt) {
      const int buf = (kt - kt0) & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's pieces of step kt have landed ...
      __syncthreads();                                    // ... everyone's have, and everyone is done reading the other stage
      if (kt + 1 < kt1) issue(kt + 1, buf ^ 1);
      compute(buf);
    }
  } else {
    for (int kt = kt0; kt < kt1; ++kt) {
      issue(kt, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute(0);
      __syncthreads();                                    // the stage is free for the next step's loads
    }
  }

  // ---- split-K: partial tile -> slab; the last workgroup of the tile to arrive sums all slabs in split order
  if (a.splits > 1 && !splitk_seam<MI>(a, acc, tile, split, t, smem)) return;

  // ---- epilogue, straight from the accumulators (see swap_halves)
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    int m = m0 + wm + j * 32 + fr;
    bool rowok = m < a.M;
    if (a.pcls) {                                          // class-local pixel -> its row of the result grid
      const int hh = a.Ho >> 1, wh = a.Wo >> 1, hw = hh * wh;
      rowok = m < a.M >> 2;
      const int mm = rowok ? m : 0, b = mm / hw, rem = mm - b * hw, py = rem / wh, px = rem - py * wh;
      m = (b * a.Ho + 2 * py + cy) * a.Wo + 2 * px + cx;
    }
    const float *sb = reinterpret_cast<const float *>(smem + SB_OFF);
    constexpr bool EPI_PIPE = !(BN == 128 && NST == 1);    // (that instantiation lives on 128 VGPRs: one batch of operands at a time)
#include "igemm_epilogue.inc"
  }
}

// ------------------------------------------------------------------------------------------------ 3 x 3, stride 1: patch form
// The gathered kernel above fetches the A tile of every K-step again — for a 3 x 3 convolution the nine taps of a channel chunk read the SAME
// source pixels nine times through L2 (one shifted copy per tap), and a step can only start when its 24 KB have landed.  Here a tile is an
// 8 x 16 BLOCK of result pixels; the source patch of a 64-channel chunk (10 x 18 pixels x 128 bytes, zero rows outside the image) goes into
// LDS ONCE and the nine taps read their fragments from it at shifted rows (same XOR swizzle, keyed by the patch row), so a step only waits for
// its 8 KB weight tile: 2.3 x fewer bytes per tile at 64 channels (23 + 72 KB instead of 216), and the patch of chunk c + 1 streams in under
// the nine steps of chunk c.  Forward (source row = result row - 1 + dy) and input gradient (result row + 1 - dy: the caller's transposed
// filter, as above).  64-column tiles, weight tiles on a three-stage ring, two workgroups per CU.  (Measured and dropped: a four-stage ring — three
// steps ahead, patch buffers trimmed to 180 rows so that two workgroups still fit a CU — 639 vs 619 us over the step's 26 launches: the steps do not
// wait for their weight tiles.)
constexpr int TH3 = 8, TW3 = 16, PW3 = TW3 + 2, PROWS3 = (TH3 + 2) * PW3, PPIECES3 = 24;       // 180 patch rows in 24 pieces of 8 rows (the last 12 rows: zeros)
constexpr int PATCH3 = PPIECES3 * 1024;

__global__ __launch_bounds__(256, 2) void igemm3x3_bf16(IgArgs a)
{
  constexpr int BN = 64, NBJ = 2, BT = BN * BKB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char *const patch = smem;                      // [2][PATCH3]
  unsigned char *const bring = smem + 2 * PATCH3;        // [3][BT]
  const float *const sb = reinterpret_cast<const float *>(smem + 2 * PATCH3 + 3 * BT);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lb = xcd_chunk(blockIdx.x, gridDim.x);
  const int tile = lb / a.splits, split = lb - tile * a.splits;     // split-K over the channel chunks (few blocks under a long contraction: res5)
  const int mt = tile / a.ntn, n0 = (tile - mt * a.ntn) * BN;
  const int cbeg = split * a.kt_per, cend = min(a.cch, cbeg + a.kt_per);   // (kt_per = channel chunks per split here)
  const int txn = (a.Wo + TW3 - 1) / TW3, tyn = (a.Ho + TH3 - 1) / TH3;
  const int b = mt / (tyn * txn), r_ = mt - b * (tyn * txn), ty0 = (r_ / txn) * TH3, tx0 = (r_ - (r_ / txn) * txn) * TW3;
  const int lrow = lane >> 3, lch = lane & 7;
  const bf16_t *zline = reinterpret_cast<const bf16_t *>(g_zero_line);
  float sbv = 0.f;
  if (t < 2 * BN) {
    const float *src = t < BN ? a.scale : a.bias;
    if (t >= BN && a.bias && a.bias_bf16) sbv = __uint_as_float((unsigned)reinterpret_cast<const bf16_t *>(a.bias)[n0 + t - BN] << 16);
    else sbv = src ? src[n0 + (t < BN ? t : t - BN)] : (t < BN ? 1.f : 0.f);
  }
  // ---- the six patch rows this thread fetches per channel chunk (piece wave * 6 + j: rows 8 piece + lane / 8, 16-byte chunk lane & 7)
  const bf16_t *pp[6];
  bool pok[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int pr = (wave * 6 + j) * 8 + lrow, py = pr / PW3, px = pr - py * PW3;
    const int sy = ty0 - 1 + py, sx = tx0 - 1 + px;
    pok[j] = pr < PROWS3 && (unsigned)sy < (unsigned)a.Hs && (unsigned)sx < (unsigned)a.Ws;
    pp[j] = pok[j] ? a.S + ((int64_t)(b * a.Hs + sy) * a.Ws + sx) * a.Cs + (lch ^ ((pr >> 1) & 7)) * 8 : zline + lch * 8;
  }
  const bf16_t *wb[NBJ];
  const int ldw = a.KT * 64;
#pragma unroll
  for (int j = 0; j < NBJ; ++j) {
    const int row = (wave * NBJ + j) * 8 + lrow;
    wb[j] = a.Wf + (int64_t)(n0 + row) * ldw + (lch ^ ((row >> 1) & 7)) * 8;
  }
  auto issue_patch = [&](int c, int buf) {
#pragma unroll
    for (int j = 0; j < 6; ++j)
      __builtin_amdgcn_global_load_lds((glb_ptr)(pok[j] ? pp[j] + c * 64 : pp[j]), (lds_ptr)(patch + buf * PATCH3 + (wave * 6 + j) * 1024), 16, 0, 0);
  };
  auto issue_b = [&](int kt, int buf) {                    // kt: the filter's K index tap * cch + c of the step's (tap, chunk)
#pragma unroll
    for (int j = 0; j < NBJ; ++j)
      __builtin_amdgcn_global_load_lds((glb_ptr)(wb[j] + (int64_t)kt * 64), (lds_ptr)(bring + buf * BT + (wave * NBJ + j) * 1024), 16, 0, 0);
  };
  f32x16 acc[2][1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][0][e] = 0.f;
  const int fr = lane & 31, kh = lane >> 5, sw = (fr >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = ((ks * 2 + kh) ^ sw) * 16;
  const int pty = 2 * wave + (fr >> 4), ptx = fr & 15;     // this lane's result pixel inside the block
  const int pr0 = pty * PW3 + ptx;                         // this lane's patch row under tap (0, 0)
  auto compute = [&](const unsigned char *Pc, int off, int bbuf) __attribute__((always_inline)) {   // off: the tap's patch-row offset
    const int pr = pr0 + off, psw = (pr >> 1) & 7;
    const unsigned char *As = Pc + pr * BKB, *Bs = bring + bbuf * BT;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      hwbf16x8 wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[i] = *reinterpret_cast<const hwbf16x8 *>(Bs + (i * 32 + fr) * BKB + foff[ks]);
      const hwbf16x8 af = *reinterpret_cast<const hwbf16x8 *>(As + ((ks * 2 + kh) ^ psw) * 16);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], af, acc[i][0], 0, 0, 0);
    }
  };
  if (t < 2 * BN) reinterpret_cast<float *>(smem + 2 * PATCH3 + 3 * BT)[t] = sbv;
  // ---- the step loop: weight tiles two steps ahead on the ring; chunk c + 1's patch is issued in the slot of chunk c's first step.
  // Counted waits (loads complete in issue order): behind step s's weight tile the queue holds step s + 1's tile (2 pieces per wavefront)
  // and, when the previous slot also issued a patch, its 6 pieces.
  // The nine taps of a chunk are UNROLLED with the tap a compile-time constant (round 6, second pass): as one loop over s = 9 c + tap the
  // step was 749 instructions around its 8 MFMAs — s / 9 and s % 9 three times over, the tap's (dy, dx), the ring position as a rotating
  // variable with three copies of the issue code — i.e. bound by instruction issue (2 600 cycles per step against 256 of matrix work).  Nine
  // steps are a multiple of the three ring stages, so the ring position is tap % 3.
  const int ncl = cend - cbeg;                             // (patch buffers alternate with the LOCAL chunk index)
  issue_patch(cbeg, 0);
  issue_b(cbeg, 0);
  issue_b(a.cch + cbeg, 1);
  for (int cl = 0; cl < ncl; ++cl) {
    const bool more = cl + 1 < ncl;
    const unsigned char *Pc = patch + (cl & 1) * PATCH3;
    const int c = cbeg + cl;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap == 8) {
        if (more) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (tap == 1) {                               // the slot before issued chunk c + 1's patch (6 pieces) in front of step 2's weights
        if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                        // this step (and its chunk's patch) landed for everyone; everyone is done reading the previous step's stage
      if (tap == 0 && more) issue_patch(c + 1, (cl + 1) & 1);
      const int tap2 = (tap + 2) % 9, carry = (tap + 2) / 9;                   // the step two ahead
      if (carry == 0 || more) issue_b(tap2 * a.cch + c + carry, (tap + 2) % 3);
      const int dy = tap / 3, dx = tap - 3 * dy;
      compute(Pc, a.dgrad ? (2 - dy) * PW3 + (2 - dx) : dy * PW3 + dx, tap % 3);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (a.splits > 1 && !splitk_seam<1>(a, acc, tile, split, t, smem)) return;
  const int y = ty0 + pty, x = tx0 + ptx;
  const bool rowok = y < a.Ho && x < a.Wo;
  const int m = rowok ? (b * a.Ho + y) * a.Wo + x : 0;
  constexpr int j = 0, wn = 0;
  constexpr bool EPI_PIPE = true;
#include "igemm_epilogue.inc"
}

struct Plan {
  IgArgs a;
  int bn, nst;
  int64_t slab_bytes;
  bool patch3;                                           // 3 x 3 stride 1: igemm3x3_bf16
};

int make_plan(const PdIgemm *p, Plan &pl)
{
  if (!p || !p->src || !p->w || !p->out) return pd_set_error(PD_ERR_INVALID_ARG, "pd_igemm_bf16: null pointer");
  if (!pd_igemm_bf16_supported(p->cs, p->n, p->k, p->stride, p->pad))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_igemm_bf16: unsupported geometry cs=%d n=%d k=%d stride=%d pad=%d", p->cs, p->n, p->k, p->stride, p->pad);
  if (p->batch <= 0 || p->ho <= 0 || p->wo <= 0 || p->hs <= 0 || p->ws <= 0)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_igemm_bf16: empty grid");
  if (p->res && p->res_mode == PD_IG_RES_UP2 && ((p->ho | p->wo) & 1))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_igemm_bf16: PD_IG_RES_UP2 needs an even result grid");
  IgArgs &a = pl.a;
  a.S = (const bf16_t *)p->src; a.Wf = (const bf16_t *)p->w; a.scale = p->scale; a.bias = p->bias;
  a.res = (const bf16_t *)p->res; a.res2 = (const bf16_t *)p->res2; a.gate = (const bf16_t *)p->gate; a.Y = (bf16_t *)p->out; a.Ypre = (bf16_t *)p->out_pre;
  a.slabs = nullptr; a.tickets = nullptr;
  const int64_t M = (int64_t)p->batch * p->ho * p->wo;
  if (M > 0x7fffffff / 2 || (int64_t)p->batch * p->hs * p->ws > 0x7fffffff / 2) return pd_set_error(PD_ERR_INVALID_ARG, "pd_igemm_bf16: grid too large");
  a.M = (int)M; a.N = p->n; a.Cs = p->cs; a.kw = p->k; a.cch = p->cs / 64; a.KT = p->k * p->k * a.cch;
  a.Hs = p->hs; a.Ws = p->ws; a.Ho = p->ho; a.Wo = p->wo; a.stride = p->stride; a.pad = p->pad; a.dgrad = p->dgrad;
  a.act = p->act; a.gate_mode = p->gate_mode; a.res_mode = p->res_mode; a.bias_bf16 = p->bias_bf16;
  a.slabN = p->out_col_slab;
  if (a.slabN && ((a.slabN & 127) || a.slabN < 0 || p->n % a.slabN || p->res || p->res2 || p->gate || p->out_pre))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_igemm_bf16: out_col_slab=%d (a multiple of 128 dividing n=%d; no res / res2 / gate / out_pre)", a.slabN, p->n);
  // tile width: 128 columns, or 64 when n is not a multiple of 128 or when 128-wide tiles would leave most workgroup slots empty
  const int mt = (a.M + BM - 1) / BM;
  const bool plain = p->k == 1 && p->stride == 1 && !p->dgrad && p->hs == p->ho && p->ws == p->wo;   // (the instantiation launch() picks)
  pl.bn = (p->n % 128 == 0) ? 128 : 64;
  // 64-wide tiles when 128-wide ones would leave most workgroup slots empty.  Plain-rows problems (Linears, 1 x 1 convolutions) flip at ~400
  // tiles, not 512 (second sweep, profiles/r04_igemm_sweep_v2.txt: 4 608 x 1 536 <- 1 536, 432 tiles: 31.7 us 128-wide, 38.4 64-wide;
  // <- 6 144: 108 vs 136; 10 368 x 512 <- 2 048, 324 tiles: 38.2 vs 33.4)
  if (pl.bn == 128 && (mt * (p->n / 128) < (plain ? 400 : 512) || (p->n < 384 && a.KT < 8))) pl.bn = 64;
  if (g_ig_bn == 64 || (g_ig_bn == 128 && p->n % 128 == 0)) pl.bn = g_ig_bn;
  a.ntn = p->n / pl.bn;
  a.ntiles = mt * a.ntn;
  // stages: one (32 KB, four workgroups per CU) for short contractions, two (one step of prefetch, two per CU) for long ones
  // (tools/bench_igemm.py sweep: 128-wide tiles run best on one stage with four workgroups per CU up to K = 4096; 64-wide tiles with one
  // step of prefetch from K = 512 — and, for the gathered (3 x 3 / strided) problems only, with the three-stage ring from K = 1024: plain
  // rows measured 33.4 vs 41.2 us (10 368 x 512 <- 2 048) and 40.7 vs 42.0 (2 592 x 1 024 <- 4 096) in favour of two stages)
  pl.nst = pl.bn == 128 ? (a.KT >= 64 ? 2 : 1) : (a.KT >= 16 && !plain ? 3 : a.KT >= 8 ? 2 : 1);
  if (g_ig_nst >= 1 && g_ig_nst <= 3) pl.nst = g_ig_nst;
  // split-K: only when the tiles alone leave most workgroup slots empty AND the contraction is long (every split costs a
  // 32-64 KB slab written and read back)
  // (round 6: in ISOLATION res5's 1 x 1 over 2 048 channels — 128 tiles x 32 K-steps — runs 18.3 us unsplit and 30.0 us in three splits, but
  // the step does not move with the rule changed to "KT >= 48" (19.04 vs 19.04 ms, two same-box pairs): inside the step the 2 MB of weights
  // come from HBM and 384 workgroups pull them faster than 128.  Rule kept.)
  int splits = 1;
  if (a.ntiles <= 128 && a.KT >= 16) {
    splits = (384 + a.ntiles - 1) / a.ntiles;
    if (splits > a.KT / 8) splits = a.KT / 8;
    if (splits > 16) splits = 16;
    if (splits < 1) splits = 1;
  }
  if (g_ig_splits > 0 && a.ntiles <= 4096) splits = g_ig_splits > a.KT ? a.KT : g_ig_splits;
  a.kt_per = (a.KT + splits - 1) / splits;
  a.splits = (a.KT + a.kt_per - 1) / a.kt_per;
  pl.slab_bytes = a.splits > 1 ? (int64_t)a.ntiles * a.splits * BM * pl.bn * 4 : 0;
  // stride-2 3 x 3 input gradient on an even result grid: parity classes (no split-K; pd_debug_set("ig_pcls", 0) keeps the nine-tap walk)
  a.pcls = 0; a.tpc = 0;
  if (p->dgrad && p->k == 3 && p->stride == 2 && !((p->ho | p->wo) & 1) && g_ig_pcls != 0) {
    a.pcls = 1;
    a.tpc = (a.M / 4 + BM - 1) / BM;
    pl.bn = (p->n % 128 == 0 && a.tpc * 4 * (p->n / 128) >= 384) ? 128 : 64;
    if (g_ig_bn == 64 || (g_ig_bn == 128 && p->n % 128 == 0)) pl.bn = g_ig_bn;
    a.ntn = p->n / pl.bn;
    a.ntiles = 4 * a.tpc * a.ntn;
    pl.nst = pl.bn == 128 ? 2 : 3;
    if (g_ig_nst >= 1 && g_ig_nst <= 3) pl.nst = g_ig_nst;
    a.splits = 1; a.kt_per = a.KT;
    pl.slab_bytes = 0;
  }
  // 3 x 3, stride 1, same-size grids: the patch form, when its 8 x 16 blocks fill the chip without split-K (res5's 16 blocks x 8 column tiles keep
  // the gathered kernel's split-K); pd_debug_set("ig_patch", 0) keeps the gathered kernel, 2 forces the patch form wherever it is defined
  pl.patch3 = false;
  if (p->k == 3 && p->stride == 1 && p->pad == 1 && p->hs == p->ho && p->ws == p->wo && g_ig_patch != 0) {
    const int blocks = p->batch * ((p->ho + TH3 - 1) / TH3) * ((p->wo + TW3 - 1) / TW3) * (p->n / 64);
    if (blocks <= 4096) {                                  // (ticket array of the split-K seam)
      pl.patch3 = true;
      pl.bn = 64; pl.nst = 3;
      a.ntn = p->n / 64;
      a.ntiles = blocks;
      // few blocks under many channel chunks (res5: 128 blocks x 8 chunks): the chunks are split over workgroups, fp32 slabs + last arriver
      int sp = 1;
      if (blocks < 192 && a.cch >= 4) sp = blocks <= 64 ? 4 : blocks <= 128 ? 3 : 2;
      if (g_ig_splits > 0) sp = g_ig_splits;
      if (sp > a.cch) sp = a.cch;
      a.kt_per = (a.cch + sp - 1) / sp;                    // channel chunks per split
      a.splits = (a.cch + a.kt_per - 1) / a.kt_per;
      pl.slab_bytes = a.splits > 1 ? (int64_t)a.ntiles * a.splits * BM * 64 * 4 : 0;
    }
  }
  return PD_OK;
}

constexpr int64_t TICKET_BYTES = 4096 * 4;

template <int BN, int NST, bool P1>
int launch1(const Plan &pl, hipStream_t stream)
{
  constexpr size_t lds = (size_t)NST * (BM + BN) * BKB + 2 * BN * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)igemm_bf16<BN, NST, P1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const int64_t nblocks = (int64_t)pl.a.ntiles * pl.a.splits;
  hipLaunchKernelGGL((igemm_bf16<BN, NST, P1>), dim3((unsigned)nblocks), dim3(256), lds, stream, pl.a);
  return pd_check_launch("pd_igemm_bf16");
}

template <int BN, int NST>
int launch(const Plan &pl, hipStream_t stream)
{
  // plain rows: a Linear / 1 x 1 stride-1 convolution (no taps, no pixel decoding, no padding)
  const bool p1 = pl.a.kw == 1 && pl.a.stride == 1 && !pl.a.dgrad && pl.a.Hs == pl.a.Ho && pl.a.Ws == pl.a.Wo;
  return p1 ? launch1<BN, NST, true>(pl, stream) : launch1<BN, NST, false>(pl, stream);
}

int launch_plan(const Plan &pl, hipStream_t st)
{
  if (pl.patch3) {
    constexpr size_t lds = (size_t)2 * PATCH3 + 3 * 64 * BKB + 2 * 64 * sizeof(float);
    static bool attr3 = false;
    if (!attr3) { (void)hipFuncSetAttribute((const void *)igemm3x3_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr3 = true; }
    hipLaunchKernelGGL(igemm3x3_bf16, dim3((unsigned)(pl.a.ntiles * pl.a.splits)), dim3(256), lds, st, pl.a);
    return pd_check_launch("pd_igemm_bf16 (3 x 3 patch form)");
  }
  if (pl.bn == 128) return pl.nst == 3 ? launch<128, 3>(pl, st) : pl.nst == 2 ? launch<128, 2>(pl, st) : launch<128, 1>(pl, st);
  return pl.nst == 3 ? launch<64, 3>(pl, st) : pl.nst == 2 ? launch<64, 2>(pl, st) : launch<64, 1>(pl, st);
}
}  // namespace

extern "C" int pd_igemm_bf16(const PdIgemm *p, void *workspace, int64_t workspace_bytes, void *stream);

// development: `iters` back-to-back launches of one problem between two events -> average microseconds per launch
extern "C" int pd_igemm_bf16_time(const PdIgemm *p, void *workspace, int64_t workspace_bytes, int iters, float *us, void *stream)
{
  if (!us || iters <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_igemm_bf16_time: bad arguments");
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return pd_set_error(PD_ERR_LAUNCH, "pd_igemm_bf16_time: events");
  int rc = PD_OK;
  for (int i = 0; i < 3 && rc == PD_OK; ++i) rc = pd_igemm_bf16(p, workspace, workspace_bytes, stream);
  (void)hipEventRecord(e0, (hipStream_t)stream);
  for (int i = 0; i < iters && rc == PD_OK; ++i) rc = pd_igemm_bf16(p, workspace, workspace_bytes, stream);
  (void)hipEventRecord(e1, (hipStream_t)stream);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *us = ms * 1e3f / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return rc;
}

extern "C" int pd_igemm_bf16_supported(int cs, int n, int k, int stride, int pad)
{
  return cs > 0 && n > 0 && cs % 64 == 0 && n % 64 == 0 && (k == 1 || k == 3) && (stride == 1 || stride == 2) && pad == k / 2;
}

extern "C" int64_t pd_igemm_bf16_workspace_bytes(const PdIgemm *p)
{
  Plan pl;
  if (make_plan(p, pl) != PD_OK) return -1;
  return pl.slab_bytes ? pl.slab_bytes + TICKET_BYTES : 0;
}

// ------------------------------------------------------------------------------------------------ filter transposes
// dst[ci][tap][co] = src[co][tap][ci] for MANY filters in one launch (the input-gradient operands of a whole backbone, rebuilt
// from the updated weights once per step): 64 x 64 bf16 blocks through LDS, 16-byte loads and stores.
struct TrProblem { const bf16_t *src; bf16_t *dst; const float *scale; int co, taps, ci, first_block; };

__global__ __launch_bounds__(256) void filter_transpose_grouped(const TrProblem *__restrict__ tab, int count)
{
  __shared__ bf16_t tile[64][72];
  int lo = 0, hi = count - 1;
  const int bid = blockIdx.x;
  while (lo < hi) {                                       // the problem this block belongs to (first_block ascending)
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].first_block <= bid) lo = mid; else hi = mid - 1;
  }
  TrProblem p = tab[lo];
  p.src = pd_as_global(p.src); p.dst = pd_as_global(p.dst); p.scale = pd_as_global(p.scale);      // (pd_common.h: table pointers would be FLAT)
  const int local = bid - p.first_block;
  const int nci = p.ci / 64, nco = p.co / 64;
  const int tap = local / (nci * nco), r = local - tap * nci * nco, bco = r / nci, bci = r - bco * nci;
  const int t = threadIdx.x, row = t >> 3, c8 = (t & 7) * 8;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int co = bco * 64 + row + 32 * j;
    uint4 v = *reinterpret_cast<const uint4 *>(p.src + ((int64_t)co * p.taps + tap) * p.ci + bci * 64 + c8);
    if (p.scale) {
      const float sc = p.scale[co];
      v.x = pk_bf16(bf_lo(v.x) * sc, bf_hi(v.x) * sc); v.y = pk_bf16(bf_lo(v.y) * sc, bf_hi(v.y) * sc);
      v.z = pk_bf16(bf_lo(v.z) * sc, bf_hi(v.z) * sc); v.w = pk_bf16(bf_lo(v.w) * sc, bf_hi(v.w) * sc);
    }
    *reinterpret_cast<uint4 *>(&tile[row + 32 * j][c8]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ci = row + 32 * j;
    unsigned short v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[c8 + e][ci];
    uint4 o;
    o.x = v[0] | ((unsigned)v[1] << 16); o.y = v[2] | ((unsigned)v[3] << 16); o.z = v[4] | ((unsigned)v[5] << 16); o.w = v[6] | ((unsigned)v[7] << 16);
    *reinterpret_cast<uint4 *>(p.dst + ((int64_t)(bci * 64 + ci) * p.taps + tap) * p.co + bco * 64 + c8) = o;
  }
}

extern "C" int64_t pd_filter_transpose_table_bytes(int count) { return (int64_t)count * sizeof(TrProblem); }

extern "C" int pd_filter_transpose_grouped(const PdFilterTranspose *descs, int count, void *table_host_pinned, void *table_device, void *stream)
{
  if (count <= 0) return PD_OK;
  if (!descs || !table_host_pinned || !table_device) return pd_set_error(PD_ERR_INVALID_ARG, "pd_filter_transpose_grouped: null pointer");
  TrProblem *h = reinterpret_cast<TrProblem *>(table_host_pinned);
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const PdFilterTranspose &d = descs[i];
    if (d.co % 64 || d.ci % 64 || d.taps <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_filter_transpose_grouped: co %% 64 == ci %% 64 == 0 required");
    h[i] = TrProblem{(const bf16_t *)d.src, (bf16_t *)d.dst, d.scale, d.co, d.taps, d.ci, blocks};
    blocks += d.taps * (d.co / 64) * (d.ci / 64);
  }
  if (hipMemcpyAsync(table_device, h, (size_t)count * sizeof(TrProblem), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess)
    return pd_set_error(PD_ERR_LAUNCH, "pd_filter_transpose_grouped: table upload failed");
  hipLaunchKernelGGL(filter_transpose_grouped, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const TrProblem *)table_device, count);
  return pd_check_launch("pd_filter_transpose_grouped");
}

// A whole sequence of problems, enqueued back to back on one stream from C++ (one call from the host language instead of one per
// layer: the R50 body is 52 of them per direction).  They run in order, so they share ONE workspace.
extern "C" int pd_igemm_bf16_seq(const PdIgemm *list, int count, void *workspace, int64_t workspace_bytes, void *stream)
{
  for (int i = 0; i < count; ++i) {
    const int rc = pd_igemm_bf16(list + i, workspace, workspace_bytes, stream);
    if (rc != PD_OK) return rc;
  }
  return PD_OK;
}

extern "C" int64_t pd_igemm_bf16_seq_workspace_bytes(const PdIgemm *list, int count)
{
  int64_t need = 0;
  for (int i = 0; i < count; ++i) {
    const int64_t b = pd_igemm_bf16_workspace_bytes(list + i);
    if (b < 0) return -1;
    need = b > need ? b : need;
  }
  return need;
}

extern "C" int pd_igemm_bf16(const PdIgemm *p, void *workspace, int64_t workspace_bytes, void *stream)
{
  Plan pl;
  const int rc = make_plan(p, pl);
  if (rc != PD_OK) return rc;
  if (pl.slab_bytes) {
    if (!workspace || workspace_bytes < pl.slab_bytes + TICKET_BYTES || pl.a.ntiles > 4096)
      return pd_set_error(PD_ERR_INVALID_ARG, "pd_igemm_bf16: workspace of %lld bytes needed (%lld given)", (long long)(pl.slab_bytes + TICKET_BYTES),
                          (long long)workspace_bytes);
    pl.a.tickets = reinterpret_cast<unsigned *>(workspace);
    pl.a.slabs = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(workspace) + TICKET_BYTES);
  }
  return launch_plan(pl, (hipStream_t)stream);
}
