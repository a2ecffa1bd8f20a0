// bf16 NHWC convolutions of the ResNet backbone as implicit GEMMs on the gfx950 matrix cores (C-ABI: include/pd_conv.h).
//
//   rows    = output pixels (all images), 128 per workgroup          columns = output channels, 128 or 64 per workgroup
//   K       = taps x source channels, walked 64 at a time: the source channel count is a multiple of 64, so one K-step lies
//             inside ONE filter tap and the A tile of a step is 128 gathered 128-byte pixel rows (zeros outside the image) —
//             nothing is unfolded, a 1 x 1 convolution is the plain GEMM on the NHWC rows
//   filter  = [columns][taps][source channels] (the channels-last filter the flat parameter buffer already holds), K contiguous
//
// One kernel serves both directions:
//   forward   source = x,  sy = oy * stride + dy - pad;   epilogue y = act(acc * scale[c] + bias[c] (+ residual))  (frozen BN)
//   dgrad     source = dz, the GEMM rows are INPUT pixels, sy = (iy + pad - dy) / stride where divisible (a stride-2 gather
//             simply finds no source for 3 of 4 (pixel, tap) pairs); filter = the [ci][taps][co] transpose; epilogue + addend
// 256 threads = 4 wavefronts, each 64 x 64 (32 x 64 for 64 columns) of v_mfma_f32_32x32x16_bf16 tiles; the filter is the
// MFMA "X" operand and the pixels the "Y" operand, so a lane ends up with 4 CONSECUTIVE channels of one pixel per register
// quad and the tile goes through LDS once (fp32) to leave as 16-byte row-contiguous bf16 stores next to a 16-byte residual
// load.  Two LDS stages of 64-deep tiles (72-element pitch: 16-byte fragment reads of 16 rows hit 64 distinct banks); the
// next tile's global loads are issued before the MFMAs of the current one and written to the other stage after them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mfma_bf16.h"
#include "pd_common.h"
#include "xcd.h"
#include "pd_conv.h"
#include "pd_msda.h"

namespace {
using namespace pdmfma;

typedef __bf16 hwbf16x8 __attribute__((ext_vector_type(8)));
constexpr int CBM = 128, CBK = 64, CP = 72;

__device__ __forceinline__ int xcd_chunk(int bid, int nb) { return (nb & 7) == 0 ? (bid & 7) * (nb >> 3) + (bid >> 3) : bid; }
__device__ __forceinline__ hwbf16x8 frag16(const bf16_t *p) { return *reinterpret_cast<const hwbf16x8 *>(p); }

struct ConvGeom {
  int M, N;                   // GEMM rows (pixels of the result grid, all images), columns
  int Cs, kw, ntaps;          // source channels, filter width, taps
  int Hs, Ws;                 // source grid
  int Ho, Wo;                 // result grid (per image)
  int stride, pad;
};

template <int BN, bool DGRAD>
__global__ __launch_bounds__(256, 2) void conv_igemm_bf16(const bf16_t *__restrict__ S, const bf16_t *__restrict__ Wf,
                                                           const float *__restrict__ scale, const float *__restrict__ bias,
                                                           const bf16_t *__restrict__ res, bf16_t *__restrict__ Y, ConvGeom g, int relu)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t *smem = reinterpret_cast<bf16_t *>(smem_raw);
  constexpr int STAGE = (CBM + BN) * CP;                 // bf16 elements per stage: A rows then B rows
  constexpr int WTM = BN == 128 ? 64 : 32;               // pixel rows per wavefront
  constexpr int MI = WTM / 32;
  constexpr int NB = BN / 32;                            // B pieces per thread
  const int ntn = g.N / BN;
  const int lb = xcd_chunk(blockIdx.x, gridDim.x);
  const int m0 = (lb / ntn) * CBM, n0 = (lb % ntn) * BN;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = BN == 128 ? (wave >> 1) * 64 : wave * 32, wn = BN == 128 ? (wave & 1) * 64 : 0;
  const int prow = t >> 3, pc = (t & 7) * 8;             // this thread's piece: rows prow + 32 j, 8 elements at pc

  // gather bookkeeping of the thread's four A rows
  int sbase[4], y0[4], x0[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + prow + 32 * j;
    if (m < g.M) {
      const int hw = g.Ho * g.Wo, b = m / hw, rem = m - b * hw, oy = rem / g.Wo, ox = rem - oy * g.Wo;
      sbase[j] = b * g.Hs * g.Ws;
      y0[j] = DGRAD ? oy + g.pad : oy * g.stride - g.pad;
      x0[j] = DGRAD ? ox + g.pad : ox * g.stride - g.pad;
    } else {
      sbase[j] = -1; y0[j] = 0; x0[j] = 0;
    }
  }
  const int cchunks = g.Cs / CBK;
  const int KT = g.ntaps * cchunks;
  const int ldw = g.ntaps * g.Cs;

  uint4 ra[4], rb[NB];
  auto gload = [&](int kt) {
    const int tap = kt / cchunks, c0 = (kt - tap * cchunks) * CBK;
    const int dy = tap / g.kw, dx = tap - dy * g.kw;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int sy, sx;
      bool ok = sbase[j] >= 0;
      if (DGRAD) {
        const int ty = y0[j] - dy, tx = x0[j] - dx;
        ok = ok && ty >= 0 && tx >= 0;
        if (g.stride == 2) { ok = ok && ((ty | tx) & 1) == 0; sy = ty >> 1; sx = tx >> 1; }
        else { sy = ty; sx = tx; }
        ok = ok && sy < g.Hs && sx < g.Ws;
      } else {
        sy = y0[j] + dy; sx = x0[j] + dx;
        ok = ok && sy >= 0 && sy < g.Hs && sx >= 0 && sx < g.Ws;
      }
      ra[j] = ok ? *reinterpret_cast<const uint4 *>(S + ((int64_t)(sbase[j] + sy * g.Ws + sx)) * g.Cs + c0 + pc) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j)
      rb[j] = *reinterpret_cast<const uint4 *>(Wf + (int64_t)(n0 + prow + 32 * j) * ldw + (int64_t)tap * g.Cs + c0 + pc);
  };
  auto lstore = [&](int buf) {
    bf16_t *As = smem + buf * STAGE, *Bs = As + CBM * CP;
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4 *>(As + (prow + 32 * j) * CP + pc) = ra[j];
#pragma unroll
    for (int j = 0; j < NB; ++j) *reinterpret_cast<uint4 *>(Bs + (prow + 32 * j) * CP + pc) = rb[j];
  };

  f32x16 acc[2][MI];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int fr = lane & 31, fk = (lane >> 5) * 8;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) gload(kt + 1);
    const bf16_t *As = smem + buf * STAGE, *Bs = As + CBM * CP;
#pragma unroll
    for (int ks = 0; ks < CBK / 16; ++ks) {
      hwbf16x8 wf[2], af[MI];
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[i] = frag16(Bs + (wn + i * 32 + fr) * CP + ks * 16 + fk);
#pragma unroll
      for (int j = 0; j < MI; ++j) af[j] = frag16(As + (wm + j * 32 + fr) * CP + ks * 16 + fk);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < KT) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: fp32 tile through LDS, then row-contiguous 16-byte stores
  constexpr int PC = BN + 4;                             // fp32 pitch
  float *Cs = reinterpret_cast<float *>(smem_raw);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const int row = wm + j * 32 + (lane & 31);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = wn + i * 32 + 8 * q + 4 * (lane >> 5);
        *reinterpret_cast<float4 *>(Cs + row * PC + col) = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
      }
    }
  __syncthreads();
  constexpr int PPR = BN / 8;                            // 8-channel pieces per row
#pragma unroll
  for (int it = 0; it < CBM * PPR / 256; ++it) {
    const int q = t + 256 * it, row = q / PPR, cp = (q - row * PPR) * 8;
    const int m = m0 + row;
    if (m >= g.M) continue;
    const float4 v0 = *reinterpret_cast<const float4 *>(Cs + row * PC + cp), v1 = *reinterpret_cast<const float4 *>(Cs + row * PC + cp + 4);
    float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    const int c = n0 + cp;
    if (scale) {
      const float4 s0 = *reinterpret_cast<const float4 *>(scale + c), s1 = *reinterpret_cast<const float4 *>(scale + c + 4);
      const float s[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= s[e];
    }
    if (bias) {
      const float4 b0 = *reinterpret_cast<const float4 *>(bias + c), b1 = *reinterpret_cast<const float4 *>(bias + c + 4);
      const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += b[e];
    }
    const int64_t off = (int64_t)m * g.N + c;
    if (res) {
      const uint4 r = *reinterpret_cast<const uint4 *>(res + off);
      v[0] += bf_lo(r.x); v[1] += bf_hi(r.x); v[2] += bf_lo(r.y); v[3] += bf_hi(r.y);
      v[4] += bf_lo(r.z); v[5] += bf_hi(r.z); v[6] += bf_lo(r.w); v[7] += bf_hi(r.w);
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    *reinterpret_cast<uint4 *>(Y + off) = make_uint4(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7]));
  }
}

template <int BN, bool DGRAD>
int launch_conv(const void *S, const void *Wf, const float *scale, const float *bias, const void *res, void *Y, const ConvGeom &g, int relu,
                hipStream_t stream)
{
  constexpr size_t lds_main = 2 * (size_t)(CBM + BN) * CP * sizeof(bf16_t), lds_epi = (size_t)CBM * (BN + 4) * sizeof(float);
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)conv_igemm_bf16<BN, DGRAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const int64_t nblocks = (int64_t)((g.M + CBM - 1) / CBM) * (g.N / BN);
  hipLaunchKernelGGL((conv_igemm_bf16<BN, DGRAD>), dim3((unsigned)nblocks), dim3(256), lds, stream, (const bf16_t *)S, (const bf16_t *)Wf, scale,
                     bias, (const bf16_t *)res, (bf16_t *)Y, g, relu);
  return pd_check_launch(DGRAD ? "pd_conv_bf16_dgrad" : "pd_conv_bf16_fwd");
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[co][tap][ci] = sum over output pixels m of dz[m][co] * x[pixel(m, tap)][ci]: the contraction runs over the ROWS of both
// operands, so the tiles are staged as they lie in memory ([pixel][channel], 16-byte loads straight into LDS, no arithmetic) and
// transposed on the way out by ds_read_b64_tr_b16 (lane mapping: tools/probes/tr_read_probe.hip; same scheme as
// gemm_wgrad_f32x3_tr in gemm_x3.hip).  128 x 128 output tile per workgroup, 32 pixels per stage (two 16-deep MFMA steps), the
// pixel range split over workgroups whose fp32 partial tiles go to a workspace in register order; conv_wgrad_reduce sums them
// and writes bf16.  Column kk of the GEMM = (tap, ci): a thread's 8 columns lie inside one tap (ci % 8 == 0), so a 128-wide tile
// may span several taps (ci = 64) — each thread gathers with its own tap offset.
constexpr int WTW = 32, WTP = 160;                       // pixels per stage, bf16 elements per LDS row (320 B: see gemm_x3.hip)
typedef short v4s16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ hwbf16x8 frag_tr(const bf16_t *p)
{
  typedef __attribute__((address_space(3))) v4s16 *lp;
  union { v4s16 h[2]; hwbf16x8 v; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p));
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p + 4 * WTP));
  return u.v;
}

struct WgradGeom {
  int M, N, K;                // output pixels (all images), co, taps * ci
  int Ci, kw;
  int Hi, Wi, Ho, Wo;
  int stride, pad;
  int tiles_k, tiles, m_chunk;
};

// TN x TK output tile = (2 WN) x (2 WK), 4 wavefronts (2 x 2) of WN x WK each; 64-wide tiles for the 64-channel layers of res2 (half
// of a 128-wide tile would be zeros there — and the fp32 partial tiles are this kernel's second-largest traffic).
// P1: a 1 x 1 stride-1 problem (36 of R50's 52 convolutions, the decoder's key / value projections): the X row of output pixel m IS row m — no
// tap, no (image, y, x) bookkeeping, no bounds but the slice's end.  The general gather spends ~40 vector instructions per 16-byte load on
// them; with 8 MFMAs per stage the loop was 260-350 instructions per stage, i.e. bound by instruction issue (SQ counters: matrix pipe busy
// 12 %, round 6).
template <int WN, int WK, bool P1>
__device__ __forceinline__ void wgrad_body(const bf16_t *__restrict__ dZ, const bf16_t *__restrict__ X, float *__restrict__ ws,
                                           const WgradGeom &g, int bid, float *__restrict__ dB = nullptr)
{
  constexpr int TN = 2 * WN, TK = 2 * WK, NI = WN / 32, KJ = WK / 32;
  constexpr int PN = TN + 32, PK = TK + 32;              // LDS row pitches: 160 or 96 bf16 = 320 / 192 B, both = 16 banks mod 64
  constexpr int LY = 256 / (TN / 8), LX = 256 / (TK / 8);   // pixels one pass of the 256 threads stages (16 or 32)
  __shared__ __attribute__((aligned(16))) bf16_t SY[2][WTW][PN];
  __shared__ __attribute__((aligned(16))) bf16_t SX[2][WTW][PK];
  const int tile = bid % g.tiles, split = bid / g.tiles;
  const int n0 = (tile / g.tiles_k) * TN, k0 = (tile % g.tiles_k) * TK;
  const int mb = split * g.m_chunk, me = min(g.M, mb + g.m_chunk);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wn = (wave >> 1) * WN, wk = (wave & 1) * WK;
  const int yr = t / (TN / 8), yc = (t % (TN / 8)) * 8;  // dz staging: pixels yr + LY j, columns yc .. yc+7
  const int xr = t / (TK / 8), xcl = (t % (TK / 8)) * 8;
  const bool ycol_ok = n0 + yc < g.N, xcol_ok = k0 + xcl < g.K;
  int tdy = 0, tdx = 0, xc = 0;                          // this thread's X columns: one tap, 8 channels
  constexpr int JY = WTW / LY, JX = WTW / LX;            // passes per stage (1 or 2)
  // running coordinates of the pixels the NEXT gload stages for X (called with m = mb, mb + 32, ... in this order)
  int cb[JX], cy[JX], cx[JX];
  if (P1) {
    xc = xcol_ok ? k0 + xcl : 0;
  } else {
    const int kk = xcol_ok ? k0 + xcl : 0, tap = kk / g.Ci;
    xc = kk - tap * g.Ci;
    tdy = tap / g.kw - g.pad; tdx = tap - (tap / g.kw) * g.kw - g.pad;
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      const int m = min(mb + xr + LX * j, g.M - 1), hw = g.Ho * g.Wo;
      cb[j] = m / hw;
      const int rem = m - cb[j] * hw;
      cy[j] = rem / g.Wo; cx[j] = rem - cy[j] * g.Wo;
    }
  }
  uint4 ry[2][JY], rx[2][JX];
  // Every load is UNCONDITIONAL — a row past the slice, a tap outside the image or a column past the matrix reads element 0 of its tensor and
  // is zeroed in registers: behind per-lane branches the compiler cannot count the loads in flight and waits vmcnt(0) in front of the LDS stores,
  // i.e. for the loads issued at the TOP of the same step — a prefetch distance of one step's MFMAs (~0.3 us) where the memory latency is 1-2 us
  // (SQ counters, tools/debug/wgrad_pmc.sh: waves waiting 48 % of their cycles).  Straight-line loads get counted waits (vmcnt(6) / vmcnt(4) in the
  // loop, none of 0), and the barriers are the raw instruction behind s_waitcnt lgkmcnt(0): __syncthreads() would drain the loads too.
  // (A zero LINE as the masked source — the igemm kernels' idiom — made clang keep rx / ry in scratch here; the masked value does not.)
  auto gload = [&](int s, int m) {
#pragma unroll
    for (int j = 0; j < JY; ++j) {
      const int r = m + yr + LY * j;
      const bool oky = r < me && ycol_ok;
      uint4 v = *reinterpret_cast<const uint4 *>(dZ + (oky ? (int64_t)r * g.N + n0 + yc : 0));
      v.x = oky ? v.x : 0u; v.y = oky ? v.y : 0u; v.z = oky ? v.z : 0u; v.w = oky ? v.w : 0u;
      ry[s][j] = v;
    }
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      const int r = m + xr + LX * j;
      if (P1) {
        const bool ok = r < me && xcol_ok;
        uint4 v = *reinterpret_cast<const uint4 *>(X + (ok ? (int64_t)r * g.Ci + xc : 0));
        v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
        rx[s][j] = v;
        continue;
      }
      const int iy = cy[j] * g.stride + tdy, ix = cx[j] * g.stride + tdx;
      const bool ok = r < me && xcol_ok && iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi;
      uint4 v = *reinterpret_cast<const uint4 *>(X + (ok ? ((int64_t)(cb[j] * g.Hi + iy) * g.Wi + ix) * g.Ci + xc : 0));
      v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
      rx[s][j] = v;
      cx[j] += WTW;
      while (cx[j] >= g.Wo) { cx[j] -= g.Wo; if (++cy[j] == g.Ho) { cy[j] = 0; ++cb[j]; } }
    }
  };
  const bool do_bias = dB != nullptr && k0 == 0;          // the k-tile-0 workgroups also sum their dz columns (bias gradient)
  float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto lstore = [&](int s, int buf) {
    if (do_bias) {
#pragma unroll
      for (int j = 0; j < JY; ++j) {
        const uint4 v = ry[s][j];
        bs[0] += bf_lo(v.x); bs[1] += bf_hi(v.x); bs[2] += bf_lo(v.y); bs[3] += bf_hi(v.y);
        bs[4] += bf_lo(v.z); bs[5] += bf_hi(v.z); bs[6] += bf_lo(v.w); bs[7] += bf_hi(v.w);
      }
    }
#pragma unroll
    for (int j = 0; j < JY; ++j) *reinterpret_cast<uint4 *>(&SY[buf][yr + LY * j][yc]) = ry[s][j];
#pragma unroll
    for (int j = 0; j < JX; ++j) *reinterpret_cast<uint4 *>(&SX[buf][xr + LX * j][xcl]) = rx[s][j];
  };
  f32x16 acc[NI][KJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < KJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int steps = (me - mb + WTW - 1) / WTW;
  gload(0, mb);                                           // (stages past the slice are zero rows: loaded, stored and multiplied like any other —
  gload(1, mb + WTW);                                     //  the loop below runs an even number of steps with no branch around a load or a wait)
  lstore(0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const int grp = lane >> 4, sl = lane & 15;
  const int frow = 8 * (grp >> 1) + (sl >> 2), fcol = 16 * (grp & 1) + 4 * (sl & 3);
  typedef __attribute__((address_space(3))) v4s16 *lp;
  auto step = [&](int st, int par) {
    gload(par, mb + (st + 2) * WTW);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      hwbf16x8 a[NI], b[KJ];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        union { v4s16 h[2]; hwbf16x8 v; } u;
        u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)&SY[par][ks * 16 + frow][wn + i * 32 + fcol]);
        u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)&SY[par][ks * 16 + frow + 4][wn + i * 32 + fcol]);
        a[i] = u.v;
      }
#pragma unroll
      for (int j = 0; j < KJ; ++j) {
        union { v4s16 h[2]; hwbf16x8 v; } u;
        u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)&SX[par][ks * 16 + frow][wk + j * 32 + fcol]);
        u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)&SX[par][ks * 16 + frow + 4][wk + j * 32 + fcol]);
        b[j] = u.v;
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < KJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    lstore(par ^ 1, par ^ 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (the raw barrier: __syncthreads() would drain the loads in flight)
    __builtin_amdgcn_s_barrier();
  };
  for (int st = 0; st < steps; st += 2) {
    step(st, 0);
    step(st + 1, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the two zero stages still in flight
  __syncthreads();
  if (do_bias) {                                         // the loop ended with a barrier: the stages are free
    float *red = reinterpret_cast<float *>(&SY[0][0][0]);                 // [LY][TN] floats <= 2 * 32 * PN bf16
#pragma unroll
    for (int e = 0; e < 8; ++e) red[yr * TN + yc + e] = bs[e];
    __syncthreads();
    if (t < TN && n0 + t < g.N) {
      float v = 0.f;
      for (int r = 0; r < LY; ++r) v += red[r * TN + t];
      unsafeAtomicAdd(dB + n0 + t, v);
    }
  }
  float *w = ws + ((int64_t)split * g.tiles + tile) * (TN * TK) + t;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < KJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) w[((i * KJ + j) * 16 + e) * 256] = acc[i][j][e];
}

template <int WN, int WK, bool P1>
__global__ __launch_bounds__(256, 3) void conv_wgrad_bf16_tr(const bf16_t *__restrict__ dZ, const bf16_t *__restrict__ X,
                                                              float *__restrict__ ws, WgradGeom g)
{
  wgrad_body<WN, WK, P1>(dZ, X, ws, g, blockIdx.x);
}

// GROUPED form: one launch works through a table of problems (the filter gradients of a whole backbone, deferred to the end of
// its backward pass): block b belongs to the problem p with block_begin[p] <= b < block_begin[p + 1].  A layer's filter
// gradient alone is a 20-40 us launch that cannot fill the chip (and ends in a second launch for its partial tiles); 52 of them
// side by side can.
struct WgradProblem {
  const bf16_t *dz, *x;
  bf16_t *dw;
  int64_t ws_off;                                        // this problem's partial tiles inside the workspace (floats)
  float *db;                                             // fp32 [co] bias-gradient accumulator (+= column sums of dz), nullable
  const float *scale;                                    // fp32 [co], nullable: dW rows are multiplied by it (frozen-BN scale of the fused backbone)
  WgradGeom g;
  int block_begin, reduce_begin, splits, variant;
};

__device__ __forceinline__ int find_problem(const WgradProblem *tab, int count, int b, bool reduce)
{
  int p = 0;
  while (p + 1 < count && (reduce ? tab[p + 1].reduce_begin : tab[p + 1].block_begin) <= b) ++p;
  return p;
}

// The tiles of ONE slice of pixels read the same dZ / X rows (a 3 x 3 layer's nine taps: five 128-wide K tiles over the same pixels), so they
// should meet in ONE L2: the problem's workgroups are renumbered XCD-major (xcd.h) and position p is (slice p / tiles, tile p % tiles) — an XCD
// then owns runs of whole slices.
// (the 1 x 1 stride-1 problems of a tile shape are a launch of their own — P1 — so that each kernel carries ONE loop: with both bodies behind a
//  branch the kernel took 169 VGPRs and the 3 x 3 workgroups of the R50 launch ran 20 % slower.  Measured and dropped: a third class for the
//  stride-1 "same" 3 x 3 problems with one running row offset instead of the (image, y, x) -> address chain — its launch of 1 440 workgroups and
//  the 816 strided ones left over took 145 + 132 us where the common launch takes 249; step 18.78 vs 18.83 ms)
template <int WN, int WK, bool P1>
__global__ __launch_bounds__(256, 3) void conv_wgrad_bf16_tr_grouped(const WgradProblem *__restrict__ tab, int count, float *__restrict__ ws, int xcd_major)
{
  const int p = find_problem(tab, count, blockIdx.x, false);
  const WgradProblem &pr = tab[p];
  const int bid = xcd_major ? pd_xcd_major(pr.block_begin, pr.g.tiles * pr.splits, (int)blockIdx.x) : (int)blockIdx.x - pr.block_begin;
  wgrad_body<WN, WK, P1>(pd_as_global(pr.dz), pd_as_global(pr.x), ws + pr.ws_off, pr.g, bid, pd_as_global(pr.db));   // (pd_common.h: table pointers would be FLAT)
}

// dW (bf16) = sum over splits of the partial tiles (register order: element (ij, e) of thread t at ((ij * 16 + e) * 256 + t)).
// A workgroup sums FOUR (ij, e) slices of a tile, a thread four consecutive producer threads' values (one 16-byte load per split; four
// consecutive lanes of the producer hold four consecutive columns of one row, so the result is one 8-byte store): a quarter of the workgroups of
// the one-float-per-thread form, whose 90 752 blocks of ~12 loads each ran at 2.4 TB/s (154 us for the step's large launch).  Each element is
// summed in the same order as before (bit-identical results).
constexpr int RQ = 4;                                    // (ij, e) slices per workgroup
template <int WN, int WK>
__device__ __forceinline__ void reduce_body(const float *__restrict__ ws, bf16_t *__restrict__ dW, int N, int K, int tiles_k, int tiles,
                                            int splits, int bid, const float *__restrict__ rscale = nullptr)
{
  constexpr int TN = 2 * WN, TK = 2 * WK, KJ = WK / 32, QN = (WN / 32) * KJ * 16;     // QN (ij, e) pairs per tile
  static_assert(QN % RQ == 0, "slices per workgroup");
  const int tile = bid / (QN / RQ), q = (bid % (QN / RQ)) * RQ + (threadIdx.x >> 6);
  const int t = (threadIdx.x & 63) * 4, lane = t & 63, wave = t >> 6;              // the first of this thread's four producer threads
  const int ij = q >> 4, e = q & 15, i = ij / KJ, j = ij % KJ;
  const int n0 = (tile / tiles_k) * TN, k0 = (tile % tiles_k) * TK;
  const int wn = (wave >> 1) * WN, wk = (wave & 1) * WK;
  const float4 *p = reinterpret_cast<const float4 *>(ws + (int64_t)tile * (TN * TK) + q * 256 + t);
  const int64_t stride = (int64_t)tiles * (TN * TK) / 4;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  int sp = 0;
  for (; sp + 7 < splits; sp += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(sp + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      s0.x += v[u].x; s0.y += v[u].y; s0.z += v[u].z; s0.w += v[u].w;
      s1.x += v[u + 1].x; s1.y += v[u + 1].y; s1.z += v[u + 1].z; s1.w += v[u + 1].w;
    }
  }
  for (; sp < splits; ++sp) { const float4 v = p[(int64_t)sp * stride]; s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w; }
  const int c = k0 + wk + j * 32 + (lane & 31);                                   // columns c .. c + 3 (K % 4 == 0: all four inside or none)
  const int row = n0 + wn + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
  if (c < K && row < N) {
    const float sc = rscale ? rscale[row] : 1.f;
    *reinterpret_cast<uint2 *>(dW + (int64_t)row * K + c) = make_uint2(pk_bf16((s0.x + s1.x) * sc, (s0.y + s1.y) * sc), pk_bf16((s0.z + s1.z) * sc, (s0.w + s1.w) * sc));
  }
}

template <int WN, int WK>
__global__ __launch_bounds__(256) void conv_wgrad_reduce(const float *__restrict__ ws, bf16_t *__restrict__ dW, int N, int K, int tiles_k,
                                                         int tiles, int splits)
{
  reduce_body<WN, WK>(ws, dW, N, K, tiles_k, tiles, splits, blockIdx.x);
}

template <int WN, int WK>
__global__ __launch_bounds__(256) void conv_wgrad_reduce_grouped(const WgradProblem *__restrict__ tab, int count, const float *__restrict__ ws)
{
  const int p = find_problem(tab, count, blockIdx.x, true);
  const WgradProblem &pr = tab[p];
  reduce_body<WN, WK>(ws + pr.ws_off, pd_as_global(pr.dw), pr.g.N, pr.g.K, pr.g.tiles_k, pr.g.tiles, pr.splits, blockIdx.x - pr.reduce_begin, pd_as_global(pr.scale));
}

struct WgradPlan { int tn, tk, tiles_k, tiles, splits, m_chunk; };
WgradPlan wgrad_plan(int M, int N, int K, int rows_per_wg = 0)
{
  WgradPlan p;
  p.tn = N <= 64 ? 64 : 128;
  p.tk = K <= 64 ? 64 : 128;
  p.tiles_k = (K + p.tk - 1) / p.tk;
  p.tiles = p.tiles_k * ((N + p.tn - 1) / p.tn);
  // two to three workgroups are resident per CU: one round of ~512, at least 8 stages (256 pixels) per workgroup — every
  // workgroup ends with a tn x tk fp32 partial tile, traffic that rivals the operands' when the pixel ranges get short
  p.splits = p.tiles >= 512 ? 1 : (512 + p.tiles / 2) / p.tiles;
  if (rows_per_wg > 0) p.splits = (M + rows_per_wg - 1) / rows_per_wg;      // grouped launches: the other problems fill the chip
  p.m_chunk = ((M + p.splits - 1) / p.splits + WTW - 1) / WTW * WTW;
  if (p.m_chunk < 8 * WTW) p.m_chunk = 8 * WTW;
  p.splits = (M + p.m_chunk - 1) / p.m_chunk;
  return p;
}

template <int WN, int WK>
void launch_wgrad(const void *dz, const void *x, void *dw, float *ws, const WgradGeom &g, int splits, hipStream_t st)
{
  if (g.kw == 1 && g.stride == 1)
    hipLaunchKernelGGL((conv_wgrad_bf16_tr<WN, WK, true>), dim3((unsigned)(g.tiles * splits)), dim3(256), 0, st, (const bf16_t *)dz, (const bf16_t *)x, ws, g);
  else
    hipLaunchKernelGGL((conv_wgrad_bf16_tr<WN, WK, false>), dim3((unsigned)(g.tiles * splits)), dim3(256), 0, st, (const bf16_t *)dz, (const bf16_t *)x, ws, g);
  hipLaunchKernelGGL((conv_wgrad_reduce<WN, WK>), dim3((unsigned)(g.tiles * (WN / 32) * (WK / 32) * 16 / RQ)), dim3(256), 0, st, (const float *)ws,
                     (bf16_t *)dw, g.N, g.K, g.tiles_k, g.tiles, splits);
}

int check_geom(int batch, int hi, int wi, int ci, int ho, int wo, int co, int k, int stride, int pad)
{
  if (batch <= 0 || hi <= 0 || wi <= 0 || ho <= 0 || wo <= 0) return PD_ERR_INVALID_ARG;
  if (ci % 64 || co % 64 || (k != 1 && k != 3) || (stride != 1 && stride != 2) || pad != k / 2) return PD_ERR_INVALID_ARG;
  if (ho != (hi + 2 * pad - k) / stride + 1 || wo != (wi + 2 * pad - k) / stride + 1) return PD_ERR_INVALID_ARG;
  if ((int64_t)batch * hi * wi >= (1ll << 31) / 8 || (int64_t)batch * hi * wi * ci >= (1ll << 40)) return PD_ERR_INVALID_ARG;
  return 0;
}

}  // namespace

extern "C" int pd_conv_bf16_supported(int ci, int co, int k, int stride, int pad)
{
  return ci % 64 == 0 && co % 64 == 0 && (k == 1 || k == 3) && (stride == 1 || stride == 2) && pad == k / 2;
}

extern "C" int pd_conv_bf16_fwd(const void *x, const void *w, const float *scale, const float *bias, const void *residual, void *y,
                                int batch, int hi, int wi, int ci, int ho, int wo, int co, int k, int stride, int pad, int relu, void *stream)
{
  if (!x || !w || !y) return PD_ERR_INVALID_ARG;
  if (int rc = check_geom(batch, hi, wi, ci, ho, wo, co, k, stride, pad)) return rc;
  ConvGeom g{batch * ho * wo, co, ci, k, k * k, hi, wi, ho, wo, stride, pad};
  return co % 128 == 0 ? launch_conv<128, false>(x, w, scale, bias, residual, y, g, relu, (hipStream_t)stream)
                       : launch_conv<64, false>(x, w, scale, bias, residual, y, g, relu, (hipStream_t)stream);
}

extern "C" int pd_conv_bf16_dgrad(const void *dz, const void *wt, const void *addend, void *dx, int batch, int hi, int wi, int ci, int ho,
                                  int wo, int co, int k, int stride, int pad, void *stream)
{
  if (!dz || !wt || !dx) return PD_ERR_INVALID_ARG;
  if (int rc = check_geom(batch, hi, wi, ci, ho, wo, co, k, stride, pad)) return rc;
  ConvGeom g{batch * hi * wi, ci, co, k, k * k, ho, wo, hi, wi, stride, pad};
  return ci % 128 == 0 ? launch_conv<128, true>(dz, wt, nullptr, nullptr, addend, dx, g, 0, (hipStream_t)stream)
                       : launch_conv<64, true>(dz, wt, nullptr, nullptr, addend, dx, g, 0, (hipStream_t)stream);
}

extern "C" int64_t pd_conv_bf16_wgrad_workspace_floats(int batch, int ho, int wo, int ci, int co, int k)
{
  const WgradPlan p = wgrad_plan(batch * ho * wo, co, k * k * ci);
  return (int64_t)p.tiles * p.splits * p.tn * p.tk;
}

extern "C" int pd_conv_bf16_wgrad(const void *dz, const void *x, void *dw, float *workspace, int64_t workspace_floats, int batch, int hi,
                                  int wi, int ci, int ho, int wo, int co, int k, int stride, int pad, void *stream)
{
  if (!dz || !x || !dw || !workspace) return PD_ERR_INVALID_ARG;
  if (batch <= 0 || hi <= 0 || wi <= 0 || ho <= 0 || wo <= 0 || ci % 8 || co % 8 || k < 1 || stride < 1 || pad < 0) return PD_ERR_INVALID_ARG;
  if (ho != (hi + 2 * pad - k) / stride + 1 || wo != (wi + 2 * pad - k) / stride + 1) return PD_ERR_INVALID_ARG;
  if ((int64_t)batch * hi * wi >= (1ll << 31) / 8 || (int64_t)batch * ho * wo >= (1ll << 31) - 4096) return PD_ERR_INVALID_ARG;
  WgradGeom g;
  g.M = batch * ho * wo; g.N = co; g.K = k * k * ci; g.Ci = ci; g.kw = k; g.Hi = hi; g.Wi = wi; g.Ho = ho; g.Wo = wo; g.stride = stride; g.pad = pad;
  const WgradPlan p = wgrad_plan(g.M, g.N, g.K);
  g.tiles_k = p.tiles_k; g.tiles = p.tiles; g.m_chunk = p.m_chunk;
  if (workspace_floats < (int64_t)p.tiles * p.splits * p.tn * p.tk) return PD_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (p.tn == 128 && p.tk == 128) launch_wgrad<64, 64>(dz, x, dw, workspace, g, p.splits, st);
  else if (p.tn == 128) launch_wgrad<64, 32>(dz, x, dw, workspace, g, p.splits, st);
  else if (p.tk == 128) launch_wgrad<32, 64>(dz, x, dw, workspace, g, p.splits, st);
  else launch_wgrad<32, 32>(dz, x, dw, workspace, g, p.splits, st);
  return pd_check_launch("pd_conv_bf16_wgrad");
}

// ------------------------------------------------------------------------------------------------ grouped filter gradients
int g_pd_dbg_conv_group_rows = 0;                        // tools/ only (pd_debug_set "conv_group_rows")
int g_pd_dbg_conv_xcd_major = 1;                         // tools/ only (pd_debug_set "conv_xcd_major"): 0 = the tiles of a slice round-robin over the XCDs
namespace {
#define kGroupRows (g_pd_dbg_conv_group_rows > 0 ? g_pd_dbg_conv_group_rows : 2048)   // pixels per workgroup in a grouped launch

int variant_of(const WgradPlan &p) { return p.tn == 128 ? (p.tk == 128 ? 0 : 1) : (p.tk == 128 ? 2 : 3); }

// plans of a group: kGroupRows pixels per workgroup, halved for a tile shape whose launch would otherwise hold fewer than ~768
// workgroups (each tile shape is its own launch: a lone 64 -> 64 1 x 1 layer of res2 is ONE tile — 64 workgroups at 2 048 rows)
void plan_group(const PdConvWgradDesc *descs, int count, WgradPlan *plans)
{
  int rows[4] = {kGroupRows, kGroupRows, kGroupRows, kGroupRows};
  for (int pass = 0; pass < 4; ++pass) {
    int blocks[4] = {0, 0, 0, 0};
    for (int i = 0; i < count; ++i) {
      const int M = descs[i].batch * descs[i].ho * descs[i].wo, N = descs[i].co, K = descs[i].k * descs[i].k * descs[i].ci;
      const int v = variant_of(wgrad_plan(M, N, K, kGroupRows));
      plans[i] = wgrad_plan(M, N, K, rows[v]);
      blocks[v] += plans[i].tiles * plans[i].splits;
    }
    bool again = false;
    for (int v = 0; v < 4; ++v)
      if (blocks[v] > 0 && blocks[v] < 768 && rows[v] > 256) { rows[v] /= 2; again = true; }
    if (!again) break;
  }
}

bool desc_ok(const PdConvWgradDesc &d)
{
  if (!d.dz || !d.x || !d.dw) return false;
  if (d.batch <= 0 || d.hi <= 0 || d.wi <= 0 || d.ho <= 0 || d.wo <= 0 || d.ci % 8 || d.co % 8 || d.k < 1 || d.stride < 1 || d.pad < 0) return false;
  if (d.ho != (d.hi + 2 * d.pad - d.k) / d.stride + 1 || d.wo != (d.wi + 2 * d.pad - d.k) / d.stride + 1) return false;
  return (int64_t)d.batch * d.hi * d.wi < (1ll << 31) / 8 && (int64_t)d.batch * d.ho * d.wo < (1ll << 31) - 4096;
}
}  // namespace

extern "C" int64_t pd_conv_bf16_wgrad_grouped_table_bytes(int count) { return (int64_t)count * (int64_t)sizeof(WgradProblem); }

extern "C" int64_t pd_conv_bf16_wgrad_grouped_workspace_floats(const PdConvWgradDesc *descs, int count)
{
  if (count <= 0 || count > 256 || !descs) return 0;
  WgradPlan plans[256];
  plan_group(descs, count, plans);
  int64_t total = 0;
  for (int i = 0; i < count; ++i) total += (int64_t)plans[i].tiles * plans[i].splits * plans[i].tn * plans[i].tk;
  return total;
}

extern "C" int pd_conv_bf16_wgrad_grouped(const PdConvWgradDesc *descs, int count, void *table_host_pinned, void *table_device,
                                          float *workspace, int64_t workspace_floats, void *stream)
{
  if (count < 0 || (count > 0 && (!descs || !table_host_pinned || !table_device || !workspace))) return PD_ERR_INVALID_ARG;
  if (count == 0) return 0;
  WgradProblem *tab = reinterpret_cast<WgradProblem *>(table_host_pinned);
  int n_of[4] = {0, 0, 0, 0}, n_p1[4] = {0, 0, 0, 0}, blocks[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}}, rblocks[4] = {0, 0, 0, 0};
  // problems sorted by tile variant (four contiguous sub-tables, one launch pair each)
  int order[4][256];
  if (count > 256) return PD_ERR_INVALID_ARG;
  WgradPlan plans[256];
  for (int i = 0; i < count; ++i)
    if (!desc_ok(descs[i])) return PD_ERR_INVALID_ARG;
  plan_group(descs, count, plans);
  for (int pass = 0; pass < 2; ++pass)                    // a variant's 1 x 1 stride-1 problems first: they are a launch of their own (P1)
    for (int i = 0; i < count; ++i) {
      const int v = variant_of(plans[i]);
      const bool p1 = descs[i].k == 1 && descs[i].stride == 1;
      if (p1 != (pass == 0)) continue;
      order[v][n_of[v]++] = i;
      n_p1[v] += p1;
    }
  int64_t ws_off = 0;
  int start[4], at = 0;
  for (int v = 0; v < 4; ++v) {
    start[v] = at;
    for (int q = 0; q < n_of[v]; ++q, ++at) {
      const int i = order[v][q];
      const PdConvWgradDesc &d = descs[i];
      const WgradPlan &p = plans[i];
      WgradProblem &w = tab[at];
      w.dz = (const bf16_t *)d.dz; w.x = (const bf16_t *)d.x; w.dw = (bf16_t *)d.dw; w.ws_off = ws_off; w.db = d.db; w.scale = d.scale;
      w.g.M = d.batch * d.ho * d.wo; w.g.N = d.co; w.g.K = d.k * d.k * d.ci; w.g.Ci = d.ci; w.g.kw = d.k; w.g.Hi = d.hi; w.g.Wi = d.wi;
      w.g.Ho = d.ho; w.g.Wo = d.wo; w.g.stride = d.stride; w.g.pad = d.pad; w.g.tiles_k = p.tiles_k; w.g.tiles = p.tiles; w.g.m_chunk = p.m_chunk;
      const int p1 = q < n_p1[v] ? 1 : 0;
      w.block_begin = blocks[v][p1]; w.reduce_begin = rblocks[v]; w.splits = p.splits; w.variant = v;
      blocks[v][p1] += p.tiles * p.splits;
      rblocks[v] += p.tiles * (p.tn / 64) * (p.tk / 64) * 16 / RQ;
      ws_off += (int64_t)p.tiles * p.splits * p.tn * p.tk;
    }
  }
  if (ws_off > workspace_floats) return PD_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemcpyAsync(table_device, table_host_pinned, (size_t)count * sizeof(WgradProblem), hipMemcpyHostToDevice, st) != hipSuccess)
    return PD_ERR_LAUNCH;
  const WgradProblem *dt = reinterpret_cast<const WgradProblem *>(table_device);
#define PD_GROUP(V, WN, WK)                                                                                                         \
  if (n_of[V]) {                                                                                                                    \
    if (n_p1[V])                                                                                                                    \
      hipLaunchKernelGGL((conv_wgrad_bf16_tr_grouped<WN, WK, true>), dim3((unsigned)blocks[V][1]), dim3(256), 0, st, dt + start[V], n_p1[V], workspace, g_pd_dbg_conv_xcd_major); \
    if (n_of[V] > n_p1[V])                                                                                                          \
      hipLaunchKernelGGL((conv_wgrad_bf16_tr_grouped<WN, WK, false>), dim3((unsigned)blocks[V][0]), dim3(256), 0, st, dt + start[V] + n_p1[V], n_of[V] - n_p1[V], workspace, g_pd_dbg_conv_xcd_major); \
    hipLaunchKernelGGL((conv_wgrad_reduce_grouped<WN, WK>), dim3((unsigned)rblocks[V]), dim3(256), 0, st, dt + start[V], n_of[V],    \
                       (const float *)workspace);                                                                                   \
  }
  PD_GROUP(0, 64, 64) PD_GROUP(1, 64, 32) PD_GROUP(2, 32, 64) PD_GROUP(3, 32, 32)
#undef PD_GROUP
  return pd_check_launch("pd_conv_bf16_wgrad_grouped");
}
