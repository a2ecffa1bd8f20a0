// MX-fp8 GEMM with fused epilogues on the gfx950 matrix cores (C-ABI: include/pd_mx8.h): the Swin Linears of BASELINE config 5,
// forward and input gradient, on v_mfma_scale_f32_32x32x64_f8f6f4 — the block-scaled instruction, the only fp8 form that runs at twice
// the bf16 rate (MI355X_MICROARCH.md; tools/probes/mfma_mx8_probe.hip pinned its operand map on the hardware):
//
//   a lane (row r = lane % 32, half h = lane / 32) supplies 32 bytes of the 64-wide step: k [16 h, 16 h + 16) in registers 0-3 and
//   k [32 + 16 h, 48 + 16 h) in registers 4-7; its scale register's byte `op_sel` is the E8M0 exponent of row r's 32-block h
//   (k [32 h, 32 h + 32)); D[i][j] with j = lane % 32 of the SECOND operand and i = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
//
// That register image is exactly two of csrc/igemm_bf16.hip's ds_read_b128 fragment reads, so the kernel is that skeleton with bytes
// for elements: 128 x 128 / 128 x 64 tiles, a K-step of 128 BYTES per row (128 elements: two matrix instructions deep instead of
// four — the LDS traffic per instruction is what bounds a 64 x 64 wavefront tile, and here every instruction is worth twice the
// flops), tiles global -> LDS directly with the XOR swizzle on the source address, rows beyond M from a zero line, one stage with four
// workgroups per CU or two stages with two.  New: the scale bytes of a step (4 per row) travel as ONE dword per row through the same
// direct-to-LDS path; a lane shifts its row's dword by 8 h once per step and op_sel picks byte 0 / 2 for the step's two instructions.
// Epilogue from the accumulators (v_permlane32_swap gives a lane 8 consecutive channels): + bias -> pre-activation copy -> exact-erf
// GELU -> * GELU'(gate) -> bf16 store, and — the point of a LOCAL scale format — the result again as MX fp8 for the next GEMM: a
// 32-channel block of a row is two lanes' sixteen values (one more permlane32_swap for the block maximum), so the quantised copy
// costs a max, a scale and a convert per value and no extra pass over memory.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gelu.h"
#include "mfma_bf16.h"
#include "mx8_quant.h"
#include "pd_common.h"
#include "xcd.h"
#include "pd_igemm.h"
#include "pd_msda.h"
#include "pd_mx8.h"

int g_mx_bn = 0, g_mx_nst = 0;                             // pd_debug_set "mx_bn" / "mx_nst" (tools/ only; 0 = automatic)

namespace {
using namespace pdmfma;

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void *lds_ptr;
typedef __attribute__((address_space(1))) const void *glb_ptr;

constexpr int BM = 128, BKB = 128;                         // rows per tile, bytes (= elements) per row of a K-step
__device__ __attribute__((aligned(128))) unsigned char g_mx_zero_line[128];

using namespace pdmx;

// ------------------------------------------------------------------------------------------------ standalone quantisation
// one lane = 8 consecutive elements (16 bytes in, 8 out), four lanes = one block: the block maximum is two quad-permute steps
__device__ __forceinline__ float quad_max(float x)
{
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true)));   // quad_perm [1,0,3,2]
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true)));   // quad_perm [2,3,0,1]
  return x;
}

template <int FMT>
__device__ __forceinline__ void quantize_piece(const bf16_t *src, uint8_t *q, uint8_t *sbyte, bool live)
{
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (live) {
    const uint4 a = *reinterpret_cast<const uint4 *>(src);
    v[0] = bf_lo(a.x); v[1] = bf_hi(a.x); v[2] = bf_lo(a.y); v[3] = bf_hi(a.y);
    v[4] = bf_lo(a.z); v[5] = bf_hi(a.z); v[6] = bf_lo(a.w); v[7] = bf_hi(a.w);
  }
  float m = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));                       // fmaxf drops a NaN: it does not poison the block's scale
  m = quad_max(m);
  float mult;
  const unsigned byte = mx_exponent<FMT>(m, mult);
  if (!live) return;
  *reinterpret_cast<uint2 *>(q) = mx_pack8<FMT>(v, mult);
  if (sbyte) *sbyte = (uint8_t)byte;
}

template <int FMT>
__global__ __launch_bounds__(256) void mx8_quantize_rows(const bf16_t *__restrict__ x, int64_t rows, int cols, int64_t ldx,
                                                         uint8_t *__restrict__ q, uint8_t *__restrict__ s)
{
  const int bpr = cols >> 5;
  const int64_t npieces = rows * bpr * 4;
  for (int64_t base = (int64_t)blockIdx.x * 256; base < npieces; base += (int64_t)gridDim.x * 256) {
    const int64_t i = base + threadIdx.x;
    const bool live = i < npieces;
    const int64_t blk = (live ? i : 0) >> 2;
    const int sub = (int)(i & 3);
    const int64_t row = blk / bpr;
    const int cb = (int)(blk - row * bpr);
    quantize_piece<FMT>(x + row * ldx + cb * 32 + sub * 8, q + row * cols + cb * 32 + sub * 8, sub == 0 ? s + blk : nullptr, live);
  }
}

struct QtEntry { const bf16_t *x; uint8_t *q, *s; int64_t first_piece; };       // pieces of 8 elements, ascending first_piece

template <int FMT>
__global__ __launch_bounds__(256) void mx8_quantize_grouped(const QtEntry *__restrict__ tab, int count, int64_t npieces)
{
  for (int64_t base = (int64_t)blockIdx.x * 256; base < npieces; base += (int64_t)gridDim.x * 256) {
    const int64_t i = base + threadIdx.x;
    const bool live = i < npieces;
    const int64_t ii = live ? i : npieces - 1;
    int lo = 0, hi = count - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (tab[mid].first_piece <= ii) lo = mid; else hi = mid - 1;
    }
    const QtEntry e = tab[lo];
    const int64_t p = ii - e.first_piece;                                       // a tensor's pieces are a multiple of 4: a quad never straddles two
    quantize_piece<FMT>(e.x + p * 8, e.q + p * 8, (p & 3) == 0 ? e.s + (p >> 2) : nullptr, live);
  }
}

// ------------------------------------------------------------------------------------------------ the GEMM
struct MxArgs {
  const uint8_t *Aq, *As, *Wq, *Ws;
  const void *bias;
  const bf16_t *gate;
  bf16_t *Y, *Ypre;
  uint8_t *Yq, *Ys;
  int M, N, K, KT;
  int act, gate_mode, bias_bf16;
  int ntn;
};

__device__ __forceinline__ int xcd_chunk(int bid, int nb) { return pd_xcd_chunk(bid, nb); }   // xcd.h: any workgroup count
__device__ __forceinline__ float gelu_f(float x) { return pdgelu::gelu(x); }
__device__ __forceinline__ float gelu_grad_f(float x) { return pdgelu::gelu_grad(x); }
__device__ __forceinline__ void swap_halves(float &a, float &b)
{
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}

// ABF8: the A operand (second MFMA operand) is e5m2;  OFMT: format of the quantised output copy
template <int BN, int NST, bool ABF8, int OFMT>
__global__ __launch_bounds__(256, NST == 1 ? 4 : 2) void gemm_mx8(MxArgs a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TILE = (BM + BN) * BKB;                    // element bytes per stage: A rows then W rows
  constexpr int STAGE = TILE + (BM + BN) * 4;              // + one scale dword per row
  constexpr int SB_OFF = NST * STAGE;                      // behind the stages: bias[BN] (fp32)
  constexpr int MI = BN == 128 ? 2 : 1;
  constexpr int NBJ = BN / 32;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lb = xcd_chunk(blockIdx.x, gridDim.x);
  const int m0 = (lb / a.ntn) * BM, n0 = (lb % a.ntn) * BN;
  const int wm = BN == 128 ? (wave >> 1) * 64 : wave * 32, wn = BN == 128 ? (wave & 1) * 64 : 0;
  const int kb = a.K >> 5;                                 // scale bytes per row

  const int lrow = lane >> 3, lch = lane & 7;
  const uint8_t *zline = g_mx_zero_line;
  float bv = 0.f;
  if (t < BN && a.bias)
    bv = a.bias_bf16 ? __uint_as_float((unsigned)reinterpret_cast<const bf16_t *>(a.bias)[n0 + t] << 16) : reinterpret_cast<const float *>(a.bias)[n0 + t];
  const uint8_t *ap[4];
  bool aok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (wave * 4 + j) * 8 + lrow, m = m0 + row;
    aok[j] = m < a.M;
    ap[j] = aok[j] ? a.Aq + (int64_t)m * a.K + (lch ^ ((row >> 1) & 7)) * 16 : zline + lch * 16;
  }
  const uint8_t *wb[NBJ];
#pragma unroll
  for (int j = 0; j < NBJ; ++j) {
    const int row = (wave * NBJ + j) * 8 + lrow;
    wb[j] = a.Wq + (int64_t)(n0 + row) * a.K + (lch ^ ((row >> 1) & 7)) * 16;
  }
  // the scale dword of combined row t (A rows 0..127, then the W rows): 4 bytes per step, contiguous along k
  const bool s_live = t < BM + BN;
  bool s_ok = false;
  const uint8_t *sp = zline + (lane & 31) * 4;
  if (t < BM) { s_ok = m0 + t < a.M; if (s_ok) sp = a.As + (int64_t)(m0 + t) * kb; }
  else if (s_live) { s_ok = true; sp = a.Ws + (int64_t)(n0 + t - BM) * kb; }

  auto issue = [&](int kt, int buf) {
    unsigned char *As = smem + buf * STAGE, *Bs = As + BM * BKB, *Sc = As + TILE;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((glb_ptr)(aok[j] ? ap[j] + (int64_t)kt * BKB : ap[j]), (lds_ptr)(As + (wave * 4 + j) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < NBJ; ++j)
      __builtin_amdgcn_global_load_lds((glb_ptr)(wb[j] + (int64_t)kt * BKB), (lds_ptr)(Bs + (wave * NBJ + j) * 1024), 16, 0, 0);
    if (BN == 128 || wave < 3)                             // wave-uniform: 128 + 64 rows are three wavefronts' worth
      __builtin_amdgcn_global_load_lds((glb_ptr)(s_ok ? sp + kt * 4 : sp), (lds_ptr)(Sc + wave * 256), 4, 0, 0);
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

  auto frag = [&](const unsigned char *rowp, int j) {
    const int4 lo = *reinterpret_cast<const int4 *>(rowp + foff[2 * j]), hi = *reinterpret_cast<const int4 *>(rowp + foff[2 * j + 1]);
    i32x8 r;
    r[0] = lo.x; r[1] = lo.y; r[2] = lo.z; r[3] = lo.w; r[4] = hi.x; r[5] = hi.y; r[6] = hi.z; r[7] = hi.w;
    return r;
  };

  auto compute = [&](int buf) {
    const unsigned char *As = smem + buf * STAGE, *Bs = As + BM * BKB;
    const unsigned *Sc = reinterpret_cast<const unsigned *>(As + TILE);
    int sa[MI], swt[2];
#pragma unroll
    for (int j = 0; j < MI; ++j) sa[j] = (int)(Sc[wm + j * 32 + fr] >> (8 * kh));
#pragma unroll
    for (int i = 0; i < 2; ++i) swt[i] = (int)(Sc[BM + wn + i * 32 + fr] >> (8 * kh));
    {
      i32x8 wf[2], af[MI];
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[i] = frag(Bs + (wn + i * 32 + fr) * BKB, 0);
#pragma unroll
      for (int j = 0; j < MI; ++j) af[j] = frag(As + (wm + j * 32 + fr) * BKB, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[i], af[j], acc[i][j], 0, ABF8 ? 1 : 0, 0, swt[i], 0, sa[j]);
    }
    {
      i32x8 wf[2], af[MI];
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[i] = frag(Bs + (wn + i * 32 + fr) * BKB, 1);
#pragma unroll
      for (int j = 0; j < MI; ++j) af[j] = frag(As + (wm + j * 32 + fr) * BKB, 1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[i], af[j], acc[i][j], 0, ABF8 ? 1 : 0, 2, swt[i], 2, sa[j]);
    }
  };

  if (t < BN) reinterpret_cast<float *>(smem + SB_OFF)[t] = bv;
  if (NST == 2) {
    issue(0, 0);
    for (int kt = 0; kt < a.KT; ++kt) {
      const int buf = kt & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wavefront's pieces of step kt have landed ...
      __syncthreads();                                     // ... everyone's have, and everyone is done reading the other stage
      if (kt + 1 < a.KT) issue(kt + 1, buf ^ 1);
      compute(buf);
    }
  } else {
    for (int kt = 0; kt < a.KT; ++kt) {
      issue(kt, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute(0);
      __syncthreads();                                     // the stage is free for the next step's loads
    }
  }

  // ---- epilogue, straight from the accumulators
  const float *sb = reinterpret_cast<const float *>(smem + SB_OFF);
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int m = m0 + wm + j * 32 + fr;
    if (m >= a.M) continue;                                // (lanes l and l + 32 hold the same row: the pair leaves together)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float v[2][8];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[p][e] = acc[i][j][8 * p + e]; v[p][4 + e] = acc[i][j][8 * p + 4 + e]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) swap_halves(v[p][e], v[p][4 + e]);
        // lane < 32: channels 16 p .. 16 p + 7 of the 32-wide piece; lane >= 32: 16 p + 8 .. 16 p + 15
        const int cl = wn + i * 32 + p * 16 + kh * 8;
        const int64_t off = (int64_t)m * a.N + n0 + cl;
        const float4 b0 = *reinterpret_cast<const float4 *>(sb + cl), b1 = *reinterpret_cast<const float4 *>(sb + cl + 4);
        v[p][0] += b0.x; v[p][1] += b0.y; v[p][2] += b0.z; v[p][3] += b0.w; v[p][4] += b1.x; v[p][5] += b1.y; v[p][6] += b1.z; v[p][7] += b1.w;
        if (a.Ypre)
          *reinterpret_cast<uint4 *>(a.Ypre + off) = make_uint4(pk_bf16(v[p][0], v[p][1]), pk_bf16(v[p][2], v[p][3]), pk_bf16(v[p][4], v[p][5]), pk_bf16(v[p][6], v[p][7]));
        if (a.act == PD_IG_ACT_GELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[p][e] = gelu_f(v[p][e]);
        }
        if (a.gate) {
          const uint4 r = *reinterpret_cast<const uint4 *>(a.gate + off);
          const float g[8] = {bf_lo(r.x), bf_hi(r.x), bf_lo(r.y), bf_hi(r.y), bf_lo(r.z), bf_hi(r.z), bf_lo(r.w), bf_hi(r.w)};
#pragma unroll
          for (int e = 0; e < 8; ++e) v[p][e] *= gelu_grad_f(g[e]);
        }
        const uint4 o = make_uint4(pk_bf16(v[p][0], v[p][1]), pk_bf16(v[p][2], v[p][3]), pk_bf16(v[p][4], v[p][5]), pk_bf16(v[p][6], v[p][7]));
        *reinterpret_cast<uint4 *>(a.Y + off) = o;
        if (a.Yq) {                                        // quantise what the bf16 copy holds (the two copies then describe the same tensor)
          v[p][0] = bf_lo(o.x); v[p][1] = bf_hi(o.x); v[p][2] = bf_lo(o.y); v[p][3] = bf_hi(o.y);
          v[p][4] = bf_lo(o.z); v[p][5] = bf_hi(o.z); v[p][6] = bf_lo(o.w); v[p][7] = bf_hi(o.w);
        }
      }
      if (a.Yq) {
        float mx = 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(v[p][e]));
        {
          const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
          mx = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));   // the pair's maximum = the 32-channel block's
        }
        float mult;
        const unsigned byte = mx_exponent<OFMT>(mx, mult);
        const int cb = n0 + wn + i * 32;
#pragma unroll
        for (int p = 0; p < 2; ++p)
          *reinterpret_cast<uint2 *>(a.Yq + (int64_t)m * a.N + cb + p * 16 + kh * 8) = mx_pack8<OFMT>(v[p], mult);
        if (kh == 0) a.Ys[(int64_t)m * (a.N >> 5) + (cb >> 5)] = (uint8_t)byte;
      }
    }
  }
}

template <int BN, int NST, bool ABF8, int OFMT>
int launch1(const MxArgs &a, hipStream_t stream)
{
  constexpr size_t lds = (size_t)NST * ((BM + BN) * BKB + (BM + BN) * 4) + BN * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)gemm_mx8<BN, NST, ABF8, OFMT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const int64_t nblocks = (int64_t)((a.M + BM - 1) / BM) * a.ntn;
  hipLaunchKernelGGL((gemm_mx8<BN, NST, ABF8, OFMT>), dim3((unsigned)nblocks), dim3(256), lds, stream, a);
  return pd_check_launch("pd_mx8_gemm");
}

template <int BN, int NST>
int launch(const MxArgs &a, bool abf8, int ofmt, hipStream_t st)
{
  if (abf8) return ofmt == PD_MX8_E5M2 ? launch1<BN, NST, true, PD_MX8_E5M2>(a, st) : launch1<BN, NST, true, PD_MX8_E4M3>(a, st);
  return ofmt == PD_MX8_E5M2 ? launch1<BN, NST, false, PD_MX8_E5M2>(a, st) : launch1<BN, NST, false, PD_MX8_E4M3>(a, st);
}

int blocks_for(int64_t npieces) { const int64_t b = (npieces + 255) / 256; return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); }
}  // namespace

extern "C" int pd_mx8_gemm_supported(int32_t m, int32_t n, int32_t k) { return m >= 1 && n > 0 && k > 0 && n % 64 == 0 && k % 128 == 0; }

extern "C" int pd_mx8_gemm(const PdMx8Gemm *p, void *stream)
{
  if (!p || !p->a_q || !p->a_s || !p->w_q || !p->w_s || !p->out) return pd_set_error(PD_ERR_INVALID_ARG, "pd_mx8_gemm: null pointer");
  if (!pd_mx8_gemm_supported(p->m, p->n, p->k))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_mx8_gemm: unsupported shape m=%d n=%d k=%d (n %% 64 == 0, k %% 128 == 0)", p->m, p->n, p->k);
  if ((p->a_format != PD_MX8_E4M3 && p->a_format != PD_MX8_E5M2) || (p->out_q && p->out_format != PD_MX8_E4M3 && p->out_format != PD_MX8_E5M2))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_mx8_gemm: unknown fp8 format");
  if ((p->out_q != nullptr) != (p->out_s != nullptr)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_mx8_gemm: out_q and out_s come together");
  if (p->act != PD_IG_ACT_NONE && p->act != PD_IG_ACT_GELU) return pd_set_error(PD_ERR_INVALID_ARG, "pd_mx8_gemm: act must be NONE or GELU");
  if (p->gate && p->gate_mode != PD_IG_GATE_GELU) return pd_set_error(PD_ERR_INVALID_ARG, "pd_mx8_gemm: gate_mode must be PD_IG_GATE_GELU");
  if ((((uintptr_t)p->a_q | (uintptr_t)p->w_q | (uintptr_t)p->out | (uintptr_t)p->out_pre | (uintptr_t)p->gate) & 15) || ((uintptr_t)p->out_q & 7))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_mx8_gemm: misaligned operand");
  MxArgs a;
  a.Aq = p->a_q; a.As = p->a_s; a.Wq = p->w_q; a.Ws = p->w_s; a.bias = p->bias; a.gate = (const bf16_t *)p->gate;
  a.Y = (bf16_t *)p->out; a.Ypre = (bf16_t *)p->out_pre; a.Yq = p->out_q; a.Ys = p->out_s;
  a.M = p->m; a.N = p->n; a.K = p->k; a.KT = p->k / BKB; a.act = p->act; a.gate_mode = p->gate_mode; a.bias_bf16 = p->bias_bf16;
  const int mt = (a.M + BM - 1) / BM;
  // schedule from tools/bench_mx8.py (profiles/r04_mx8_sweep.txt): 128-wide tiles whenever n allows (64-wide lost 5-40 % at every Swin-L shape,
  // also where 128-wide tiles number fewer than the 1 024 workgroup slots); one stage with four workgroups per CU, except few tiles under
  // a long contraction (stage 4: 36 x 12 tiles, k >= 1 536), where a step of prefetch is worth 3-4 %
  int bn = (p->n % 128 == 0) ? 128 : 64;
  if (g_mx_bn == 64 || (g_mx_bn == 128 && p->n % 128 == 0)) bn = g_mx_bn;
  int nst = bn == 128 ? ((mt * (p->n / 128) <= 512 && a.KT >= 12) ? 2 : 1) : (a.KT >= 8 ? 2 : 1);
  if (g_mx_nst == 1 || g_mx_nst == 2) nst = g_mx_nst;
  a.ntn = p->n / bn;
  const bool abf8 = p->a_format == PD_MX8_E5M2;
  const int ofmt = p->out_q ? p->out_format : PD_MX8_E4M3;
  hipStream_t st = (hipStream_t)stream;
  if (bn == 128) return nst == 2 ? launch<128, 2>(a, abf8, ofmt, st) : launch<128, 1>(a, abf8, ofmt, st);
  return nst == 2 ? launch<64, 2>(a, abf8, ofmt, st) : launch<64, 1>(a, abf8, ofmt, st);
}

extern "C" int pd_mx8_quantize_bf16(const void *x, int64_t rows, int32_t cols, int64_t ldx, int32_t format, uint8_t *q, uint8_t *s, void *stream)
{
  if (rows < 0 || cols <= 0 || cols % 32 || ldx < cols || ldx % 8) return pd_set_error(PD_ERR_INVALID_ARG, "pd_mx8_quantize_bf16: cols %% 32 == 0, ldx %% 8 == 0, ldx >= cols required");
  if (format != PD_MX8_E4M3 && format != PD_MX8_E5M2) return pd_set_error(PD_ERR_INVALID_ARG, "pd_mx8_quantize_bf16: unknown format %d", format);
  if (rows == 0) return PD_OK;
  if (!x || !q || !s || ((uintptr_t)x & 15) || ((uintptr_t)q & 7)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_mx8_quantize_bf16: null / misaligned pointer");
  const int64_t npieces = rows * (cols / 8);
  hipStream_t st = (hipStream_t)stream;
  if (format == PD_MX8_E4M3) hipLaunchKernelGGL((mx8_quantize_rows<PD_MX8_E4M3>), dim3(blocks_for(npieces)), dim3(256), 0, st, (const bf16_t *)x, rows, cols, ldx, q, s);
  else hipLaunchKernelGGL((mx8_quantize_rows<PD_MX8_E5M2>), dim3(blocks_for(npieces)), dim3(256), 0, st, (const bf16_t *)x, rows, cols, ldx, q, s);
  return pd_check_launch("pd_mx8_quantize_bf16");
}

extern "C" int64_t pd_mx8_quantize_table_bytes(int32_t count) { return (int64_t)(count > 0 ? count : 0) * (int64_t)sizeof(QtEntry); }

extern "C" int pd_mx8_quantize_grouped(const PdMx8Tensor *list, int32_t count, int32_t format, void *table_host_pinned, void *table_device, void *stream)
{
  if (count <= 0) return PD_OK;
  if (!list || !table_host_pinned || !table_device) return pd_set_error(PD_ERR_INVALID_ARG, "pd_mx8_quantize_grouped: null pointer");
  if (format != PD_MX8_E4M3 && format != PD_MX8_E5M2) return pd_set_error(PD_ERR_INVALID_ARG, "pd_mx8_quantize_grouped: unknown format %d", format);
  QtEntry *h = reinterpret_cast<QtEntry *>(table_host_pinned);
  int64_t pieces = 0;
  for (int i = 0; i < count; ++i) {
    const PdMx8Tensor &d = list[i];
    if (d.numel <= 0 || d.numel % 32 || !d.x || !d.q || !d.s || ((uintptr_t)d.x & 15) || ((uintptr_t)d.q & 7))
      return pd_set_error(PD_ERR_INVALID_ARG, "pd_mx8_quantize_grouped: tensor %d: numel %% 32 == 0 and aligned non-null pointers required", i);
    h[i] = QtEntry{(const bf16_t *)d.x, d.q, d.s, pieces};
    pieces += d.numel / 8;
  }
  hipStream_t st = (hipStream_t)stream;
  if (hipMemcpyAsync(table_device, h, (size_t)count * sizeof(QtEntry), hipMemcpyHostToDevice, st) != hipSuccess)
    return pd_set_error(PD_ERR_LAUNCH, "pd_mx8_quantize_grouped: table upload failed");
  if (format == PD_MX8_E4M3) hipLaunchKernelGGL((mx8_quantize_grouped<PD_MX8_E4M3>), dim3(blocks_for(pieces)), dim3(256), 0, st, (const QtEntry *)table_device, count, pieces);
  else hipLaunchKernelGGL((mx8_quantize_grouped<PD_MX8_E5M2>), dim3(blocks_for(pieces)), dim3(256), 0, st, (const QtEntry *)table_device, count, pieces);
  return pd_check_launch("pd_mx8_quantize_grouped");
}
