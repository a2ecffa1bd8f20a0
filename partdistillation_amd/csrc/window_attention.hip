// Swin (shifted-)window attention for gfx950: 12 x 12 windows (144 tokens), head_dim 32, bf16 on
// v_mfma_f32_16x16x16_bf16_1k, fp32 softmax.  C-ABI and the math it replaces: include/pd_window_attention.h.
//
// One WORKGROUP of 9 wavefronts owns one (window, head) at a time: the 144 x 32 k / v (/ q / dO / O) tiles land in LDS once
// (row-major, written by global_load_lds, the next window's while this one is computed: see "tiles come global -> LDS" below) and wave w
// owns the w-th 16-row tile — 16 queries in the forward and in the dQ phase of the
// backward, 16 keys in its dK / dV phase — so every accumulator is complete inside one wave (no cross-wave reduction) and
// per-wave register state stays small enough for ~5 waves per SIMD (the first version ran one wave per (window, head)
// with ~400 VGPRs: one wave per SIMD, every MFMA / LDS / exp latency exposed, 10x slower).  The workgroup walks `chunk`
// windows of the same head so the bias table / its gradient are staged once.
//
// MFMA operand convention used below (16x16x16, lane = 16*g + c):
//     mma16(acc, x, y): acc[i][j] += sum_k X[i][k] * Y[j][k]
//     x: this lane holds X[row c][k = 4g .. 4g+3],   y: this lane holds Y[row c][k = 4g .. 4g+3]
//     acc: this lane holds acc[i = 4g + e][j = c], e = 0..3
// A lane therefore ends up with 4 CONSECUTIVE i for one j — which is again the "4 consecutive k" an operand needs, so
// softmax probabilities feed the next MFMA straight from registers:
//   forward, dQ phase   S^T = K.Q^T -> lane: query c, keys 4g+e  -> P / dS are the y-operands of O^T = Vt.P, dQ^T = Kt.dS
//   dK / dV phase       S   = Q.K^T -> lane: key c, queries 4g+e -> P / dS are the y-operands of dV^T = dOt.P, dK^T = Qt.dS
// The x-operands with a "t" need 4 consecutive ROWS of a row-major tile per lane: they come from the transposing LDS read ds_read_b64_tr_b16
// (below; through round 5 the backward spent an MFMA against the identity on each and the forward staged V transposed through registers,
// eight 2-byte LDS writes per thread).
//
// Relative-position bias: 4 consecutive tokens starting at a multiple of 4 lie in one row of the 12 x 12 window, so with
// A(t) = t + 11*(t/12) the 4 table indices a lane needs, A(q) - A(key) + 264, are consecutive: one address, 4 LDS reads.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mfma_bf16.h"
#include "mx8_quant.h"
#include "pd_common.h"
#include "pd_msda.h"
#include "pd_window_attention.h"

int g_pd_dbg_wattn = 0;   // tools/ only: 1 no table gradient at all (its own instantiation), 2 no global flush, 4 skip dK/dV phase, 8 skip dQ phase, 16 skip the table-gradient reduction; forward: 32 loads only, 64 no table staging

namespace {
using namespace pdmfma;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int N = 144, D = 32, NT = 9, TBL = 529, THREADS = 64 * NT;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float MASKED = -100.0f * LOG2E;    // the reference's additive -100 in base-2 units

__device__ __forceinline__ void mma16(f32x4 &c, bf16x4 x, bf16x4 y) { c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, c, 0, 0, 0); }
__device__ __forceinline__ int rel_a(int t) { return t + 11 * ((t * 171) >> 11); }            // t + 11*(t/12), t < 144
__device__ __forceinline__ void st4(bf16_t *p, float a, float b, float c, float d) { *reinterpret_cast<bf16x4 *>(p) = pack4(a, b, c, d); }
// The 32 channels of one (token, head) piece sit in four lanes (g = 0..3 of the same c), 8 values each (4 g .. 4 g + 3 and 16 + 4 g ..):
// exactly one block of the MX-fp8 operand format (include/pd_mx8.h), so the piece can leave as the next GEMM's operand next to its bf16
// copy — quantised from the bf16-rounded values, block maximum = two cross-lane steps.  q: the piece's 32 bytes, s: its exponent byte.
template <int FMT>
__device__ __forceinline__ void st_mx_piece(uint8_t *q, uint8_t *s, int g, const f32x4 &a, const f32x4 &b)
{
  const bf16x4 ra = pack4(a[0], a[1], a[2], a[3]), rb = pack4(b[0], b[1], b[2], b[3]);
  float v[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float((unsigned)(unsigned short)ra[e] << 16); v[4 + e] = __uint_as_float((unsigned)(unsigned short)rb[e] << 16); }
  float m = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));
  m = fmaxf(m, __shfl_xor(m, 16));
  m = fmaxf(m, __shfl_xor(m, 32));
  float mult;
  const unsigned byte = pdmx::mx_exponent<FMT>(m, mult);
  *reinterpret_cast<unsigned *>(q + 4 * g) = pdmx::mx_pack4<FMT>(v[0], v[1], v[2], v[3], mult);
  *reinterpret_cast<unsigned *>(q + 16 + 4 * g) = pdmx::mx_pack4<FMT>(v[4], v[5], v[6], v[7], mult);
  if (g == 0) *s = (uint8_t)byte;
}
__device__ __forceinline__ void st_mx_piece(int fmt, uint8_t *q, uint8_t *s, int g, const f32x4 &a, const f32x4 &b)
{
  if (fmt == PD_MX8_E4M3) st_mx_piece<PD_MX8_E4M3>(q, s, g, a, b);
  else st_mx_piece<PD_MX8_E5M2>(q, s, g, a, b);
}
__device__ __forceinline__ bf16x4 tr16(bf16x4 x, bf16x4 ident)                                // X[rows][16] -> X^T operand: acc = X . I^T (exact: the values are bf16)
{
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  mma16(a, x, ident);
  return pack4(a[0], a[1], a[2], a[3]);
}
__device__ __forceinline__ bf16x4 identity_operand(int c, int g)
{
  bf16x4 r;
#pragma unroll
  for (int m = 0; m < 4; ++m) r[m] = (4 * g + m == c) ? (short)0x3F80 : (short)0;
  return r;
}

// ---- tiles come global -> LDS DIRECTLY (global_load_lds_dwordx4: no staging registers, no wait until the data is needed), the next
// window's while this one is computed.  A [144][32] bf16 tile is 144 rows of 64 bytes, unpadded — the instruction writes lane i's 16
// bytes at base + 16 i — with the four 16-byte chunks of row r stored at slot chunk ^ swz(r), swz(r) = bit 3 of r | bit 2 of r << 1: the 8-byte operand
// reads of 16 consecutive rows (rows r, r + 4, r + 8, r + 12 share a 16-bank window: four different slots) and the transposing reads below
// (rows r, r + 4 of an 8-row group: slot pairs {0, 1} ^ swz differ) are both free of bank conflicts in the 64-bank model of the guide (the
// padded pitch-36 layout of the forward cannot be written by the instruction).  The transposed operands (4 consecutive ROWS of one column per
// lane) come from ds_read_b64_tr_b16 — lane (c, g) points at row 4 g + c / 4, elements 4 (c % 4) .. +3 and receives column c of rows 4 g .. 4 g + 3
// (mapping: tools/probes/tr_read_probe.hip) — where rounds 3-5 spent an MFMA against the identity + a repack on each (6 of 20 MFMAs per tile pair).  Round 6 measured the register-staged version: 21 of 91 us per launch at Swin-B's third stage were the exposed load + two
// barriers of each window (tools/bench_window_attention.py, ablation 13).
typedef __attribute__((address_space(3))) void *lds_ptr;
typedef __attribute__((address_space(1))) const void *glb_ptr;
constexpr int TILE_B = N * 64;                                   // bytes of one tile
constexpr int L_DELTA = 2176, L_LSE = 2752, L_REG = 7360, L_OT = 12288, L_TILES = L_OT + TILE_B, BWD_LDS = L_TILES + 2 * 4 * TILE_B;
constexpr int LSE_B = 9 * 64, REG_B = 9 * 256, FWD_TBL = 4 * TILE_B + 2 * REG_B, FWD_LDS = FWD_TBL + (TBL + 3) * 4;                   // per buffer: 64 floats / 256 bytes per 16-row block (its first 16 entries used)
__device__ __forceinline__ bf16x4 rd8(const unsigned char *p) { return *reinterpret_cast<const bf16x4 *>(p); }
__device__ __forceinline__ bf16x4 rdtr(const unsigned char *p)
{
  typedef __attribute__((address_space(3))) bf16x4 *lp;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)p);
}
__device__ __forceinline__ int swz(int row) { return ((row >> 3) & 1) | (((row >> 2) & 1) << 1); }
// byte offset inside a 16-row block of the 8-byte piece p (elements 4 p .. 4 p + 3) of row `row` (0..15)
__device__ __forceinline__ int piece_at(int row, int p) { return row * 64 + ((((p >> 1) ^ swz(row)) & 3) << 4) + ((p & 1) << 3); }

template <bool MASK>
__global__ __launch_bounds__(THREADS) void wattn_fwd(const bf16_t *__restrict__ qkv, const float *__restrict__ table,
                                                     const uint8_t *__restrict__ region, const uint8_t *__restrict__ flags,
                                                     bf16_t *__restrict__ out, float *__restrict__ lse, int B_, int nW,
                                                     int heads, float c1, int chunk, uint8_t *__restrict__ out_q, uint8_t *__restrict__ out_s, int qfmt, int ablate)
{
  // (dynamic LDS like the backward: with static arrays the compiler put s_waitcnt vmcnt(0) — the loads in flight for the NEXT window — in front
  // of the first plain read of this one)
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  unsigned char *kv = lds;                                                // [buffer][K, V] tiles, chunk-swizzled rows of 64 bytes
  unsigned char *reg_all = lds + 4 * TILE_B;                              // [buffer][REG_B]
  float *tbl = reinterpret_cast<float *>(lds + FWD_TBL);
  const int tid = threadIdx.x, lane = tid & 63, qt = __builtin_amdgcn_readfirstlane(tid >> 6), c = lane & 15, g = lane >> 4, h = blockIdx.y;
  const int C = heads * D;
  const int64_t ld = 3 * C;
  auto issue = [&](int b, int buf) {                                      // K and V of window b (+ its region labels): this lane's 16 bytes of each
    int lo = lane;
    asm volatile("" : "+v"(lo));
    const int lrow = 16 * qt + (lo >> 2), lch = (lo & 3) ^ swz(lo >> 2), l3 = lo < 3 ? lo : 3;
    const bf16_t *base = qkv + (int64_t)b * N * ld + h * D + lrow * (int)ld + lch * 8;
    unsigned char *t = kv + buf * 2 * TILE_B + qt * 1024;
    __builtin_amdgcn_global_load_lds((glb_ptr)(base + C), (lds_ptr)t, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr)(base + 2 * C), (lds_ptr)(t + TILE_B), 16, 0, 0);
    if (MASK) __builtin_amdgcn_global_load_lds((glb_ptr)(region + (int64_t)(b % nW) * N + 16 * qt + 4 * l3), (lds_ptr)(reg_all + buf * REG_B + qt * 256), 4, 0, 0);
  };
  const int off0 = piece_at(c, g), off1 = piece_at(c, 4 + g), toff0 = piece_at(4 * g + (c >> 2), c & 3), toff1 = piece_at(4 * g + (c >> 2), 4 + (c & 3));
  const int q = 16 * qt + c;
  const int b0 = blockIdx.x * chunk, n_units = min(chunk, B_ - b0);
  // this wave's query rows as MFMA operands come straight from memory, one window ahead like the tiles (loads issued BEFORE the tile loads of
  // the same window: vmcnt counts in order, so nothing in the compute part ever waits behind a tile load)
  bf16x4 qn0 = {0, 0, 0, 0}, qn1 = qn0;
  if (n_units > 0) {
    const bf16_t *qb = qkv + (int64_t)b0 * N * ld + h * D + q * ld;
    qn0 = *reinterpret_cast<const bf16x4 *>(qb + 4 * g); qn1 = *reinterpret_cast<const bf16x4 *>(qb + 16 + 4 * g);
    issue(b0, 0);
  }
  unsigned long long mbits = 0;
  if (MASK) mbits = __ballot(lane < n_units && flags[(b0 + (lane < n_units ? lane : 0)) % nW] != 0);
  if (!(ablate & 64)) for (int i = tid; i < TBL; i += THREADS) tbl[i] = table[i * heads + h] * LOG2E;

  for (int u = 0; u < n_units; ++u) {
    const int b = b0 + u, buf = u & 1;
    const unsigned char *ks = kv + buf * 2 * TILE_B, *vs = ks + TILE_B, *r = reg_all + buf * REG_B;
    __syncthreads();                                                      // window u is in LDS (and q in registers); window u - 1 is done with
    const bf16x4 q0 = qn0, q1 = qn1;
    if (u + 1 < n_units) {
      const bf16_t *qb = qkv + (int64_t)(b + 1) * N * ld + h * D + q * ld;
      qn0 = *reinterpret_cast<const bf16x4 *>(qb + 4 * g); qn1 = *reinterpret_cast<const bf16x4 *>(qb + 16 + 4 * g);
      issue(b + 1, buf ^ 1);
    }
    if (ablate & 32) continue;
    const bool masked = MASK && ((mbits >> u) & 1);
    const unsigned rq = masked ? r[qt * 256 + c] : 0u;
    f32x4 s[NT];
    int aq = rel_a(q) + 264 - 3, gt = g;
    asm volatile("" : "+v"(aq), "+v"(gt));
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      mma16(s[kt], rd8(ks + kt * 1024 + off0), q0);                       // S^T[key = 4g+e][q = c]
      mma16(s[kt], rd8(ks + kt * 1024 + off1), q1);
    }
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      const float *tb = tbl + (aq - rel_a(16 * kt + 4 * gt));             // index for key0+e is tb[3 - e]
      s[kt][0] = fmaf(s[kt][0], c1, tb[3]); s[kt][1] = fmaf(s[kt][1], c1, tb[2]);
      s[kt][2] = fmaf(s[kt][2], c1, tb[1]); s[kt][3] = fmaf(s[kt][3], c1, tb[0]);
      if (masked) {
        const unsigned rk = *reinterpret_cast<const unsigned *>(r + kt * 256 + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) if (((rk >> (8 * e)) & 255u) != rq) s[kt][e] += MASKED;
      }
      m = fmaxf(fmaxf(m, fmaxf(s[kt][0], s[kt][1])), fmaxf(s[kt][2], s[kt][3]));
    }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[kt][e] = __builtin_amdgcn_exp2f(s[kt][e] - m); sum += s[kt][e]; }
      const bf16x4 p = pack4(s[kt][0], s[kt][1], s[kt][2], s[kt][3]);
      mma16(o0, rdtr(vs + kt * 1024 + toff0), p);                         // O^T[d = 4g+e][q = c]: V^T read transposed out of the row-major tile
      mma16(o1, rdtr(vs + kt * 1024 + toff1), p);
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.f / sum;
    bf16_t *o = out + ((int64_t)b * N + q) * C + h * D + 4 * g;
    o0 *= inv; o1 *= inv;
    st4(o, o0[0], o0[1], o0[2], o0[3]);
    st4(o + 16, o1[0], o1[1], o1[2], o1[3]);
    if (out_q) st_mx_piece(qfmt, out_q + ((int64_t)b * N + q) * C + h * D, out_s + ((int64_t)b * N + q) * (C / 32) + h, g, o0, o1);
    if (g == 0) lse[((int64_t)b * heads + h) * N + q] = m + log2f(sum);
  }
}

template <bool MASK, int ABL>
__global__ __launch_bounds__(THREADS) void wattn_bwd(const bf16_t *__restrict__ qkv, const float *__restrict__ table,
                                                     const uint8_t *__restrict__ region, const uint8_t *__restrict__ flags,
                                                     const bf16_t *__restrict__ out, const bf16_t *__restrict__ dout,
                                                     const float *__restrict__ lse, bf16_t *__restrict__ dqkv,
                                                     float *__restrict__ dtable, int B_, int nW, int heads, float scale,
                                                     float c1, int chunk, int ablate, uint8_t *__restrict__ dq_q, uint8_t *__restrict__ dq_s, int qfmt)
{
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  float *tbl = reinterpret_cast<float *>(lds);
  float *delta_s = reinterpret_cast<float *>(lds + L_DELTA), *lse_all = reinterpret_cast<float *>(lds + L_LSE);
  unsigned char *reg_all = lds + L_REG, *otile = lds + L_OT, *tiles = lds + L_TILES;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), c = lane & 15, g = lane >> 4, h = blockIdx.y;   // (wv in a scalar register: LDS bases of the wave's blocks stay out of the vector registers)
  const int C = heads * D;
  const int64_t ld = 3 * C;
  for (int i = tid; i < TBL; i += THREADS) tbl[i] = table[i * heads + h] * LOG2E;
  // table gradient: a lane meets the same 36 (query, key) pairs in every window, so it sums dS over the `chunk` windows in
  // registers; the pairs -> table-entry reduction happens once per workgroup, without atomics (below).  ds_add_f32 costs
  // ~175 cycles per wave instruction on gfx950 whatever the addresses: per-window LDS atomics were 60 % of the kernel.
  f32x4 dacc[NT];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) dacc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this lane's piece of every tile: LDS slot lane & 3 of row 16 wv + lane / 4 holds chunk slot ^ swz(row) of that row.  (The offsets
  // are recomputed from an opaque lane id at every issue: kept across the phases they were the three spilled 64-bit values of the kernel.)
  auto issue = [&](int b, int buf) {
    int lo = lane;
    asm volatile("" : "+v"(lo));
    const int lrow = 16 * wv + (lo >> 2), lch = (lo & 3) ^ swz(lo >> 2);
    const int go_qkv = lrow * (int)ld + lch * 8, go_o = lrow * C + lch * 8, l15 = lo < 15 ? lo : 15, l3 = lo < 3 ? lo : 3;
    const bf16_t *base = qkv + (int64_t)b * N * ld + h * D + go_qkv;
    const bf16_t *dob = dout + (int64_t)b * N * C + h * D + go_o, *ob = out + (int64_t)b * N * C + h * D + go_o;
    unsigned char *t = tiles + buf * 4 * TILE_B + wv * 1024;
    __builtin_amdgcn_global_load_lds((glb_ptr)base, (lds_ptr)t, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr)(base + C), (lds_ptr)(t + TILE_B), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr)(base + 2 * C), (lds_ptr)(t + 2 * TILE_B), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr)dob, (lds_ptr)(t + 3 * TILE_B), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr)ob, (lds_ptr)(otile + wv * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr)(lse + ((int64_t)b * heads + h) * N + 16 * wv + l15), (lds_ptr)(lse_all + buf * LSE_B + wv * 64), 4, 0, 0);
    if (MASK) __builtin_amdgcn_global_load_lds((glb_ptr)(region + (int64_t)(b % nW) * N + 16 * wv + 4 * l3), (lds_ptr)(reg_all + buf * REG_B + wv * 256), 4, 0, 0);
  };
  const int b0 = blockIdx.x * chunk, n_units = min(chunk, B_ - b0);
  if (n_units > 0) issue(b0, 0);
  // which of this workgroup's windows carry a mask (chunk <= 8): one vote in the prologue instead of a dependent global load per window
  unsigned long long mbits = 0;
  if (MASK) mbits = __ballot(lane < n_units && flags[(b0 + (lane < n_units ? lane : 0)) % nW] != 0);
  for (int u = 0; u < n_units; ++u) {
    const int b = b0 + u, buf = u & 1;
    const unsigned char *tb = tiles + buf * 4 * TILE_B;
    const float *lse_s = lse_all + buf * LSE_B;
    const unsigned char *r = reg_all + buf * REG_B;
    bf16_t *dbase = dqkv + (int64_t)b * N * ld + h * D;
    __syncthreads();                                                      // (waits for this wavefront's loads, then for everyone's): window u is in LDS; window u - 1 is done with.
                                                                          // (Its vmcnt(0) also waits for the dQ stores just issued: holding them back until after the barriers measured no gain.)
    {                                                                     // delta = rowsum(dO * O): 4 threads per row, the same slot of both tiles
      const uint4 a = *reinterpret_cast<const uint4 *>(tb + 3 * TILE_B + tid * 16), o = *reinterpret_cast<const uint4 *>(otile + tid * 16);
      float acc = bf_lo(a.x) * bf_lo(o.x) + bf_hi(a.x) * bf_hi(o.x) + bf_lo(a.y) * bf_lo(o.y) + bf_hi(a.y) * bf_hi(o.y)
                + bf_lo(a.z) * bf_lo(o.z) + bf_hi(a.z) * bf_hi(o.z) + bf_lo(a.w) * bf_lo(o.w) + bf_hi(a.w) * bf_hi(o.w);
      acc += __shfl_xor(acc, 1);
      acc += __shfl_xor(acc, 2);
      if ((tid & 3) == 0) delta_s[tid >> 2] = acc;
    }
    const bool masked = MASK && ((mbits >> u) & 1);
    __syncthreads();
    const unsigned char *qs = tb, *ks = tb + TILE_B, *vs = tb + 2 * TILE_B, *dos = tb + 3 * TILE_B;
    // operand reads: row 16 kt + c, pieces g and 4 + g;  transposed operand reads: this lane's address is row 16 kt + 4 g + c / 4, pieces c % 4 and
    // 4 + c % 4.  (Rebuilt per window from opaque lane ids: carried across the windows they were spilled and reloaded behind the barrier.)
    int cu = c, gu = g;
    asm volatile("" : "+v"(cu), "+v"(gu));
    const int off0 = piece_at(cu, gu), off1 = piece_at(cu, 4 + gu), toff0 = piece_at(4 * gu + (cu >> 2), cu & 3), toff1 = piece_at(4 * gu + (cu >> 2), 4 + (cu & 3));

    if (!(ablate & 4)) {                                                  // ---- phase 1: this wave's 16 KEYS -> dK, dV
      const int key = 16 * wv + c;
      const bf16x4 k0 = rd8(ks + wv * 1024 + off0), k1 = rd8(ks + wv * 1024 + off1);
      const bf16x4 v0 = rd8(vs + wv * 1024 + off0), v1 = rd8(vs + wv * 1024 + off1);
      f32x4 dk0 = {0.f, 0.f, 0.f, 0.f}, dk1 = dk0, dv0 = dk0, dv1 = dk0;
      // (lane constants the nine table addresses derive from, made opaque per unit: hoisted out of the unit loop they cost ~20 registers
      // the kernel does not have — 24 spilled dwords reloaded inside the phases, 10 us per launch at Swin-B's third stage)
      int ak = rel_a(key) - 264, gt = g;
      asm volatile("" : "+v"(ak), "+v"(gt));
      const unsigned rk = masked ? r[wv * 256 + c] : 0u;
#pragma unroll 3
      for (int qt = 0; qt < NT; ++qt) {
        const bf16x4 qa0 = rd8(qs + qt * 1024 + off0), qa1 = rd8(qs + qt * 1024 + off1);
        const bf16x4 da0 = rd8(dos + qt * 1024 + off0), da1 = rd8(dos + qt * 1024 + off1);
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = s;
        mma16(s, qa0, k0);                                                // S[q = 4g+e][key = c]
        mma16(s, qa1, k1);
        mma16(dp, da0, v0);
        mma16(dp, da1, v1);
        const bf16x4 dot0 = rdtr(dos + qt * 1024 + toff0), dot1 = rdtr(dos + qt * 1024 + toff1);      // dO^T, Q^T: rows 4 g .. 4 g + 3 of column c
        const bf16x4 qt0 = rdtr(qs + qt * 1024 + toff0), qt1 = rdtr(qs + qt * 1024 + toff1);
        const float *tbp = tbl + (rel_a(16 * qt + 4 * gt) - ak);          // table index of (q0 + e, key) is tbp[e]
        const f32x4 L = *reinterpret_cast<const f32x4 *>(lse_s + qt * 64 + 4 * g), Dl = *reinterpret_cast<const f32x4 *>(delta_s + 16 * qt + 4 * g);
        unsigned rq = 0;
        if (masked) rq = *reinterpret_cast<const unsigned *>(r + qt * 256 + 4 * g);
        float p[4], ds[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = fmaf(s[e], c1, tbp[e]);
          if (masked && ((rq >> (8 * e)) & 255u) != rk) t += MASKED;
          p[e] = __builtin_amdgcn_exp2f(t - L[e]);
          ds[e] = p[e] * (dp[e] - Dl[e]) * scale;
        }
        const bf16x4 pb = pack4(p[0], p[1], p[2], p[3]), dsb = pack4(ds[0], ds[1], ds[2], ds[3]);
        mma16(dv0, dot0, pb);                                             // dV^T[d = 4g+e][key = c]
        mma16(dv1, dot1, pb);
        mma16(dk0, qt0, dsb);
        mma16(dk1, qt1, dsb);
      }
      bf16_t *dkp = dbase + C + key * ld + 4 * g, *dvp = dbase + 2 * C + key * ld + 4 * g;
      st4(dkp, dk0[0], dk0[1], dk0[2], dk0[3]);
      st4(dkp + 16, dk1[0], dk1[1], dk1[2], dk1[3]);
      st4(dvp, dv0[0], dv0[1], dv0[2], dv0[3]);
      st4(dvp + 16, dv1[0], dv1[1], dv1[2], dv1[3]);
      if (dq_q) {                                                         // dqkv again as an MX operand: rows of 3 C elements, 3 C / 32 exponents
        const int64_t row = (int64_t)b * N + key;
        st_mx_piece(qfmt, dq_q + row * ld + C + h * D, dq_s + row * (ld / 32) + C / 32 + h, g, dk0, dk1);
        st_mx_piece(qfmt, dq_q + row * ld + 2 * C + h * D, dq_s + row * (ld / 32) + 2 * (C / 32) + h, g, dv0, dv1);
      }
    }
    // The next window's loads go out HERE, between the phases: the compiler puts s_waitcnt vmcnt(0) in front of the first transposing read that
    // follows a global -> LDS load (it cannot tell the buffers apart; seen in the ISA) — issued before phase 1, the loads were waited for at its
    // first read.  Phase 2 therefore takes K^T the old way, by an MFMA against the identity from the plain fragments, and has no such read.
    if (u + 1 < n_units) issue(b + 1, buf ^ 1);                           // (the O tile is free again after delta; the other buffer since the last barrier)
    if (!(ablate & 8)) {                                                  // ---- phase 2: this wave's 16 QUERIES -> dQ, dtable
      const int q = 16 * wv + c;
      const bf16x4 q0 = rd8(qs + wv * 1024 + off0), q1 = rd8(qs + wv * 1024 + off1);
      const bf16x4 d0 = rd8(dos + wv * 1024 + off0), d1 = rd8(dos + wv * 1024 + off1);
      const float L = lse_s[wv * 64 + c], Dl = delta_s[q];
      int aq = rel_a(q) + 264 - 3, gt = g, ct = c;
      asm volatile("" : "+v"(aq), "+v"(gt), "+v"(ct));
      const bf16x4 ident = identity_operand(ct, gt);                      // (rebuilt per window: two registers the phases do not have to carry)
      const unsigned rq = masked ? r[wv * 256 + c] : 0u;
      f32x4 dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = dq0;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        const bf16x4 ka0 = rd8(ks + kt * 1024 + off0), ka1 = rd8(ks + kt * 1024 + off1);
        const bf16x4 va0 = rd8(vs + kt * 1024 + off0), va1 = rd8(vs + kt * 1024 + off1);
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = s;
        mma16(s, ka0, q0);                                                // S^T[key = 4g+e][q = c]
        mma16(s, ka1, q1);
        mma16(dp, va0, d0);
        mma16(dp, va1, d1);
        const bf16x4 kt0 = tr16(ka0, ident), kt1 = tr16(ka1, ident);
        const int ti = aq - rel_a(16 * kt + 4 * gt);                      // table index of (q, key0 + e) is ti + 3 - e
        unsigned rk = 0;
        if (masked) rk = *reinterpret_cast<const unsigned *>(r + kt * 256 + 4 * g);
        float ds[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = fmaf(s[e], c1, tbl[ti + 3 - e]);
          if (masked && ((rk >> (8 * e)) & 255u) != rq) t += MASKED;
          const float p = __builtin_amdgcn_exp2f(t - L);
          ds[e] = p * (dp[e] - Dl);
        }
        const bf16x4 dsb = pack4(ds[0] * scale, ds[1] * scale, ds[2] * scale, ds[3] * scale);
        mma16(dq0, kt0, dsb);                                             // dQ^T[d = 4g+e][q = c]
        mma16(dq1, kt1, dsb);
        dacc[kt][0] += ds[0]; dacc[kt][1] += ds[1]; dacc[kt][2] += ds[2]; dacc[kt][3] += ds[3];
      }
      bf16_t *dqp = dbase + q * ld + 4 * g;
      st4(dqp, dq0[0], dq0[1], dq0[2], dq0[3]);
      st4(dqp + 16, dq1[0], dq1[1], dq1[2], dq1[3]);
      if (dq_q) st_mx_piece(qfmt, dq_q + ((int64_t)b * N + q) * ld + h * D, dq_s + ((int64_t)b * N + q) * (ld / 32) + h, g, dq0, dq1);
    }
  }
  if (ABL != 1 && !(ablate & 16)) {
    // sum of dS over this workgroup's windows, [144 queries][144 keys] fp32, goes through the (now dead) tile space 48
    // query rows = 4 rows of the 12 x 12 grid at a time; thread i < 529 owns table entry i = (dr + 11) * 23 + (dc + 11)
    // and adds up the pairs (rq, cq) -> (rq - dr, cq - dc) it is made of: every element is read exactly once
    float *red = reinterpret_cast<float *>(tiles);                           // (no load is in flight: the last window issued none)
    const int dr = tid / 23 - 11, dc = tid % 23 - 11;
    const int c_lo = dc > 0 ? dc : 0, c_hi = dc < 0 ? 12 + dc : 12;
    float tsum = 0.f;
    for (int r = 0; r < 3; ++r) {
      __syncthreads();
      if (wv / 3 == r) {
        float *row = red + (16 * (wv - 3 * r) + c) * N + 4 * g;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) *reinterpret_cast<f32x4 *>(row + 16 * kt) = dacc[kt];
      }
      __syncthreads();
      if (tid < TBL) {
#pragma unroll
        for (int rl = 0; rl < 4; ++rl) {
          const int rk = 4 * r + rl - dr;
          const bool row_ok = rk >= 0 && rk < 12;
          const float *src = red + 12 * rl * N + 12 * (row_ok ? rk : 0) - dc;
#pragma unroll
          for (int cq = 0; cq < 12; ++cq) {                                 // fixed trip count: the 12 loads pipeline
            const bool ok = row_ok && cq >= c_lo && cq < c_hi;
            const float v = src[ok ? cq * (N + 1) : c_lo * (N + 1)];
            tsum += ok ? v : 0.f;
          }
        }
      }
    }
    if (tid < TBL && !(ablate & 2)) atomicAdd(dtable + tid * heads + h, tsum);
  }
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

// windows one workgroup walks (<= 8), chosen to minimise (rounds of workgroups over the 256 CUs) x (windows + the
// per-workgroup fixed cost in window units: table staging, and in the backward the table-gradient reduction, which costs
// about as much as one window)
int chunk_of(int B_, int heads, float fixed_cost, int wgs_per_cu)
{
  int best = 1;
  float best_cost = 1e30f;
  for (int chunk = 1; chunk <= 8; ++chunk) {
    const int64_t wgs = (int64_t)heads * ((B_ + chunk - 1) / chunk);
    const int64_t rounds = (wgs + 256 * wgs_per_cu - 1) / (256 * wgs_per_cu);
    const float cost = rounds * (chunk + fixed_cost);
    if (cost < best_cost) { best_cost = cost; best = chunk; }
  }
  return best;
}
}  // namespace

extern "C" int pd_window_attn_fwd_w12(const void *qkv, const float *table, const uint8_t *region, const uint8_t *win_flags,
                                      void *out, float *lse, int B_, int nW, int heads, float scale, void *out_q, void *out_s, int q_format,
                                      void *stream_)
{
  if ((out_q != nullptr) != (out_s != nullptr) || (out_q && q_format != PD_MX8_E4M3 && q_format != PD_MX8_E5M2) || ((uintptr_t)out_q & 3))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_window_attn_fwd_w12: the MX copy needs both pointers (4-byte aligned elements) and a known format");
  int rc = check(qkv, table, region, win_flags, B_, nW, heads, "pd_window_attn_fwd_w12");
  if (rc || B_ == 0) return rc;
  if (!out || !lse) return pd_set_error(PD_ERR_INVALID_ARG, "pd_window_attn_fwd_w12: null output");
  const int chunk = chunk_of(B_, heads, 0.2f, 2);
  const dim3 grid((B_ + chunk - 1) / chunk, heads);
  hipStream_t s = (hipStream_t)stream_;
  if (region) hipLaunchKernelGGL(wattn_fwd<true>, grid, dim3(THREADS), FWD_LDS, s, (const bf16_t *)qkv, table, region, win_flags, (bf16_t *)out, lse, B_, nW, heads, scale * LOG2E, chunk, (uint8_t *)out_q, (uint8_t *)out_s, q_format, g_pd_dbg_wattn);
  else hipLaunchKernelGGL(wattn_fwd<false>, grid, dim3(THREADS), FWD_LDS, s, (const bf16_t *)qkv, table, region, win_flags, (bf16_t *)out, lse, B_, nW, heads, scale * LOG2E, chunk, (uint8_t *)out_q, (uint8_t *)out_s, q_format, g_pd_dbg_wattn);
  return pd_check_launch("pd_window_attn_fwd_w12");
}

extern "C" int pd_window_attn_bwd_w12(const void *qkv, const float *table, const uint8_t *region, const uint8_t *win_flags,
                                      const void *out, const void *d_out, const float *lse, void *dqkv, float *dtable, int B_,
                                      int nW, int heads, float scale, void *dqkv_q, void *dqkv_s, int q_format, void *stream_)
{
  if ((dqkv_q != nullptr) != (dqkv_s != nullptr) || (dqkv_q && q_format != PD_MX8_E4M3 && q_format != PD_MX8_E5M2) || ((uintptr_t)dqkv_q & 3))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_window_attn_bwd_w12: the MX copy needs both pointers (4-byte aligned elements) and a known format");
  int rc = check(qkv, table, region, win_flags, B_, nW, heads, "pd_window_attn_bwd_w12");
  if (rc || B_ == 0) return rc;
  if (!out || !d_out || !lse || !dqkv || !dtable) return pd_set_error(PD_ERR_INVALID_ARG, "pd_window_attn_bwd_w12: null pointer");
  if ((((uintptr_t)out | (uintptr_t)d_out | (uintptr_t)dqkv) & 15) != 0)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_window_attn_bwd_w12: out / d_out / dqkv must be 16-byte aligned");
  const int chunk = chunk_of(B_, heads, 1.0f, 1);
  const dim3 grid((B_ + chunk - 1) / chunk, heads);
  hipStream_t s = (hipStream_t)stream_;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)wattn_bwd<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS);
    (void)hipFuncSetAttribute((const void *)wattn_bwd<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS);
    (void)hipFuncSetAttribute((const void *)wattn_bwd<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, BWD_LDS);
    attr = true;
  }
  if (g_pd_dbg_wattn & 1) hipLaunchKernelGGL((wattn_bwd<false, 1>), grid, dim3(THREADS), BWD_LDS, s, (const bf16_t *)qkv, table, region, win_flags, (const bf16_t *)out, (const bf16_t *)d_out, lse, (bf16_t *)dqkv, dtable, B_, nW, heads, scale, scale * LOG2E, chunk, g_pd_dbg_wattn, (uint8_t *)dqkv_q, (uint8_t *)dqkv_s, q_format);
  else if (region) hipLaunchKernelGGL((wattn_bwd<true, 0>), grid, dim3(THREADS), BWD_LDS, s, (const bf16_t *)qkv, table, region, win_flags, (const bf16_t *)out, (const bf16_t *)d_out, lse, (bf16_t *)dqkv, dtable, B_, nW, heads, scale, scale * LOG2E, chunk, g_pd_dbg_wattn, (uint8_t *)dqkv_q, (uint8_t *)dqkv_s, q_format);
  else hipLaunchKernelGGL((wattn_bwd<false, 0>), grid, dim3(THREADS), BWD_LDS, s, (const bf16_t *)qkv, table, region, win_flags, (const bf16_t *)out, (const bf16_t *)d_out, lse, (bf16_t *)dqkv, dtable, B_, nW, heads, scale, scale * LOG2E, chunk, g_pd_dbg_wattn, (uint8_t *)dqkv_q, (uint8_t *)dqkv_s, q_format);
  return pd_check_launch("pd_window_attn_bwd_w12");
}
