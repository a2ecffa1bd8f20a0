// Fused query-side decoder layer kernels (gfx950): C-ABI in include/pd_declayer.h.
//
// A workgroup of 8 wavefronts owns RB = 16 rows of the [R, 256] query tensor for a chain of row-local operators.  Every product is
//     Y^T[n][m] = sum_k W[n][k] X[m][k]      on v_mfma_f32_16x16x32_bf16 with  A = a 16 x 32 block of W  (lane: row n = lane & 15,
//                                             8 consecutive k from 8 (lane >> 4)), loaded STRAIGHT from global memory (16 bytes per lane)
//                                             out of a PACKED copy of the weight in which that operand is 1 KB contiguous (dec_pack_grouped),
//                                             B = the rows' activations from LDS (lane: row m = lane & 15, the same 8 k),
// so a lane ends with 4 consecutive output channels n = 4 (lane >> 4) .. + 3 of row m = lane & 15 (row-major 8-byte pieces).  A wavefront
// owns 32 of the 256 output columns; its "weight block" = 32 rows x 256 k = 16 x 16 bytes per lane is the unit of the software pipeline:
// the block after the one being multiplied is always in flight, across phase boundaries too (weight addresses depend on nothing).  The
// kernels are bound by that stream (a workgroup pulls every weight of its chain once: 0.5 .. 2.7 MB), not by the matrix pipe: read in the
// operand's row-major pattern (16 rows x 64 bytes per wave instruction) it ran at 30-36 GB/s per workgroup, from the packed copy at
// ~108 GB/s (pd_dec_fwd_b 79 -> 25.5 us; tools/probes/stream_probe.hip).
// LayerNorm (forward and backward) runs wave-per-row on the fp32 rows in LDS with the DPP reductions of csrc/rowwise.hip, so its sums are
// taken in the same order as there.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "pd_common.h"
#include "pd_msda.h"
#include "pd_declayer.h"

namespace {

typedef unsigned short bf16_t;
typedef __bf16 hwbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hwbf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int C = 256, FF = 2048, RB = 16, NTH = 512, NW = 8;
constexpr int PA = C + 8;        // bf16 elements per LDS row of a [16][256] tile (528 B)
constexpr int PZ = C + 4;        // floats per LDS row of an fp32 [16][256] tile

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi)
{
  const f32x2 x = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(x, hwbf16x2));
}
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) { return make_uint2(pk_bf16(a, b), pk_bf16(c, d)); }
// the value a bf16 store + reload would give (round to nearest even)
__device__ __forceinline__ float rbf(float x) { return bf_lo(pk_bf16(x, 0.f)); }

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v)            // same order as csrc/rowwise.hip
{
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

__device__ __forceinline__ void zero_acc(f32x4 (&acc)[2])
{
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
}

// the 4 values of tile t a lane holds: row m = lane & 15, columns nb + 16 t + 4 (lane >> 4) .. + 3
template <bool BIAS, bool RELU>
__device__ __forceinline__ void epi4(const f32x4 &a, const bf16_t *bias_lds, int n, float (&v)[4])
{
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = a[r];
  if (BIAS) {
    const uint2 b = *reinterpret_cast<const uint2 *>(bias_lds + n);
    v[0] += bf_lo(b.x); v[1] += bf_hi(b.x); v[2] += bf_lo(b.y); v[3] += bf_hi(b.y);
  }
  if (RELU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
  }
}
// product result -> bf16 [16][pitch] LDS tile
template <bool BIAS, bool RELU>
__device__ __forceinline__ void store_tile_bf16(const f32x4 (&acc)[2], const bf16_t *bias_lds, int nb_bias, bf16_t *dst, int pitch, int nb, int lane)
{
  const int m = lane & 15, g = lane >> 4;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float v[4];
    epi4<BIAS, RELU>(acc[t], bias_lds, nb_bias + 16 * t + 4 * g, v);
    *reinterpret_cast<uint2 *>(dst + m * pitch + nb + 16 * t + 4 * g) = pack4(v[0], v[1], v[2], v[3]);
  }
}
// product result, rounded to bf16 like the unfused GEMM's output, widened again -> fp32 [16][PZ] LDS tile
template <bool BIAS>
__device__ __forceinline__ void store_tile_f32r(const f32x4 (&acc)[2], const bf16_t *bias_lds, int nb_bias, float *dst, int nb, int lane)
{
  const int m = lane & 15, g = lane >> 4;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float v[4];
    epi4<BIAS, false>(acc[t], bias_lds, nb_bias + 16 * t + 4 * g, v);
    st4(dst + m * PZ + nb + 16 * t + 4 * g, make_float4(rbf(v[0]), rbf(v[1]), rbf(v[2]), rbf(v[3])));
  }
}

// [16][256] bf16 rows: global -> LDS tile (one 16-byte piece per thread); rows past R read row R - 1 (never stored back)
__device__ __forceinline__ void tile_in(bf16_t *dst, const bf16_t *__restrict__ src, int r0, int R, int tid)
{
  const int row = tid >> 5, col = (tid & 31) * 8;
  const uint4 v = *reinterpret_cast<const uint4 *>(src + (size_t)min(r0 + row, R - 1) * C + col);
  *reinterpret_cast<uint4 *>(dst + row * PA + col) = v;
}
// LDS tile -> global rows (16-byte pieces, 512 B per row contiguous)
__device__ __forceinline__ void tile_out(bf16_t *__restrict__ dst, const bf16_t *src, int r0, int R, int tid)
{
  const int row = tid >> 5, col = (tid & 31) * 8;
  if (r0 + row < R) *reinterpret_cast<uint4 *>(dst + (size_t)(r0 + row) * C + col) = *reinterpret_cast<const uint4 *>(src + row * PA + col);
}
__device__ __forceinline__ void lds_copy_bf16(bf16_t *dst, const bf16_t *__restrict__ src, int n, int tid)       // n % 8 == 0
{
  for (int i = tid * 8; i < n; i += NTH * 8) *reinterpret_cast<uint4 *>(dst + i) = *reinterpret_cast<const uint4 *>(src + i);
}
__device__ __forceinline__ void lds_copy_f32(float *dst, const float *__restrict__ src, int n, int tid)          // n % 4 == 0
{
  for (int i = tid * 4; i < n; i += NTH * 4) st4(dst + i, ld4(src + i));
}

// ---- LayerNorm forward of one row held as a float4 per lane (channels 4 lane .. + 3): -> normalised * gamma + beta
__device__ __forceinline__ float4 ln_row(float4 v, float4 gm, float4 bt, float eps, float &mu, float &rs)
{
  const float s = (v.x + v.y) + (v.z + v.w);
  mu = wave_sum(s) * (1.f / C);
  v.x -= mu; v.y -= mu; v.z -= mu; v.w -= mu;
  const float q = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  rs = rsqrtf(wave_sum(q) * (1.f / C) + eps);
  return make_float4(v.x * rs * gm.x + bt.x, v.y * rs * gm.y + bt.y, v.z * rs * gm.z + bt.z, v.w * rs * gm.w + bt.w);
}
// ---- LayerNorm backward of one row: t = incoming gradient, z = the forward's input row; ag / ab += t xhat / t (dgamma / dbeta)
__device__ __forceinline__ float4 ln_row_bwd(float4 t, float4 z, float mu, float rs, float4 gm, float4 &ag, float4 &ab)
{
  float4 h = make_float4((z.x - mu) * rs, (z.y - mu) * rs, (z.z - mu) * rs, (z.w - mu) * rs);
  ag.x += t.x * h.x; ag.y += t.y * h.y; ag.z += t.z * h.z; ag.w += t.w * h.w;
  ab = add4(ab, t);
  t.x *= gm.x; t.y *= gm.y; t.z *= gm.z; t.w *= gm.w;
  const float s1 = (t.x + t.y) + (t.z + t.w);
  const float s2 = (t.x * h.x + t.y * h.y) + (t.z * h.z + t.w * h.w);
  const float m1 = wave_sum(s1) * (1.f / C), m2 = wave_sum(s2) * (1.f / C);
  return make_float4(rs * (t.x - m1 - h.x * m2), rs * (t.y - m1 - h.y * m2), rs * (t.z - m1 - h.z * m2), rs * (t.w - m1 - h.w * m2));
}
__device__ __forceinline__ float4 bf4(const bf16_t *p)
{
  const uint2 u = *reinterpret_cast<const uint2 *>(p);
  return make_float4(bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y));
}
__device__ __forceinline__ void st_bf4(bf16_t *p, float4 v) { *reinterpret_cast<uint2 *>(p) = pack4(v.x, v.y, v.z, v.w); }

// column sums of the 8 wavefronts' partials (red[wave][k][256]) -> atomics; k-th array goes to outs[k] (nullptr: skipped)
template <int NARR>
__device__ __forceinline__ void flush_colsums(const float *red, float *const (&outs)[NARR], int tid)
{
  for (int i = tid; i < NARR * C; i += NTH) {
    const int k = i / C, c = i - k * C;
    if (!outs[k]) continue;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[(w * NARR + k) * C + c];
    atomicAdd(outs[k] + c, s);
  }
}

// ---- packed weights: a wavefront's block (32 rows x 256 k) is 16 KB contiguous, instruction i = 8 t + s reads 1 KB (pd_dec_pack_grouped)
constexpr int BLK = 8192;                                           // bf16 elements per block
// The pipeline's unit is HALF a block (its 128 first / last contraction elements: 8 loads); a wavefront keeps THREE halves in flight
// beside the one it multiplies (four 32-register buffers, 24 KB per wavefront on the way instead of 16 with whole blocks one ahead).
struct WHalf { uint4 v[2][4]; };
__device__ __forceinline__ void wloadh(WHalf &w, const bf16_t *__restrict__ blk, int half, int lane)
{
  const bf16_t *p = blk + lane * 8 + half * (4 * 512);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s) w.v[t][s] = *reinterpret_cast<const uint4 *>(p + (t * 8 + s) * 512);
}
__device__ __forceinline__ void wmmah(f32x4 (&acc)[2], const WHalf &w, const bf16_t *xs)
{
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const hwbf16x8 x = *reinterpret_cast<const hwbf16x8 *>(xs + 32 * s);
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(hwbf16x8, w.v[0][s]), x, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(hwbf16x8, w.v[1][s]), x, acc[1], 0, 0, 0);
  }
}
// half h of a wavefront's block sequence (bp(b) = base of its b-th block, NBLK blocks): issue / multiply; indices are compile-time at every use
#define PD_HP(h) do { if ((h) < 2 * NBLK) wloadh(w[(h) & 3], bp((h) >> 1), (h) & 1, lane); } while (0)
#define PD_BLK(b, XS) do { PD_HP(2 * (b) + 3); wmmah(acc, w[(2 * (b)) & 3], (XS)); PD_HP(2 * (b) + 4); wmmah(acc, w[(2 * (b) + 1) & 3], (XS) + 128); } while (0)

// ---- the FFN cut in two launches (PART 1 / PART 2 of dec_fwd_b / dec_bwd_b): PART 1 runs P workgroups per row block, each with 1 / P of the
// hidden columns, and leaves fp32 partial rows [16][256] in its slab; PART 2 (one workgroup per row block, the NEXT launch: the kernel
// boundary is the synchronisation) sums the slabs in slab order.  An in-launch seam (ticket + last arriver) measured 33 / 38 / 56 us at
// P = 2 / 4 / 8 against 25.5 for one workgroup per row block; two launches pay one boundary (~1.5 us) instead.
__device__ __forceinline__ void slab_store(float *__restrict__ slab, const f32x4 (&acc)[2], int nb, int lane)
{
  const int m = lane & 15, g = lane >> 4;
#pragma unroll
  for (int t = 0; t < 2; ++t) st4(slab + m * C + nb + 16 * t + 4 * g, make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]));
}
__device__ __forceinline__ float4 slab_sum(const float *__restrict__ slabs, int P, int row, int lane)
{
  float4 s = zero4();
  for (int p = 0; p < P; ++p) s = add4(s, ld4(slabs + (size_t)p * (RB * C) + row * C + 4 * lane));
  return s;
}

// =================================================================================================== forward A
struct FwdA {
  const bf16_t *o; const float *res, *qpos; int pos_div;
  const bf16_t *w_o, *b_o; const float *ln_w, *ln_b; float eps;
  const bf16_t *w_qkv, *b_qkv;
  float *z, *stats, *y; bf16_t *y_c, *ypos_c, *q, *k, *v; int R;
};

__global__ __launch_bounds__(NTH) void dec_fwd_a(const FwdA a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t *X0 = reinterpret_cast<bf16_t *>(smem);                  // o rows, later q
  bf16_t *X1 = X0 + RB * PA;                                       // y_c
  bf16_t *X2 = X1 + RB * PA;                                       // ypos_c
  bf16_t *X3 = X2 + RB * PA;                                       // k
  bf16_t *X4 = X3 + RB * PA;                                       // v
  float *Zs = reinterpret_cast<float *>(X4 + RB * PA);             // [16][PZ]
  bf16_t *Bs = reinterpret_cast<bf16_t *>(Zs + RB * PZ);           // biases: o [256] | qkv [768]
  float *Ls = reinterpret_cast<float *>(Bs + 4 * C);               // ln_w | ln_b
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r0 = blockIdx.x * RB, R = a.R;
  constexpr int NBLK = 4;                                          // W_o | W_q | W_k | W_v
  auto bp = [&](int b) { return b == 0 ? a.w_o + (size_t)wave * BLK : a.w_qkv + (size_t)((b - 1) * 8 + wave) * BLK; };
  WHalf w[4];
  PD_HP(0); PD_HP(1); PD_HP(2);
  tile_in(X0, a.o, r0, R, tid);
  lds_copy_bf16(Bs, a.b_o, C, tid);
  lds_copy_bf16(Bs + C, a.b_qkv, 3 * C, tid);
  lds_copy_f32(Ls, a.ln_w, C, tid);
  lds_copy_f32(Ls + C, a.ln_b, C, tid);
  float4 res[2], pos[2];                                           // the LayerNorm phase's rows of this wavefront: 2 wave, 2 wave + 1
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = min(r0 + 2 * wave + i, R - 1);
    res[i] = ld4(a.res + (size_t)m * C + 4 * lane);
    pos[i] = ld4(a.qpos + (size_t)(m / a.pos_div) * C + 4 * lane);
  }
  __syncthreads();
  const int xo = (lane & 15) * PA + 8 * (lane >> 4);
  f32x4 acc[2];
  zero_acc(acc);
  PD_BLK(0, X0 + xo);
  store_tile_f32r<true>(acc, Bs, 32 * wave, Zs, 32 * wave, lane);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 2 * wave + i, m = r0 + row;
    const float4 zv = add4(ld4(Zs + row * PZ + 4 * lane), res[i]);
    float mu, rs;
    const float4 y = ln_row(zv, ld4(Ls + 4 * lane), ld4(Ls + C + 4 * lane), a.eps, mu, rs);
    const float4 yp = add4(y, pos[i]);
    st_bf4(X1 + row * PA + 4 * lane, y);
    st_bf4(X2 + row * PA + 4 * lane, yp);
    if (m < R) {
      st4(a.z + (size_t)m * C + 4 * lane, zv);
      st4(a.y + (size_t)m * C + 4 * lane, y);
      if (lane == 0) { a.stats[m] = mu; a.stats[R + m] = rs; }
    }
  }
  __syncthreads();
  tile_out(a.y_c, X1, r0, R, tid);
  tile_out(a.ypos_c, X2, r0, R, tid);
  zero_acc(acc);                                                   // q, k from ypos_c; v from y_c
  PD_BLK(1, X2 + xo);
  store_tile_bf16<true, false>(acc, Bs + C, 32 * wave, X0, PA, 32 * wave, lane);
  zero_acc(acc);
  PD_BLK(2, X2 + xo);
  store_tile_bf16<true, false>(acc, Bs + 2 * C, 32 * wave, X3, PA, 32 * wave, lane);
  zero_acc(acc);
  PD_BLK(3, X1 + xo);
  store_tile_bf16<true, false>(acc, Bs + 3 * C, 32 * wave, X4, PA, 32 * wave, lane);
  __syncthreads();
  tile_out(a.q, X0, r0, R, tid);
  tile_out(a.k, X3, r0, R, tid);
  tile_out(a.v, X4, r0, R, tid);
}
constexpr size_t kSmemFwdA = (size_t)5 * RB * PA * 2 + (size_t)RB * PZ * 4 + 4 * C * 2 + 2 * C * 4;

// =================================================================================================== forward B
struct FwdB {
  const bf16_t *o; const float *res, *qpos; int pos_div;
  const bf16_t *w_o, *b_o; const float *ln2_w, *ln2_b;
  const bf16_t *w_1, *b_1, *w_2, *b_2; const float *ln3_w, *ln3_b, *dn_w, *dn_b;
  const bf16_t *m_w[3], *m_b[3], *wq_next, *bq_next; float eps;
  float *z2, *stats2; bf16_t *y2_c, *h; float *z3, *stats3, *y3; bf16_t *ypos_c; float *dec_out, *hstats;
  bf16_t *ef, *qc_next; int R; float *slabs, *rows; int nsplit;
};

// (Measured and dropped: P workgroups per row block that each take 1 / P of the hidden columns and meet in fp32 slabs, the last to arrive
//  finishing the layer — 25.5 us with one workgroup per row block against 33 / 38 / 56 us at P = 2 / 4 / 8: the seam costs more than the
//  shorter weight stream saves.)
// PART 0: the whole chain, one workgroup per row block.  PART 1 (grid x P): output projection + LN + this workgroup's hidden columns through
// linear1 / linear2 -> slab (and, from p = 0, the fp32 rows of y2).  PART 2 (LAYER = false here): y3 from the slabs, then the head.
template <int P, int PART, bool LAYER, bool MLP>
__global__ __launch_bounds__(NTH) void dec_fwd_b(const FwdB a)
{
  static_assert(PART != 2 || !LAYER, "PART 2 starts behind the FFN");
  constexpr int NB = 8 / P, HW = FF / P, PHS = HW + 8;            // blocks per wavefront and phase; hidden columns / LDS pitch of this workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t *X0 = reinterpret_cast<bf16_t *>(smem);
  bf16_t *X1 = X0 + RB * PA;
  bf16_t *X2 = X1 + RB * PA;
  bf16_t *Hs = X2 + RB * PA;                                       // [16][PHS] (at least one [16][PA] tile)
  float *Zs = reinterpret_cast<float *>(Hs + RB * (PHS > PA ? PHS : PA));   // [16][PZ]
  bf16_t *Bs = reinterpret_cast<bf16_t *>(Zs + RB * PZ);           // b_o 256 | b_1 2048 | b_2 256 | m_b 3 x 256 | bq 256
  float *Ls = reinterpret_cast<float *>(Bs + (6 * C + FF));        // ln2 w,b | ln3 w,b | dn w,b
  float *Ys = Ls + 6 * C;                                          // [16][PZ] residual stream rows (res, then y2): owner lanes only
  float *Ps = Ys + RB * PZ;                                        // [16][PZ] positional rows
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, R = a.R;
  const int rb = blockIdx.x / P, p = blockIdx.x - rb * P, r0 = rb * RB;
  constexpr int B1 = C, B2 = C + FF, BM = 2 * C + FF, BQ = 5 * C + FF;
  constexpr int I0 = LAYER ? 2 * NB + 1 : 0;                       // index of the first MLP block in this wavefront's block sequence
  constexpr int NBLK = I0 + (MLP && PART != 1 ? 4 : 0);            // W_o | W_1 x NB | W_2 x NB | M_0 M_1 M_2 | W_q of the next layer
  auto bp = [&](int b) -> const bf16_t * {
    if (LAYER && b == 0) return a.w_o + (size_t)wave * BLK;
    if (LAYER && b <= NB) return a.w_1 + (size_t)(p * (64 / P) + wave * NB + b - 1) * BLK;
    if (LAYER && b <= 2 * NB) return a.w_2 + (size_t)(8 * wave + p * NB + b - NB - 1) * BLK;
    const int j = b - I0;
    return (j == 0 ? a.m_w[0] : j == 1 ? a.m_w[1] : j == 2 ? a.m_w[2] : a.wq_next) + (size_t)wave * BLK;
  };
  WHalf w[4];
  PD_HP(0); PD_HP(1); PD_HP(2);
  if (LAYER) {
    tile_in(X0, a.o, r0, R, tid);
    lds_copy_bf16(Bs, a.b_o, C, tid);
    lds_copy_bf16(Bs + B1, a.b_1, FF, tid);
    lds_copy_bf16(Bs + B2, a.b_2, C, tid);
    lds_copy_f32(Ls, a.ln2_w, C, tid);
    lds_copy_f32(Ls + C, a.ln2_b, C, tid);
    lds_copy_f32(Ls + 2 * C, a.ln3_w, C, tid);
    lds_copy_f32(Ls + 3 * C, a.ln3_b, C, tid);
  }
  if (PART == 2) {
    lds_copy_bf16(Bs + B2, a.b_2, C, tid);
    lds_copy_f32(Ls + 2 * C, a.ln3_w, C, tid);
    lds_copy_f32(Ls + 3 * C, a.ln3_b, C, tid);
  }
  if (MLP && PART != 1) {
#pragma unroll
    for (int j = 0; j < 3; ++j) lds_copy_bf16(Bs + BM + j * C, a.m_b[j], C, tid);
    lds_copy_bf16(Bs + BQ, a.bq_next, C, tid);
  }
  lds_copy_f32(Ls + 4 * C, a.dn_w, C, tid);
  lds_copy_f32(Ls + 5 * C, a.dn_b, C, tid);
#pragma unroll
  for (int i = 0; i < 2; ++i) {                                    // parked in LDS: not in registers under the products
    const int row = 2 * wave + i, m = min(r0 + row, R - 1);
    // PART 2: the residual rows are PART 1's y2 (workspace); otherwise the kernel's input rows
    st4(Ys + row * PZ + 4 * lane, ld4((PART == 2 ? a.rows : a.res) + (size_t)m * C + 4 * lane));
    st4(Ps + row * PZ + 4 * lane, ld4(a.qpos + (size_t)(m / a.pos_div) * C + 4 * lane));
  }
  __syncthreads();
  const int xo = (lane & 15) * PA + 8 * (lane >> 4);
  f32x4 acc[2];
  if (LAYER) {
    // ---- attention output projection + residual + LayerNorm (every workgroup of the row block)
    zero_acc(acc);
    PD_BLK(0, X0 + xo);
    store_tile_f32r<true>(acc, Bs, 32 * wave, Zs, 32 * wave, lane);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 2 * wave + i, m = r0 + row;
      const float4 zv = add4(ld4(Zs + row * PZ + 4 * lane), ld4(Ys + row * PZ + 4 * lane));
      float mu, rs;
      const float4 y = ln_row(zv, ld4(Ls + 4 * lane), ld4(Ls + C + 4 * lane), a.eps, mu, rs);
      st_bf4(X1 + row * PA + 4 * lane, y);
      st4(Ys + row * PZ + 4 * lane, y);
      if (p == 0 && m < R) {
        st4(a.z2 + (size_t)m * C + 4 * lane, zv);
        if (PART == 1) st4(a.rows + (size_t)m * C + 4 * lane, y);
        if (lane == 0) { a.stats2[m] = mu; a.stats2[R + m] = rs; }
      }
    }
    __syncthreads();
    if (p == 0) tile_out(a.y2_c, X1, r0, R, tid);
    // ---- linear1 + ReLU: NB blocks of 32 hidden columns per wavefront
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      zero_acc(acc);
      PD_BLK(1 + c, X1 + xo);
      const int lc = (wave * NB + c) * 32;                         // local hidden column
      store_tile_bf16<true, true>(acc, Bs + B1, p * HW + lc, Hs, PHS, lc, lane);
    }
    __syncthreads();
    // the hidden rows leave for the backward pass (16-byte pieces)
#pragma unroll
    for (int i = 0; i < HW * RB / 8 / NTH; ++i) {
      const int idx = i * NTH + tid, row = idx / (HW / 8), col = (idx % (HW / 8)) * 8;
      if (r0 + row < R) *reinterpret_cast<uint4 *>(a.h + (size_t)(r0 + row) * FF + p * HW + col) = *reinterpret_cast<const uint4 *>(Hs + row * PHS + col);
    }
    // ---- linear2 over this workgroup's hidden columns
    const int ho = (lane & 15) * PHS + 8 * (lane >> 4);
    zero_acc(acc);
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      PD_BLK(NB + 1 + c, Hs + ho + 256 * c);
    }
    if (PART == 1) {
      slab_store(a.slabs + (size_t)(rb * P + p) * (RB * C), acc, 32 * wave, lane);
      return;
    }
    store_tile_f32r<true>(acc, Bs + B2, 32 * wave, Zs, 32 * wave, lane);
    __syncthreads();
  }
  // ---- FFN LayerNorm (layer) -> y3; decoder_norm of y3 -> dec_out; positions added for the next cross-attention's queries
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 2 * wave + i, m = r0 + row;
    float4 y3 = ld4(Ys + row * PZ + 4 * lane);
    float mu, rs;
    if (LAYER || PART == 2) {
      float4 x3;
      if (PART == 2) {                                             // the slabs' sum + bias, rounded like the bf16 Linear's output
        x3 = add4(slab_sum(a.slabs + (size_t)rb * a.nsplit * (RB * C), a.nsplit, row, lane), bf4(Bs + B2 + 4 * lane));
        x3 = make_float4(rbf(x3.x), rbf(x3.y), rbf(x3.z), rbf(x3.w));
      } else {
        x3 = ld4(Zs + row * PZ + 4 * lane);
      }
      const float4 zv = add4(x3, y3);
      y3 = ln_row(zv, ld4(Ls + 2 * C + 4 * lane), ld4(Ls + 3 * C + 4 * lane), a.eps, mu, rs);
      if (m < R) {
        st4(a.z3 + (size_t)m * C + 4 * lane, zv);
        st4(a.y3 + (size_t)m * C + 4 * lane, y3);
        if (lane == 0) { a.stats3[m] = mu; a.stats3[R + m] = rs; }
      }
    }
    st_bf4(X2 + row * PA + 4 * lane, add4(y3, ld4(Ps + row * PZ + 4 * lane)));
    const float4 d = ln_row(y3, ld4(Ls + 4 * C + 4 * lane), ld4(Ls + 5 * C + 4 * lane), a.eps, mu, rs);
    st_bf4(X0 + row * PA + 4 * lane, d);
    if (m < R) {
      st4(a.dec_out + (size_t)m * C + 4 * lane, d);
      if (lane == 0) { a.hstats[m] = mu; a.hstats[R + m] = rs; }
    }
  }
  __syncthreads();
  tile_out(a.ypos_c, X2, r0, R, tid);
  if (!MLP) return;
  // ---- mask-embedding MLP (bf16 between the layers, like three bf16 Linears) and the next layer's query projection
  zero_acc(acc);
  PD_BLK(I0, X0 + xo);
  store_tile_bf16<true, true>(acc, Bs + BM, 32 * wave, X1, PA, 32 * wave, lane);
  __syncthreads();
  zero_acc(acc);
  PD_BLK(I0 + 1, X1 + xo);
  store_tile_bf16<true, true>(acc, Bs + BM + C, 32 * wave, X0, PA, 32 * wave, lane);
  __syncthreads();
  zero_acc(acc);
  PD_BLK(I0 + 2, X0 + xo);
  store_tile_bf16<true, false>(acc, Bs + BM + 2 * C, 32 * wave, X1, PA, 32 * wave, lane);
  zero_acc(acc);
  PD_BLK(I0 + 3, X2 + xo);
  store_tile_bf16<true, false>(acc, Bs + BQ, 32 * wave, Hs, PA, 32 * wave, lane);       // (the hidden tile is free: its first rows take qc)
  __syncthreads();
  tile_out(a.qc_next, Hs, r0, R, tid);
  {                                                                // e -> ef[b][q][:]  (row m = q B + b)
    const int row = tid >> 5, col = (tid & 31) * 8, m = r0 + row;
    if (m < R) {
      const int B = a.pos_div, q = m / B, b = m - q * B, Q = R / B;
      *reinterpret_cast<uint4 *>(a.ef + ((size_t)b * Q + q) * C + col) = *reinterpret_cast<const uint4 *>(X1 + row * PA + col);
    }
  }
}
template <int P>
constexpr size_t smem_fwd_b()
{
  return (size_t)3 * RB * PA * 2 + (size_t)RB * ((FF / P + 8) > PA ? (FF / P + 8) : PA) * 2 + (size_t)3 * RB * PZ * 4 + (size_t)(6 * C + FF) * 2 + 6 * C * 4;
}

// =================================================================================================== backward B
struct BwdB {
  const bf16_t *dqc_next, *wqT_next; const float *d_out, *d_res, *y3, *hstats, *dn_w; float *dgb_dn;
  const float *z3, *stats3, *ln3_w; float *dgb3, *db3, *pos_acc; int pos_div;
  const bf16_t *w2T, *h, *w1T; const float *z2, *stats2, *ln2_w; float *dgb2, *db2; const bf16_t *woT;
  bf16_t *dz3_c, *dh; float *dz2; bf16_t *dz2_c, *d_o; int R; float *slabs, *rows; int nsplit;
};

// PART 0: the whole chain.  PART 1 (grid x P): (d_pos) + both LayerNorm backwards + this workgroup's hidden columns of dh and its share of
// dx -> slab (from p = 0: dz3_c, the column sums, the fp32 rows of dz3).  PART 2: dx from the slabs, LayerNorm backward, d_o.
template <int P, int PART, bool NXT>
__global__ __launch_bounds__(NTH) void dec_bwd_b(const BwdB a)
{
  constexpr int NB = 8 / P, HW = FF / P, PHS = HW + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t *X0 = reinterpret_cast<bf16_t *>(smem);
  bf16_t *X1 = X0 + RB * PA;
  bf16_t *Hs = X1 + RB * PA;                                       // h columns of this workgroup (mask), overwritten in place by dh
  float *Zs = reinterpret_cast<float *>(Hs + RB * PHS);            // [16][PZ]
  float *red = Zs + RB * PZ;                                       // [8 waves][5][256]
  float *Ls = red + NW * 5 * C;                                    // dn_w | ln3_w | ln2_w
  float *Gs = Ls + 3 * C;                                          // [16][PZ] dz3 rows (owner lanes only)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, R = a.R;
  const int rb = blockIdx.x / P, p = blockIdx.x - rb * P, r0 = rb * RB;
  constexpr int I0 = NXT ? 1 : 0;                                  // (W_q^T of the next layer) | W_2^T x NB | W_1^T x NB | W_o^T
  constexpr int IW = PART == 2 ? 0 : I0 + 2 * NB;                  // index of the W_o^T block (PART 2: its only block)
  constexpr int NBLK = PART == 1 ? I0 + 2 * NB : IW + 1;
  auto bp = [&](int b) -> const bf16_t * {
    if (PART == 2) return a.woT + (size_t)wave * BLK;
    if (NXT && b == 0) return a.wqT_next + (size_t)wave * BLK;
    const int j = b - I0;
    return j < NB ? a.w2T + (size_t)(p * (64 / P) + wave * NB + j) * BLK
                  : j < 2 * NB ? a.w1T + (size_t)(8 * wave + p * NB + j - NB) * BLK : a.woT + (size_t)wave * BLK;
  };
  WHalf w[4];
  PD_HP(0); PD_HP(1); PD_HP(2);
  const int xo = (lane & 15) * PA + 8 * (lane >> 4);
  f32x4 acc[2];
  lds_copy_f32(Ls + 2 * C, a.ln2_w, C, tid);
  if (PART != 2) {
  if (NXT) tile_in(X0, a.dqc_next, r0, R, tid);
#pragma unroll
  for (int i = 0; i < HW * RB / 8 / NTH; ++i) {
    const int idx = i * NTH + tid, row = idx / (HW / 8), col = (idx % (HW / 8)) * 8;
    *reinterpret_cast<uint4 *>(Hs + row * PHS + col) = *reinterpret_cast<const uint4 *>(a.h + (size_t)min(r0 + row, R - 1) * FF + p * HW + col);
  }
  lds_copy_f32(Ls, a.dn_w, C, tid);
  lds_copy_f32(Ls + C, a.ln3_w, C, tid);
  float4 dy[2], y3v[2], z3v[2];                                    // this wavefront's rows of the LayerNorm phases
  float st[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int mr = r0 + 2 * wave + i, m = min(mr, R - 1);
    const size_t o = (size_t)m * C + 4 * lane;
    dy[i] = mr < R ? ld4(a.d_out + o) : zero4();                  // (d_res is added after the head's LayerNorm backward, below)
    y3v[i] = ld4(a.y3 + o); z3v[i] = ld4(a.z3 + o);
    st[i][0] = a.hstats[m]; st[i][1] = a.hstats[R + m];
    st[i][2] = a.stats3[m]; st[i][3] = a.stats3[R + m];
  }
  __syncthreads();
  if (NXT) {                                                       // d(y3 + pos) from the next layer's cross-attention queries
    zero_acc(acc);
    PD_BLK(0, X0 + xo);
    store_tile_f32r<false>(acc, nullptr, 0, Zs, 32 * wave, lane);
    __syncthreads();
  }
  {
    float4 ag_dn = zero4(), ab_dn = zero4(), ag = zero4(), ab = zero4(), ad = zero4();
    const float4 gdn = ld4(Ls + 4 * lane), g3w = ld4(Ls + C + 4 * lane);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 2 * wave + i, mr = r0 + row;
      const bool valid = mr < R;
      float4 t = ln_row_bwd(dy[i], y3v[i], st[i][0], st[i][1], gdn, ag_dn, ab_dn);          // decoder_norm backward
      if (a.d_res && valid) t = add4(t, ld4(a.d_res + (size_t)mr * C + 4 * lane));
      if (NXT && valid) {
        const float4 ps = ld4(Zs + row * PZ + 4 * lane);
        t = add4(t, ps);
        if (p == 0) {
          float *pa = a.pos_acc + (size_t)(mr / a.pos_div) * C + 4 * lane;
          atomicAdd(pa, ps.x); atomicAdd(pa + 1, ps.y); atomicAdd(pa + 2, ps.z); atomicAdd(pa + 3, ps.w);
        }
      }
      if (!valid) t = zero4();
      const float4 o = ln_row_bwd(t, z3v[i], st[i][2], st[i][3], g3w, ag, ab);
      ad = add4(ad, o);
      st4(Gs + row * PZ + 4 * lane, o);
      st_bf4(X0 + row * PA + 4 * lane, o);
      if (PART == 1 && p == 0 && valid) st4(a.rows + (size_t)mr * C + 4 * lane, o);        // dz3 (fp32) for PART 2
    }
    float *rw = red + wave * 5 * C + 4 * lane;
    st4(rw, ag_dn); st4(rw + C, ab_dn); st4(rw + 2 * C, ag); st4(rw + 3 * C, ab); st4(rw + 4 * C, ad);
  }
  __syncthreads();
  if (p == 0) {
    tile_out(a.dz3_c, X0, r0, R, tid);
    float *const outs[5] = {a.dgb_dn, a.dgb_dn + C, a.dgb3, a.dgb3 + C, a.db3};
    flush_colsums<5>(red, outs, tid);
  }
  // ---- dh = (dz3_c W_2) (h > 0): NB blocks of 32 hidden columns per wavefront
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    zero_acc(acc);
    PD_BLK(I0 + c, X0 + xo);
    const int m = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bf16_t *hp = Hs + m * PHS + (wave * NB + c) * 32 + 16 * t + 4 * g;
      const uint2 hv = *reinterpret_cast<const uint2 *>(hp);
      float v[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
      if (!(bf_lo(hv.x) > 0.f)) v[0] = 0.f;
      if (!(bf_hi(hv.x) > 0.f)) v[1] = 0.f;
      if (!(bf_lo(hv.y) > 0.f)) v[2] = 0.f;
      if (!(bf_hi(hv.y) > 0.f)) v[3] = 0.f;
      *reinterpret_cast<uint2 *>(hp) = pack4(v[0], v[1], v[2], v[3]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < HW * RB / 8 / NTH; ++i) {
    const int idx = i * NTH + tid, row = idx / (HW / 8), col = (idx % (HW / 8)) * 8;
    if (r0 + row < R) *reinterpret_cast<uint4 *>(a.dh + (size_t)(r0 + row) * FF + p * HW + col) = *reinterpret_cast<const uint4 *>(Hs + row * PHS + col);
  }
  // ---- dx = dh W_1 over this workgroup's hidden columns
  const int ho = (lane & 15) * PHS + 8 * (lane >> 4);
  zero_acc(acc);
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    PD_BLK(I0 + NB + c, Hs + ho + 256 * c);
  }
  if (PART == 1) {
    slab_store(a.slabs + (size_t)(rb * P + p) * (RB * C), acc, 32 * wave, lane);
    return;
  }
  store_tile_f32r<false>(acc, nullptr, 0, Zs, 32 * wave, lane);
  }                                                                // (PART != 2)
  float4 z2v[2];
  float st2[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 2 * wave + i, m = min(r0 + row, R - 1);
    z2v[i] = ld4(a.z2 + (size_t)m * C + 4 * lane);
    st2[i][0] = a.stats2[m]; st2[i][1] = a.stats2[R + m];
    if (PART == 2) {                                               // dz3 from PART 1's rows, dx = the slabs' sum rounded like the bf16 product's output
      st4(Gs + row * PZ + 4 * lane, ld4(a.rows + (size_t)m * C + 4 * lane));
      const float4 dx = slab_sum(a.slabs + (size_t)rb * a.nsplit * (RB * C), a.nsplit, row, lane);
      st4(Zs + row * PZ + 4 * lane, make_float4(rbf(dx.x), rbf(dx.y), rbf(dx.z), rbf(dx.w)));
    }
  }
  __syncthreads();
  {
    float4 ag = zero4(), ab = zero4(), ad = zero4();
    const float4 g2w = ld4(Ls + 2 * C + 4 * lane);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 2 * wave + i, mr = r0 + row;
      float4 t = add4(ld4(Gs + row * PZ + 4 * lane), ld4(Zs + row * PZ + 4 * lane));
      if (mr >= R) t = zero4();
      const float4 o = ln_row_bwd(t, z2v[i], st2[i][0], st2[i][1], g2w, ag, ab);
      ad = add4(ad, o);
      st_bf4(X0 + row * PA + 4 * lane, o);
      if (mr < R) st4(a.dz2 + (size_t)mr * C + 4 * lane, o);
    }
    float *rw = red + wave * 5 * C + 4 * lane;
    st4(rw, ag); st4(rw + C, ab); st4(rw + 2 * C, ad);
  }
  __syncthreads();
  tile_out(a.dz2_c, X0, r0, R, tid);
  for (int i = tid; i < 3 * C; i += NTH) {                         // (stride of a wavefront's partials is still 5 arrays)
    const int k = i / C, c = i - k * C;
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) s += red[(wv * 5 + k) * C + c];
    atomicAdd((k == 0 ? a.dgb2 : k == 1 ? a.dgb2 + C : a.db2) + c, s);
  }
  // ---- d(attention output) = dz2_c W_o
  zero_acc(acc);
  PD_BLK(IW, X0 + xo);
  store_tile_bf16<false, false>(acc, nullptr, 0, X1, PA, 32 * wave, lane);
  __syncthreads();
  tile_out(a.d_o, X1, r0, R, tid);
}
template <int P>
constexpr size_t smem_bwd_b()
{
  return (size_t)2 * RB * PA * 2 + (size_t)RB * (FF / P + 8) * 2 + (size_t)2 * RB * PZ * 4 + (size_t)NW * 5 * C * 4 + 3 * C * 4;
}

// =================================================================================================== backward A
struct BwdA {
  const bf16_t *dq, *dk, *dv, *wqkvT; const float *dz_in, *z, *stats, *ln_w; float *dgb, *db, *pos_acc; int pos_div;
  const bf16_t *woT; float *dz1; bf16_t *dz1_c, *d_o; int R;
};

__global__ __launch_bounds__(NTH) void dec_bwd_a(const BwdA a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t *X0 = reinterpret_cast<bf16_t *>(smem);
  bf16_t *X1 = X0 + RB * PA;
  bf16_t *X2 = X1 + RB * PA;
  float *Zs = reinterpret_cast<float *>(X2 + RB * PA);             // d_tc
  float *Z2 = Zs + RB * PZ;                                        // d_tp
  float *red = Z2 + RB * PZ;                                       // [8][3][256]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r0 = blockIdx.x * RB, R = a.R;
  constexpr int NBLK = 4;                                          // [W_q; W_k; W_v]^T ([C, 3C]: three contraction blocks per 32 rows) | W_o^T
  auto bp = [&](int b) { return b < 3 ? a.wqkvT + (size_t)(wave * 3 + b) * BLK : a.woT + (size_t)wave * BLK; };
  WHalf w[4];
  PD_HP(0); PD_HP(1); PD_HP(2);
  tile_in(X0, a.dq, r0, R, tid);
  tile_in(X1, a.dk, r0, R, tid);
  tile_in(X2, a.dv, r0, R, tid);
  float4 dzi[2], zv[2];
  float st[2][2];
  const float4 gm = ld4(a.ln_w + 4 * lane);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int mr = r0 + 2 * wave + i, m = min(mr, R - 1);
    const size_t o = (size_t)m * C + 4 * lane;
    dzi[i] = mr < R ? ld4(a.dz_in + o) : zero4();
    zv[i] = ld4(a.z + o);
    st[i][0] = a.stats[m]; st[i][1] = a.stats[R + m];
  }
  __syncthreads();
  const int xo = (lane & 15) * PA + 8 * (lane >> 4);
  f32x4 acc[2];
  zero_acc(acc);
  PD_BLK(0, X0 + xo);
  PD_BLK(1, X1 + xo);
  store_tile_f32r<false>(acc, nullptr, 0, Z2, 32 * wave, lane);
  zero_acc(acc);
  PD_BLK(2, X2 + xo);
  store_tile_f32r<false>(acc, nullptr, 0, Zs, 32 * wave, lane);
  __syncthreads();
  {
    float4 ag = zero4(), ab = zero4(), ad = zero4();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 2 * wave + i, mr = r0 + row;
      const float4 ps = ld4(Z2 + row * PZ + 4 * lane);
      float4 t = add4(add4(dzi[i], ld4(Zs + row * PZ + 4 * lane)), ps);
      if (mr < R) {
        float *pa = a.pos_acc + (size_t)(mr / a.pos_div) * C + 4 * lane;
        atomicAdd(pa, ps.x); atomicAdd(pa + 1, ps.y); atomicAdd(pa + 2, ps.z); atomicAdd(pa + 3, ps.w);
      } else {
        t = zero4();
      }
      const float4 o = ln_row_bwd(t, zv[i], st[i][0], st[i][1], gm, ag, ab);
      ad = add4(ad, o);
      st_bf4(X0 + row * PA + 4 * lane, o);
      if (mr < R) st4(a.dz1 + (size_t)mr * C + 4 * lane, o);
    }
    float *rw = red + wave * 3 * C + 4 * lane;
    st4(rw, ag); st4(rw + C, ab); st4(rw + 2 * C, ad);
  }
  __syncthreads();
  tile_out(a.dz1_c, X0, r0, R, tid);
  {
    float *const outs[3] = {a.dgb, a.dgb + C, a.db};
    flush_colsums<3>(red, outs, tid);
  }
  zero_acc(acc);
  PD_BLK(3, X0 + xo);
  store_tile_bf16<false, false>(acc, nullptr, 0, X1, PA, 32 * wave, lane);
  __syncthreads();
  tile_out(a.d_o, X1, r0, R, tid);
}
constexpr size_t kSmemBwdA = (size_t)3 * RB * PA * 2 + (size_t)2 * RB * PZ * 4 + (size_t)NW * 3 * C * 4;

// =================================================================================================== weight packing
// W_eff[n][k] (= src[n][k], or src[k][n] when transposed) -> blocks of 32 n x 256 k, block (nb, kc) at ((nb (K / 256) + kc) BLK), inside a
// block 16-byte piece (i = 8 t + s, lane) = W_eff[32 nb + 16 t + (lane & 15)][256 kc + 32 s + 8 (lane >> 4) .. + 7]: what wloadh reads
struct PackProblem { const bf16_t *src; bf16_t *dst; int rows, cols, transpose, first_block; };

__global__ __launch_bounds__(256) void dec_pack_grouped(const PackProblem *__restrict__ tab, int count)
{
  __shared__ __attribute__((aligned(16))) bf16_t tile[256][40];    // transposed problems: [k][n] as read (32 n + pad)
  int lo = 0, hi = count - 1;
  const int bid = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].first_block <= bid) lo = mid; else hi = mid - 1;
  }
  PackProblem pr = tab[lo];
  pr.src = pd_as_global(pr.src); pr.dst = pd_as_global(pr.dst);                      // (pd_common.h: table pointers would be FLAT)
  const int local = bid - pr.first_block;
  const int K = pr.transpose ? pr.rows : pr.cols, kcn = K / 256, nb = local / kcn, kc = local - nb * kcn;
  bf16_t *dst = pr.dst + (size_t)local * BLK;
  const int tid = threadIdx.x;
  if (!pr.transpose) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int piece = j * 256 + tid, i = piece >> 6, lane = piece & 63, t = i >> 3, s = i & 7;
      const uint4 v = *reinterpret_cast<const uint4 *>(pr.src + (size_t)(32 * nb + 16 * t + (lane & 15)) * pr.cols + 256 * kc + 32 * s + 8 * (lane >> 4));
      *reinterpret_cast<uint4 *>(dst + (size_t)piece * 8) = v;
    }
    return;
  }
  // src rows k = 256 kc .. + 255, columns n = 32 nb .. + 31: 64 bytes per row
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int piece = j * 256 + tid, k = piece >> 2, c8 = (piece & 3) * 8;
    *reinterpret_cast<uint4 *>(&tile[k][c8]) = *reinterpret_cast<const uint4 *>(pr.src + (size_t)(256 * kc + k) * pr.cols + 32 * nb + c8);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int piece = j * 256 + tid, i = piece >> 6, lane = piece & 63, t = i >> 3, s = i & 7;
    const int n = 16 * t + (lane & 15), k0 = 32 * s + 8 * (lane >> 4);
    unsigned short e[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) e[q] = tile[k0 + q][n];
    uint4 o;
    o.x = e[0] | ((unsigned)e[1] << 16); o.y = e[2] | ((unsigned)e[3] << 16); o.z = e[4] | ((unsigned)e[5] << 16); o.w = e[6] | ((unsigned)e[7] << 16);
    *reinterpret_cast<uint4 *>(dst + (size_t)piece * 8) = o;
  }
}

template <class K>
int allow_smem(K kernel, size_t bytes, const char *who)
{
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
    return pd_set_error(PD_ERR_LAUNCH, "%s: cannot reserve %zu bytes of LDS", who, bytes);
  return PD_OK;
}

int g_split = -1;                                                  // workgroups per row block of the FFN launches (PD_DEC_SPLIT = 1: one launch; 2 | 4 | 8: two launches)
int split_mode()
{
  if (g_split < 0) {
    const char *e = getenv("PD_DEC_SPLIT");
    const int v = e ? atoi(e) : 4;
    g_split = (v == 1 || v == 2 || v == 4 || v == 8) ? v : 4;
  }
  return g_split;
}

template <int P, int PART, bool LAYER, bool MLP>
int launch_fwd_b(const FwdB &a, hipStream_t s)
{
  static int ok = allow_smem(dec_fwd_b<P, PART, LAYER, MLP>, smem_fwd_b<P>(), "pd_dec_fwd_b");
  if (ok != PD_OK) return ok;
  hipLaunchKernelGGL((dec_fwd_b<P, PART, LAYER, MLP>), dim3(((a.R + RB - 1) / RB) * (PART == 1 ? P : 1)), dim3(NTH), smem_fwd_b<P>(), s, a);
  return pd_check_launch("pd_dec_fwd_b");
}
// the layer part as two launches: P workgroups per row block through the FFN, then one per row block from the slabs
template <int P>
int launch_fwd_b_split(const FwdB &a, bool mlp, hipStream_t s)
{
  const int rc = launch_fwd_b<P, 1, true, false>(a, s);
  if (rc != PD_OK) return rc;
  return mlp ? launch_fwd_b<1, 2, false, true>(a, s) : launch_fwd_b<1, 2, false, false>(a, s);
}
template <int P, int PART, bool NXT>
int launch_bwd_b(const BwdB &a, hipStream_t s)
{
  static int ok = allow_smem(dec_bwd_b<P, PART, NXT>, smem_bwd_b<P>(), "pd_dec_bwd_b");
  if (ok != PD_OK) return ok;
  hipLaunchKernelGGL((dec_bwd_b<P, PART, NXT>), dim3(((a.R + RB - 1) / RB) * (PART == 1 ? P : 1)), dim3(NTH), smem_bwd_b<P>(), s, a);
  return pd_check_launch("pd_dec_bwd_b");
}
template <int P>
int launch_bwd_b_split(const BwdB &a, hipStream_t s)
{
  const int rc = a.dqc_next ? launch_bwd_b<P, 1, true>(a, s) : launch_bwd_b<P, 1, false>(a, s);
  if (rc != PD_OK) return rc;
  return launch_bwd_b<1, 2, false>(a, s);
}

}  // namespace

extern "C" int pd_dec_split(void) { return split_mode(); }
extern "C" int64_t pd_dec_workspace_bytes(int R) { return R <= 0 ? 0 : (int64_t)((R + RB - 1) / RB) * (8 + 1) * RB * C * 4; }
extern "C" int64_t pd_dec_pack_table_bytes(int count) { return (int64_t)count * sizeof(PackProblem); }

extern "C" int pd_dec_pack_grouped(const PdDecPack *descs, int count, void *table_host_pinned, void *table_device, void *stream)
{
  if (count <= 0) return PD_OK;
  if (!descs || !table_host_pinned || !table_device) return pd_set_error(PD_ERR_INVALID_ARG, "pd_dec_pack_grouped: null pointer");
  PackProblem *h = reinterpret_cast<PackProblem *>(table_host_pinned);
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const PdDecPack &d = descs[i];
    const int N = d.transpose ? d.cols : d.rows, K = d.transpose ? d.rows : d.cols;
    if (!d.src || !d.dst || N <= 0 || K <= 0 || N % 32 || K % 256)
      return pd_set_error(PD_ERR_INVALID_ARG, "pd_dec_pack_grouped: problem %d: output rows %% 32 == 0 and contraction %% 256 == 0 required (%d x %d)", i, N, K);
    h[i] = PackProblem{(const bf16_t *)d.src, (bf16_t *)d.dst, d.rows, d.cols, d.transpose, blocks};
    blocks += (N / 32) * (K / 256);
  }
  if (hipMemcpyAsync(table_device, h, (size_t)count * sizeof(PackProblem), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess)
    return pd_set_error(PD_ERR_LAUNCH, "pd_dec_pack_grouped: table upload failed");
  hipLaunchKernelGGL(dec_pack_grouped, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const PackProblem *)table_device, count);
  return pd_check_launch("pd_dec_pack_grouped");
}

extern "C" int pd_dec_fwd_a(const void *o, const float *res, const float *qpos, int pos_div, const void *w_o, const void *b_o, const float *ln_w,
                            const float *ln_b, float eps, const void *w_qkv, const void *b_qkv, float *z, float *stats, float *y, void *y_c,
                            void *ypos_c, void *q, void *k, void *v, int R, void *stream)
{
  if (!o || !res || !qpos || !w_o || !b_o || !ln_w || !ln_b || !w_qkv || !b_qkv || !z || !stats || !y || !y_c || !ypos_c || !q || !k || !v)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_dec_fwd_a: null pointer");
  if (R <= 0 || pos_div <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_dec_fwd_a: R = %d, pos_div = %d", R, pos_div);
  static int ok = allow_smem(dec_fwd_a, kSmemFwdA, "pd_dec_fwd_a");
  if (ok != PD_OK) return ok;
  FwdA a{(const bf16_t *)o, res, qpos, pos_div, (const bf16_t *)w_o, (const bf16_t *)b_o, ln_w, ln_b, eps, (const bf16_t *)w_qkv, (const bf16_t *)b_qkv,
         z, stats, y, (bf16_t *)y_c, (bf16_t *)ypos_c, (bf16_t *)q, (bf16_t *)k, (bf16_t *)v, R};
  hipLaunchKernelGGL(dec_fwd_a, dim3((R + RB - 1) / RB), dim3(NTH), kSmemFwdA, (hipStream_t)stream, a);
  return pd_check_launch("pd_dec_fwd_a");
}

extern "C" int pd_dec_fwd_b(const void *o, const float *res, const float *qpos, int pos_div, const void *w_o, const void *b_o, const float *ln2_w,
                            const float *ln2_b, const void *w_1, const void *b_1, const void *w_2, const void *b_2, const float *ln3_w,
                            const float *ln3_b, const float *dn_w, const float *dn_b, const void *m0_w, const void *m0_b, const void *m1_w,
                            const void *m1_b, const void *m2_w, const void *m2_b, const void *wq_next, const void *bq_next, float eps, float *z2,
                            float *stats2, void *y2_c, void *h, float *z3, float *stats3, float *y3, void *ypos_c, float *dec_out, float *hstats,
                            void *ef, void *qc_next, void *workspace, int R, int flags, void *stream)
{
  const bool layer = flags & 1, mlp = flags & 2;
  if (!res || !qpos || !dn_w || !dn_b || !ypos_c || !dec_out || !hstats) return pd_set_error(PD_ERR_INVALID_ARG, "pd_dec_fwd_b: null pointer");
  if (layer && (!o || !w_o || !b_o || !ln2_w || !ln2_b || !w_1 || !b_1 || !w_2 || !b_2 || !ln3_w || !ln3_b || !z2 || !stats2 || !y2_c || !h || !z3 || !stats3 ||
                !y3))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_dec_fwd_b: null pointer (layer part)");
  if (mlp && (!m0_w || !m0_b || !m1_w || !m1_b || !m2_w || !m2_b || !wq_next || !bq_next || !ef || !qc_next))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_dec_fwd_b: null pointer (head part)");
  if (R <= 0 || pos_div <= 0 || R % pos_div) return pd_set_error(PD_ERR_INVALID_ARG, "pd_dec_fwd_b: R = %d, pos_div = %d", R, pos_div);
  FwdB a{(const bf16_t *)o, res, qpos, pos_div, (const bf16_t *)w_o, (const bf16_t *)b_o, ln2_w, ln2_b, (const bf16_t *)w_1, (const bf16_t *)b_1,
         (const bf16_t *)w_2, (const bf16_t *)b_2, ln3_w, ln3_b, dn_w, dn_b,
         {(const bf16_t *)m0_w, (const bf16_t *)m1_w, (const bf16_t *)m2_w}, {(const bf16_t *)m0_b, (const bf16_t *)m1_b, (const bf16_t *)m2_b},
         (const bf16_t *)wq_next, (const bf16_t *)bq_next, eps, z2, stats2, (bf16_t *)y2_c, (bf16_t *)h, z3, stats3, y3, (bf16_t *)ypos_c, dec_out, hstats,
         (bf16_t *)ef, (bf16_t *)qc_next, R, reinterpret_cast<float *>(workspace),
         reinterpret_cast<float *>(workspace) + (size_t)((R + RB - 1) / RB) * 8 * RB * C, (layer && workspace) ? split_mode() : 1};
  hipStream_t s = (hipStream_t)stream;
  if (!layer) return mlp ? launch_fwd_b<1, 0, false, true>(a, s) : launch_fwd_b<1, 0, false, false>(a, s);
  switch (a.nsplit) {
    case 2: return launch_fwd_b_split<2>(a, mlp, s);
    case 4: return launch_fwd_b_split<4>(a, mlp, s);
    case 8: return launch_fwd_b_split<8>(a, mlp, s);
    default: return mlp ? launch_fwd_b<1, 0, true, true>(a, s) : launch_fwd_b<1, 0, true, false>(a, s);
  }
}

extern "C" int pd_dec_bwd_b(const void *dqc_next, const void *wqT_next, const float *d_out, const float *d_res, const float *y3, const float *hstats,
                            const float *dn_w, float *dgb_dn, const float *z3, const float *stats3, const float *ln3_w, float *dgb3, float *db3,
                            float *pos_acc, int pos_div, const void *w2T, const void *h, const void *w1T, const float *z2, const float *stats2,
                            const float *ln2_w, float *dgb2, float *db2, const void *woT, void *dz3_c, void *dh, float *dz2, void *dz2_c, void *d_o,
                            void *workspace, int R, void *stream)
{
  if (!d_out || !y3 || !hstats || !dn_w || !dgb_dn || !z3 || !stats3 || !ln3_w || !dgb3 || !db3 || !w2T || !h || !w1T || !z2 || !stats2 || !ln2_w ||
      !dgb2 || !db2 || !woT || !dz3_c || !dh || !dz2 || !dz2_c || !d_o)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_dec_bwd_b: null pointer");
  if ((dqc_next != nullptr) != (wqT_next != nullptr) || (dqc_next && !pos_acc)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_dec_bwd_b: next-layer operands");
  if (R <= 0 || pos_div <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_dec_bwd_b: R = %d, pos_div = %d", R, pos_div);
  BwdB a{(const bf16_t *)dqc_next, (const bf16_t *)wqT_next, d_out, d_res, y3, hstats, dn_w, dgb_dn, z3, stats3, ln3_w, dgb3, db3, pos_acc, pos_div,
         (const bf16_t *)w2T, (const bf16_t *)h, (const bf16_t *)w1T, z2, stats2, ln2_w, dgb2, db2, (const bf16_t *)woT, (bf16_t *)dz3_c, (bf16_t *)dh, dz2,
         (bf16_t *)dz2_c, (bf16_t *)d_o, R, reinterpret_cast<float *>(workspace),
         reinterpret_cast<float *>(workspace) + (size_t)((R + RB - 1) / RB) * 8 * RB * C, workspace ? split_mode() : 1};
  hipStream_t s = (hipStream_t)stream;
  switch (a.nsplit) {
    case 2: return launch_bwd_b_split<2>(a, s);
    case 4: return launch_bwd_b_split<4>(a, s);
    case 8: return launch_bwd_b_split<8>(a, s);
    default: return a.dqc_next ? launch_bwd_b<1, 0, true>(a, s) : launch_bwd_b<1, 0, false>(a, s);
  }
}

extern "C" int pd_dec_bwd_a(const void *dq, const void *dk, const void *dv, const void *wqkvT, const float *dz_in, const float *z, const float *stats,
                            const float *ln_w, float *dgb, float *db, float *pos_acc, int pos_div, const void *woT, float *dz1, void *dz1_c, void *d_o,
                            int R, void *stream)
{
  if (!dq || !dk || !dv || !wqkvT || !dz_in || !z || !stats || !ln_w || !dgb || !db || !pos_acc || !woT || !dz1 || !dz1_c || !d_o)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_dec_bwd_a: null pointer");
  if (R <= 0 || pos_div <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_dec_bwd_a: R = %d, pos_div = %d", R, pos_div);
  static int ok = allow_smem(dec_bwd_a, kSmemBwdA, "pd_dec_bwd_a");
  if (ok != PD_OK) return ok;
  BwdA a{(const bf16_t *)dq, (const bf16_t *)dk, (const bf16_t *)dv, (const bf16_t *)wqkvT, dz_in, z, stats, ln_w, dgb, db, pos_acc, pos_div,
         (const bf16_t *)woT, dz1, (bf16_t *)dz1_c, (bf16_t *)d_o, R};
  hipLaunchKernelGGL(dec_bwd_a, dim3((R + RB - 1) / RB), dim3(NTH), kSmemBwdA, (hipStream_t)stream, a);
  return pd_check_launch("pd_dec_bwd_a");
}
