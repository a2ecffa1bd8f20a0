// Multi-scale deformable attention for gfx950 (MI355X) — forward + backward.
//
// Semantics follow the reference's intended native op
// (part_distillation/modeling/pixel_decoder/ops/src/cuda/ms_deform_im2col_cuda.cuh:38-304,
// ms_deform_attn_cuda.cu:26-159) == its PyTorch fallback
// (functions/ms_deform_attn_func.py:55-75).  The launch geometry is NOT the
// reference's (1 thread / output scalar, 32-thread backward blocks):
//
//  * fast path (fp32, head_dim 32, the only shape Mask2Former uses): one
//    (batch, query, head) triple per 8-lane group, each lane owning 4 channels
//    as a float4.  A corner read is then one 128-byte row per group, issued as
//    a single global_load_dwordx4 per lane; a 64-lane wavefront covers all 8
//    heads of one query.  (A 4-lanes x 8-channels forward with 24-bit index arithmetic and zero-row reads for masked
//    corners — the recipe that sped the backward up — measured SLOWER, 0.123 vs 0.099 ms: the forward is bound by gather
//    latency, not by instruction issue.)  Backward reduces grad_sampling_loc / grad_attn_weight
//    across the 8 lanes with DPP (no LDS, no barrier) and parks the results so
//    that the stores are 32/64-byte contiguous per group.
//  * blockIdx is remapped so that each XCD (block b is dispatched to XCD b%8)
//    walks one contiguous eighth of the query range: its private 4 MiB L2 then
//    holds just the band of `value` rows those queries sample.
//  * generic path (any head_dim, fp32/fp64): forward = one lane per output
//    scalar; backward = one wavefront per (batch, query, head), lanes striding
//    over channels, wave-level shuffle reduction.
//
// HBM-bound kernel: algorithmic bytes per launch are value + loc + attn + out
// (forward), see DESIGN.md.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pd_msda.h"
#include "pd_common.h"

int g_pd_dbg_force_generic = 0;
int g_pd_dbg_bwd_threads = 0;
int g_pd_dbg_ablate = 0;
int g_pd_dbg_atomic_scope = 0;   // experiments only (pd_debug_set): 0 = agent scope, 1 = workgroup scope
int g_msda_fwd_q4 = []() { const char *e = getenv("PD_MSDA_FWD_Q4"); return e ? atoi(e) : 1; }();   // 0: the 8-lane forward kernels (A/B)
int g_pd_dbg_bwd_variant = 0;    // experiments only: 1 = the per-destination-level tiled backward instead of the all-level owner kernel

namespace {

template <int SCOPE>
__device__ __forceinline__ void scoped_add(float *p, float v)
{
  if (SCOPE == 0) unsafeAtomicAdd(p, v);
  else if (SCOPE == 2) asm volatile("" ::"v"(p), "v"(v));   // ablation: no atomic at all
  else (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ----------------------------------------------------------------------------------------- helpers
__device__ __forceinline__ int xcd_chunked_block(int bid, int nblocks)
{
  // nblocks is a multiple of 8 (host rounds up).  XCD k = bid % 8 gets logical
  // blocks [k*per, (k+1)*per).
  const int per = nblocks >> 3;
  return (bid & 7) * per + (bid >> 3);
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float x)
{
  int y = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true);
  return x + __int_as_float(y);
}

// sum over the 8 lanes of an aligned 8-lane group; every lane gets the total
__device__ __forceinline__ float group8_sum(float x)
{
  x = dpp_add<0xB1>(x);   // quad_perm [1,0,3,2]
  x = dpp_add<0x4E>(x);   // quad_perm [2,3,0,1]
  x = dpp_add<0x141>(x);  // row_half_mirror: lane i <-> 7-i inside each 8-lane half row
  return x;
}

// reference .cuh:38-89 geometry; `stride_w` = num_heads*channels elements
template <typename T>
__device__ __forceinline__ void corner_setup(T h, T w, int H, int W, int stride_w, bool in_range,
                                             int off[4], bool ok[4], T cw[4], T &lh, T &lw, T &hh, T &hw)
{
  const T hf = floor(h), wf = floor(w);
  int h_low = in_range ? (int)hf : 0, w_low = in_range ? (int)wf : 0;
  const int h_high = h_low + 1, w_high = w_low + 1;
  lh = h - hf; lw = w - wf; hh = 1 - lh; hw = 1 - lw;
  const bool hl = h_low >= 0, wl = w_low >= 0, hh_ok = h_high <= H - 1, wh_ok = w_high <= W - 1;
  ok[0] = in_range && hl && wl;
  ok[1] = in_range && hl && wh_ok;
  ok[2] = in_range && hh_ok && wl;
  ok[3] = in_range && hh_ok && wh_ok;
  const int hlc = min(max(h_low, 0), H - 1), hhc = min(max(h_high, 0), H - 1);
  const int wlc = min(max(w_low, 0), W - 1), whc = min(max(w_high, 0), W - 1);
  const int hs = W * stride_w;
  off[0] = hlc * hs + wlc * stride_w;
  off[1] = hlc * hs + whc * stride_w;
  off[2] = hhc * hs + wlc * stride_w;
  off[3] = hhc * hs + whc * stride_w;
  if (!in_range) { lh = 0; lw = 0; hh = 0; hw = 0; }   // NaN / inf locations contribute exactly nothing
  cw[0] = hh * hw; cw[1] = hh * lw; cw[2] = lh * hw; cw[3] = lh * lw;
}

// ----------------------------------------------------------------------------------------- fast forward
// D == 32, fp32.  256 threads = 32 (b,q,m) triples per block.
// FUSED (round 5, pd_msda_fused_forward): the kernel reads the RAW output row of the merged sampling_offsets / attention_weights
// projection instead of materialised sampling locations and attention probabilities — `loc` is then that matrix ([batch * Lq, ld_oa]:
// columns [0, 2 M L P) the offsets in (head, level, point, xy) order, [2 M L P, 3 M L P) the logits in (head, level, point) order, i.e.
// ms_deform_attn.py:108-111's two views), `attn` the reference points [batch * Lq, L, 2] — and forms
//   softmax over the head's L P logits          (ms_deform_attn.py:111)
//   loc = reference point + offset / (W_l, H_l)  (ms_deform_attn.py:114-117)
// in registers (every lane of the 8-lane group redundantly: ~80 instructions next to the ~900 of the gathers).  The softmax
// statistics {max, 1 / sum} of every (query, head) go to `stats` (8 bytes: the backward re-forms a probability with ONE exponential).
// Saves the prep launch and the 144-byte-per-(query, head) round trip of loc / attn through memory.
template <int L_, int P_, bool FUSED = false>
__global__ __launch_bounds__(256, 4) void msda_fwd_d32(const float *__restrict__ value, const int64_t *__restrict__ shapes,
                                                     const int64_t *__restrict__ lvl_start, const float *__restrict__ loc,
                                                     const float *__restrict__ attn, float *__restrict__ out,
                                                     int S, int M, int Lq, int total_qm, unsigned *__restrict__ row_amax,
                                                     int ld_oa = 0, float2 *__restrict__ stats = nullptr, int band_major = 0)
{
  const int lb = xcd_chunked_block(blockIdx.x, gridDim.x);
  int qm = lb * 32 + (threadIdx.x >> 3);
  if constexpr (FUSED) {
    // Self-attention over the pyramid (queries == pixels): XCD k (a contiguous eighth of the logical blocks) takes band k of EVERY level of
    // every image instead of a contiguous eighth of the concatenated levels — the queries of one band sample the same band (+ halo) of all
    // three levels' value rows, 1 / 8 of `value` per image (2.75 MB at config 2: inside the XCD's 4 MB L2), where the first XCDs used to
    // hold whole small levels whose queries sample every row of every level.  Needs M == 8 (4 queries per block), level sizes % 32 == 0.
    const int n0 = (int)(shapes[0] * shapes[1]), n1 = (int)(shapes[2] * shapes[3]), n2 = (int)(shapes[4] * shapes[5]);
    if (band_major && M == 8 && Lq == S && n0 + n1 + n2 == S && !((n0 | n1 | n2) & 31) && (gridDim.x & 7) == 0 && total_qm == (int)gridDim.x * 32) {
      const int per = gridDim.x >> 3, k = lb / per, p = (lb - k * per) * 4 + (threadIdx.x >> 6);       // query index inside XCD k's share
      const int sq = S >> 3, bimg = p / sq, r = p - bimg * sq;
      const int b0 = n0 >> 3, b1 = n1 >> 3, b2 = n2 >> 3;
      int q;
      if (r < b0) q = k * b0 + r;
      else if (r < b0 + b1) q = n0 + k * b1 + (r - b0);
      else q = n0 + n1 + k * b2 + (r - b0 - b1);
      qm = (bimg * S + q) * 8 + ((threadIdx.x >> 3) & 7);
    }
  }
  if (qm >= total_qm) return;
  const int sub = threadIdx.x & 7;
  const int m = qm % M;
  const int b = (qm / M) / Lq;
  const int stride_w = M * 32;
  static_assert(P_ == 4, "fast path is written for 4 points per level");
  const float4 *lp4, *ap4 = nullptr;
  const float *refp = nullptr;
  float prob[FUSED ? L_ * P_ : 1];
  if constexpr (FUSED) {
    const int64_t bq = qm / M;
    const float *row = loc + bq * ld_oa;
    lp4 = reinterpret_cast<const float4 *>(row + m * (L_ * P_ * 2));
    const float4 *lg4 = reinterpret_cast<const float4 *>(row + M * (L_ * P_ * 2) + m * (L_ * P_));
    refp = attn + bq * (L_ * 2);
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < L_; ++i) {
      const float4 t = lg4[i];
      prob[4 * i] = t.x; prob[4 * i + 1] = t.y; prob[4 * i + 2] = t.z; prob[4 * i + 3] = t.w;
      mx = fmaxf(fmaxf(fmaxf(mx, t.x), fmaxf(t.y, t.z)), t.w);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < L_ * P_; ++i) { prob[i] = __expf(prob[i] - mx); sum += prob[i]; }
    const float inv = 1.f / sum;
#pragma unroll
    for (int i = 0; i < L_ * P_; ++i) prob[i] *= inv;
    if (stats && sub == 0) stats[qm] = make_float2(mx, inv);
  } else {
    lp4 = reinterpret_cast<const float4 *>(loc + (int64_t)qm * (L_ * P_ * 2));
    ap4 = reinterpret_cast<const float4 *>(attn + (int64_t)qm * (L_ * P_));
  }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
  for (int l = 0; l < L_; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const float *vbase = value + ((int64_t)b * S + lvl_start[l]) * stride_w + m * 32 + sub * 4;
    float4 l01 = lp4[2 * l], l23 = lp4[2 * l + 1], a4;
    if constexpr (FUSED) {
      const float rx = refp[2 * l], ry = refp[2 * l + 1];
      const float fw = (float)W, fh = (float)H;
      if (((W & (W - 1)) | (H & (H - 1))) == 0) {            // powers of two (the usual pyramid): x * (1 / W) IS x / W, bit for bit
        const float iw = 1.f / fw, ih = 1.f / fh;
        l01 = make_float4(rx + l01.x * iw, ry + l01.y * ih, rx + l01.z * iw, ry + l01.w * ih);
        l23 = make_float4(rx + l23.x * iw, ry + l23.y * ih, rx + l23.z * iw, ry + l23.w * ih);
      } else {
        l01 = make_float4(rx + l01.x / fw, ry + l01.y / fh, rx + l01.z / fw, ry + l01.w / fh);
        l23 = make_float4(rx + l23.x / fw, ry + l23.y / fh, rx + l23.z / fw, ry + l23.w / fh);
      }
      // (l is a runtime index: select instead of indexing the register array)
      a4 = l == 0 ? make_float4(prob[0], prob[1], prob[2], prob[3]) : l == 1 ? make_float4(prob[4 % (L_ * P_)], prob[5 % (L_ * P_)], prob[6 % (L_ * P_)], prob[7 % (L_ * P_)])
                                                                             : make_float4(prob[8 % (L_ * P_)], prob[9 % (L_ * P_)], prob[10 % (L_ * P_)], prob[11 % (L_ * P_)]);
    } else {
      a4 = ap4[l];
    }
    const float locs[8] = {l01.x, l01.y, l01.z, l01.w, l23.x, l23.y, l23.z, l23.w};
    const float aw[4] = {a4.x, a4.y, a4.z, a4.w};
    float4 v[P_][4];
    float cwk[P_][4];
#pragma unroll
    for (int p = 0; p < P_; ++p) {
      const float x = locs[2 * p], y = locs[2 * p + 1];
      const float h_im = y * H - 0.5f, w_im = x * W - 0.5f;
      const bool in_range = h_im > -1 && w_im > -1 && h_im < H && w_im < W;
      int off[4]; bool ok[4]; float cw[4], lh, lw, hh, hw;
      corner_setup<float>(h_im, w_im, H, W, stride_w, in_range, off, ok, cw, lh, lw, hh, hw);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[p][k] = *reinterpret_cast<const float4 *>(vbase + off[k]);
        cwk[p][k] = ok[k] ? cw[k] * aw[p] : 0.f;   // masked corners get weight 0 ...
        if (!ok[k]) v[p][k] = make_float4(0.f, 0.f, 0.f, 0.f);   // ... and a 0 value (inf/NaN safe)
      }
    }
#pragma unroll
    for (int p = 0; p < P_; ++p) {
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        val.x += cwk[p][k] * v[p][k].x;
        val.y += cwk[p][k] * v[p][k].y;
        val.z += cwk[p][k] * v[p][k].z;
        val.w += cwk[p][k] * v[p][k].w;
      }
      acc.x += val.x; acc.y += val.y; acc.z += val.z; acc.w += val.w;
    }
  }
  *reinterpret_cast<float4 *>(out + (int64_t)qm * 32 + sub * 4) = acc;
  if (row_amax) {
    // absolute maximum of the output ROW (all heads of the query) for the fp16 two-plane GEMM that reads it (pd_gemm.h): the 8 lanes
    // of the (query, head) group reduce, one atomic max per group into the zero-filled array (non-negative floats order like their bits)
    float mx = fmaxf(fmaxf(fabsf(acc.x), fabsf(acc.y)), fmaxf(fabsf(acc.z), fabsf(acc.w)));
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64)); mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
    if (M == 8) {
      // a wavefront's eight groups are the eight heads of ONE query (32 groups per block, total a multiple of 8): reduce across them
      // and store — eight atomics of one wave instruction on one address serialise (they cost this kernel 35 us per launch)
      mx = fmaxf(mx, __shfl_xor(mx, 8, 64)); mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if ((threadIdx.x & 63) == 0) row_amax[qm / M] = __float_as_uint(mx);
    } else if (sub == 0 && mx > 0.f) {
      atomicMax(row_amax + qm / M, __float_as_uint(mx));
    }
  }
}

// ----------------------------------------------------------------------------------------- fast forward, 4 lanes per (query, head)  (round 6)
// msda_fwd_d32 spends most of its instructions on work all 8 lanes of a (query, head) group do alike: the 12-way softmax and, per sampling
// point, ~45 instructions of geometry (floor, clamps, four offsets, four masks, four weights) — 540 of its ~890 instructions per lane.  Here a
// (query, head) is FOUR lanes that own 8 channels each (4 sub .. + 3 and 16 + 4 sub .. + 3: a wave instruction still reads 64 contiguous
// bytes per group) and lane `sub` works out point `sub` of each level ONCE — 3 points per lane instead of 12 — then hands the point's four
// offsets, four weights (mask and attention weight folded in) and mask bits round the quad as DPP quad_perm broadcasts (9 moves per point).
// The softmax is one exponential per lane and level + two quad reductions.  Per (query, head): ~2 800 issued instructions instead of ~7 100.
// A masked corner loads from a zero row (the reference never reads it: a NaN there must not leak through a 0 weight).
__device__ __attribute__((aligned(128))) float g_msda_zero_row[32];

template <int P>
__device__ __forceinline__ int quad_bcast_i(int x) { return __builtin_amdgcn_update_dpp(0, x, P * 0x55, 0xF, 0xF, true); }
template <int P>
__device__ __forceinline__ float quad_bcast_f(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), P * 0x55, 0xF, 0xF, true)); }
__device__ __forceinline__ float quad_max(float x)
{
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true)));
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true)));
  return x;
}
__device__ __forceinline__ float quad_sum(float x)
{
  x = dpp_add<0xB1>(x);
  x = dpp_add<0x4E>(x);
  return x;
}

template <int L_, int P_, bool FUSED>
__global__ __launch_bounds__(256, 4) void msda_fwd_q4(const float *__restrict__ value, const int64_t *__restrict__ shapes,
                                                    const int64_t *__restrict__ lvl_start, const float *__restrict__ loc,
                                                    const float *__restrict__ attn, float *__restrict__ out,
                                                    int S, int M, int Lq, int total_qm, unsigned *__restrict__ row_amax,
                                                    int ld_oa = 0, float2 *__restrict__ stats = nullptr, int band_major = 0)
{
  static_assert(P_ == 4, "one sampling point per lane of the quad");
  const int lb = xcd_chunked_block(blockIdx.x, gridDim.x);
  int qm = lb * 64 + (threadIdx.x >> 2);
  if constexpr (FUSED) {
    // band-major query order (see msda_fwd_d32): 8 queries x 8 heads per block here
    const int n0 = (int)(shapes[0] * shapes[1]), n1 = (int)(shapes[2] * shapes[3]), n2 = (int)(shapes[4] * shapes[5]);
    if (band_major && M == 8 && Lq == S && n0 + n1 + n2 == S && !((n0 | n1 | n2) & 63) && (gridDim.x & 7) == 0 && total_qm == (int)gridDim.x * 64) {
      const int per = gridDim.x >> 3, k = lb / per, p = (lb - k * per) * 8 + (threadIdx.x >> 5);
      const int sq = S >> 3, bimg = p / sq, r = p - bimg * sq;
      const int b0 = n0 >> 3, b1 = n1 >> 3, b2 = n2 >> 3;
      int q;
      if (r < b0) q = k * b0 + r;
      else if (r < b0 + b1) q = n0 + k * b1 + (r - b0);
      else q = n0 + n1 + k * b2 + (r - b0 - b1);
      qm = (bimg * S + q) * 8 + ((threadIdx.x >> 2) & 7);
    }
  }
  if (qm >= total_qm) return;                              // (uniform inside a quad)
  const int sub = threadIdx.x & 3;
  const int m = qm % M;
  const int b = (qm / M) / Lq;
  const int stride_w = M * 32;
  // ---- this lane's point of every level: location and attention weight
  float px[L_], py[L_], pa[L_];
  if constexpr (FUSED) {
    const int64_t bq = qm / M;
    const float *row = loc + bq * ld_oa;
    const float *offs = row + m * (L_ * P_ * 2) + sub * 2, *lg = row + M * (L_ * P_ * 2) + m * (L_ * P_) + sub;
    const float *refp = attn + bq * (L_ * 2);
    float mx = -INFINITY;
#pragma unroll
    for (int l = 0; l < L_; ++l) { pa[l] = lg[l * P_]; mx = fmaxf(mx, pa[l]); }
    mx = quad_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int l = 0; l < L_; ++l) { pa[l] = __expf(pa[l] - mx); sum += pa[l]; }
    const float inv = 1.f / quad_sum(sum);
#pragma unroll
    for (int l = 0; l < L_; ++l) pa[l] *= inv;
    if (stats && sub == 0) stats[qm] = make_float2(mx, inv);
#pragma unroll
    for (int l = 0; l < L_; ++l) {
      const float2 o = *reinterpret_cast<const float2 *>(offs + l * (P_ * 2));
      const float rx = refp[2 * l], ry = refp[2 * l + 1];
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const float fw = (float)W, fh = (float)H;
      if (((W & (W - 1)) | (H & (H - 1))) == 0) { px[l] = rx + o.x * (1.f / fw); py[l] = ry + o.y * (1.f / fh); }   // powers of two: the product IS the quotient
      else { px[l] = rx + o.x / fw; py[l] = ry + o.y / fh; }
    }
  } else {
#pragma unroll
    for (int l = 0; l < L_; ++l) {
      const float2 o = *reinterpret_cast<const float2 *>(loc + ((int64_t)qm * (L_ * P_) + l * P_ + sub) * 2);
      px[l] = o.x; py[l] = o.y;
      pa[l] = attn[(int64_t)qm * (L_ * P_) + l * P_ + sub];
    }
  }
  const float *zrow = g_msda_zero_row + sub * 4;
  float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
#pragma unroll
  for (int l = 0; l < L_; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const float *vbase = value + ((int64_t)b * S + lvl_start[l]) * stride_w + m * 32 + sub * 4;
    const float h_im = py[l] * H - 0.5f, w_im = px[l] * W - 0.5f;
    const bool in_range = h_im > -1 && w_im > -1 && h_im < H && w_im < W;
    int off[4]; bool ok[4]; float cw[4], lh, lw, hh, hw;
    corner_setup<float>(h_im, w_im, H, W, stride_w, in_range, off, ok, cw, lh, lw, hh, hw);
    float wk[4];
    int okm = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { wk[k] = ok[k] ? cw[k] * pa[l] : 0.f; okm |= ok[k] ? 1 << k : 0; }
    // ---- the level's four points in turn, two at a time in flight: lane p's geometry to the whole quad
    auto point = [&](auto PC, float4 (&v)[4][2], float (&w4)[4]) {
      constexpr int Pn = decltype(PC)::value;
      const int bm = quad_bcast_i<Pn>(okm);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int bo = quad_bcast_i<Pn>(off[k]);
        w4[k] = quad_bcast_f<Pn>(wk[k]);
        const float *src = (bm >> k) & 1 ? vbase + bo : zrow;
        v[k][0] = *reinterpret_cast<const float4 *>(src);
        v[k][1] = *reinterpret_cast<const float4 *>(src + 16);
      }
    };
    auto accumulate = [&](const float4 (&v)[4][2], const float (&w4)[4]) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc0.x += w4[k] * v[k][0].x; acc0.y += w4[k] * v[k][0].y; acc0.z += w4[k] * v[k][0].z; acc0.w += w4[k] * v[k][0].w;
        acc1.x += w4[k] * v[k][1].x; acc1.y += w4[k] * v[k][1].y; acc1.z += w4[k] * v[k][1].z; acc1.w += w4[k] * v[k][1].w;
      }
    };
    float4 va[4][2], vb[4][2];
    float wa[4], wb[4];
    point(std::integral_constant<int, 0>(), va, wa);
    point(std::integral_constant<int, 1>(), vb, wb);
    accumulate(va, wa);
    point(std::integral_constant<int, 2>(), va, wa);
    accumulate(vb, wb);
    point(std::integral_constant<int, 3>(), vb, wb);
    accumulate(va, wa);
    accumulate(vb, wb);
  }
  *reinterpret_cast<float4 *>(out + (int64_t)qm * 32 + sub * 4) = acc0;
  *reinterpret_cast<float4 *>(out + (int64_t)qm * 32 + 16 + sub * 4) = acc1;
  if (row_amax) {
    float mx = fmaxf(fmaxf(fmaxf(fabsf(acc0.x), fabsf(acc0.y)), fmaxf(fabsf(acc0.z), fabsf(acc0.w))),
                     fmaxf(fmaxf(fabsf(acc1.x), fabsf(acc1.y)), fmaxf(fabsf(acc1.z), fabsf(acc1.w))));
    mx = quad_max(mx);
    if (M == 8) {                                          // a half wavefront = the eight heads of one query: reduce across them, one plain store
      mx = fmaxf(mx, __shfl_xor(mx, 4, 64)); mx = fmaxf(mx, __shfl_xor(mx, 8, 64)); mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      if ((threadIdx.x & 31) == 0) row_amax[qm / M] = __float_as_uint(mx);
    } else if (sub == 0 && mx > 0.f) {
      atomicMax(row_amax + qm / M, __float_as_uint(mx));
    }
  }
}

// ----------------------------------------------------------------------------------------- fast backward
template <int L_, int P_, int SCOPE>
__global__ __launch_bounds__(256) void msda_bwd_d32(const float *__restrict__ value, const int64_t *__restrict__ shapes,
                                                     const int64_t *__restrict__ lvl_start, const float *__restrict__ loc,
                                                     const float *__restrict__ attn, const float *__restrict__ grad_out,
                                                     float *__restrict__ grad_value, float *__restrict__ grad_loc,
                                                     float *__restrict__ grad_attn, int S, int M, int Lq, int total_qm)
{
  const int lb = xcd_chunked_block(blockIdx.x, gridDim.x);
  const int qm = lb * 32 + (threadIdx.x >> 3);
  if (qm >= total_qm) return;   // whole 8-lane groups leave together; DPP stays inside a group
  const int sub = threadIdx.x & 7;
  const int m = qm % M;
  const int b = (qm / M) / Lq;
  const int stride_w = M * 32;
  constexpr int LP = L_ * P_;
  const float4 *lp4 = reinterpret_cast<const float4 *>(loc + (int64_t)qm * (LP * 2));
  const float4 *ap4 = reinterpret_cast<const float4 *>(attn + (int64_t)qm * LP);
  float locs[LP * 2], aw[LP];
#pragma unroll
  for (int i = 0; i < LP * 2 / 4; ++i) {
    float4 t = lp4[i];
    locs[4 * i] = t.x; locs[4 * i + 1] = t.y; locs[4 * i + 2] = t.z; locs[4 * i + 3] = t.w;
  }
#pragma unroll
  for (int i = 0; i < LP / 4; ++i) {
    float4 t = ap4[i];
    aw[4 * i] = t.x; aw[4 * i + 1] = t.y; aw[4 * i + 2] = t.z; aw[4 * i + 3] = t.w;
  }
  const float4 go = *reinterpret_cast<const float4 *>(grad_out + (int64_t)qm * 32 + sub * 4);
  // results parked per lane: lane `sub` keeps points sub and sub+8 (LP <= 16)
  static_assert(LP <= 16, "fast path parks at most 16 points");
  float keep_a[2] = {0.f, 0.f}, keep_x[2] = {0.f, 0.f}, keep_y[2] = {0.f, 0.f};
#pragma unroll
  for (int l = 0; l < L_; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const int64_t voff = ((int64_t)b * S + lvl_start[l]) * stride_w + m * 32 + sub * 4;
    const float *vbase = value + voff;
    float *gbase = grad_value + voff;
#pragma unroll
    for (int p = 0; p < P_; ++p) {
      const int i = l * P_ + p;
      const float x = locs[2 * i], y = locs[2 * i + 1], a = aw[i];
      const float h_im = y * H - 0.5f, w_im = x * W - 0.5f;
      const bool in_range = h_im > -1 && w_im > -1 && h_im < H && w_im < W;
      int off[4]; bool ok[4]; float cw[4], lh, lw, hh, hw;
      corner_setup<float>(h_im, w_im, H, W, stride_w, in_range, off, ok, cw, lh, lw, hh, hw);
      float4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float4 t = *reinterpret_cast<const float4 *>(vbase + off[k]);
        v[k] = ok[k] ? t : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      // d(val)/dh and d(val)/dw per channel (reference .cuh:122-158), then dot with grad_out
      const float gs[4] = {go.x, go.y, go.z, go.w};
      float pa = 0.f, ph = 0.f, pw = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float v1 = (&v[0].x)[c], v2 = (&v[1].x)[c], v3 = (&v[2].x)[c], v4 = (&v[3].x)[c];
        const float gh = -hw * v1 - lw * v2 + hw * v3 + lw * v4;
        const float gw = -hh * v1 + hh * v2 - lh * v3 + lh * v4;
        const float val = cw[0] * v1 + cw[1] * v2 + cw[2] * v3 + cw[3] * v4;
        pa += gs[c] * val;
        ph += gs[c] * gh;
        pw += gs[c] * gw;
      }
      // scatter-add grad_value (reference .cuh:130-157 atomicAdd)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (ok[k]) {
          const float s = cw[k] * a;
          float *g = gbase + off[k];
          scoped_add<SCOPE>(g + 0, s * go.x);
          scoped_add<SCOPE>(g + 1, s * go.y);
          scoped_add<SCOPE>(g + 2, s * go.z);
          scoped_add<SCOPE>(g + 3, s * go.w);
        }
      }
      pa = group8_sum(pa);
      pw = group8_sum(pw) * (a * W);
      ph = group8_sum(ph) * (a * H);
      if (sub == (i & 7)) { keep_a[i >> 3] = pa; keep_x[i >> 3] = pw; keep_y[i >> 3] = ph; }
    }
  }
  float *ga = grad_attn + (int64_t)qm * LP;
  float2 *gl = reinterpret_cast<float2 *>(grad_loc + (int64_t)qm * LP * 2);
  if (sub < LP) { ga[sub] = keep_a[0]; gl[sub] = make_float2(keep_x[0], keep_y[0]); }
  if (sub + 8 < LP) { ga[sub + 8] = keep_a[1]; gl[sub + 8] = make_float2(keep_x[1], keep_y[1]); }
}

// ----------------------------------------------------------------------------------------- tiled backward
// Self-attention geometry (num_query == spatial_size, query q sits at a pixel of
// some level).  Measured on MI355X (tools/probes/lds_atomic_probe.hip, bench_msda.py):
//   * global fp32 atomics retire ~1 dword / clock / L2 channel: the 528 M dword
//     atomics of one config-2 backward take 2.2 ms with full-line lanes and 6.4 ms
//     with float4 lanes — so grad_value is NOT scattered to HBM sample by sample;
//   * ds_add_f32 (LDS float atomic) runs at 0.38 lanes/clk/CU, 42x slower than
//     ds_add_u32 (16 lanes/clk/CU) — so the LDS accumulator is INTEGER.
// The unit square is cut into G x G tiles; one workgroup owns (batch, tile, dest
// level, head): it walks the queries whose pixel centre lies in the tile (all
// source levels) and accumulates their contributions to the destination level
// into an LDS window (tile + HALO cells each side, <= WIN x WIN cells x 32
// channels) as int32 fixed point with a per-(workgroup, channel) power-of-two
// scale 2^e, e = 22 - exponent(max |grad_out| over the tile's queries): every
// contribution is |grad_out*attn*w| <= max < 2^22 after scaling and the bilinear
// x attention weights of one tile's queries sum to < 2^9 per cell, so the int32
// sum cannot overflow; quantisation error is <= 2^-23 of that max per add (the
// same order as the fp32 re-association noise of the reference's atomics) and
// the accumulation is order-independent.  The window is flushed once with
// full-128-byte-line fp32 atomics.  Samples that land outside the window (large
// learned offsets) take the direct global atomic, so the result never depends on
// G, HALO or WIN — the window is only a cache.
// G is derived in-kernel from the device-resident shapes (the ABI carries no host
// copy): G = ceil(max level dim / TILE); the host launches GMAX^2 tiles and the
// surplus blocks exit at once.
constexpr int kTile = 16, kHalo = 4, kWin = kTile + 2 * kHalo, kGmax = 16;

__device__ __forceinline__ int lds_slot(int cell, int ch) { return cell * 32 + ((ch + (cell & 3)) & 31); }

template <int P_, int ABL>
__global__ __launch_bounds__(1024) void msda_bwd_tiled_d32(const float *__restrict__ value, const int64_t *__restrict__ shapes,
                                                           const int64_t *__restrict__ lvl_start, const float *__restrict__ loc,
                                                           const float *__restrict__ attn, const float *__restrict__ grad_out,
                                                           float *__restrict__ grad_value, float *__restrict__ grad_loc,
                                                           float *__restrict__ grad_attn, int S, int M, int L)
{
  static_assert(P_ == 4, "tiled path is written for 4 points per level");
  extern __shared__ __attribute__((aligned(16))) int win[];   // kWin*kWin*32 int32 accumulators + 32 max slots
  // ---- decode block -> (b, tile, dest level l, head m); heads fastest so the 8 XCDs split the heads
  int idx = blockIdx.x;
  const int m = idx % M; idx /= M;
  const int l = idx % L; idx /= L;
  const int tile = idx % (kGmax * kGmax);
  const int b = idx / (kGmax * kGmax);
  int maxdim = 1;
  for (int j = 0; j < L; ++j) maxdim = max(maxdim, max((int)shapes[2 * j], (int)shapes[2 * j + 1]));
  const int G = min(kGmax, (maxdim + kTile - 1) / kTile);
  if (tile >= G * G) return;
  const int ty = tile / G, tx = tile % G;
  const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
  // destination window (clipped); at most kWin x kWin cells are cached
  const int wy0 = max(0, ty * H / G - kHalo), wx0 = max(0, tx * W / G - kHalo);
  const int wh = min(min(H, (ty + 1) * H / G + kHalo) - wy0, kWin), ww = min(min(W, (tx + 1) * W / G + kHalo) - wx0, kWin);
  const int cells = max(wh, 0) * max(ww, 0);
  int *chmax = win + kWin * kWin * 32;
  const int NT = blockDim.x, NG = NT >> 3;      // threads, 8-lane query groups per workgroup
  for (int i = threadIdx.x; i < cells * 32; i += NT) win[i] = 0;
  if (threadIdx.x < 32) chmax[threadIdx.x] = 0;
  __syncthreads();

  const int sub = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const int stride_w = M * 32;
  const int LP = L * P_;
  const int64_t voff = ((int64_t)b * S + lvl_start[l]) * stride_w + m * 32 + sub * 4;
  const float *vbase = value + voff;
  float *gbase = grad_value + voff;
  // ---- pre-pass: per-channel max |grad_out| over this tile's queries -> fixed-point scale
  {
    float mx[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < L; ++j) {
      const int Hj = (int)shapes[2 * j], Wj = (int)shapes[2 * j + 1];
      const int ry0 = ty * Hj / G, ry1 = (ty + 1) * Hj / G, rx0 = tx * Wj / G, rx1 = (tx + 1) * Wj / G;
      const int rw = rx1 - rx0, nq = (ry1 - ry0) * rw;
      const int qbase = (int)lvl_start[j];
      for (int i = grp; i < nq; i += NG) {
        const int q = qbase + (ry0 + i / rw) * Wj + rx0 + i % rw;
        const float4 go = *reinterpret_cast<const float4 *>(grad_out + (((int64_t)b * S + q) * M + m) * 32 + sub * 4);
        mx[0] = fmaxf(mx[0], fabsf(go.x)); mx[1] = fmaxf(mx[1], fabsf(go.y));
        mx[2] = fmaxf(mx[2], fabsf(go.z)); mx[3] = fmaxf(mx[3], fabsf(go.w));
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)   // non-negative floats order like their bit patterns; NaN/inf handled below
      atomicMax(&chmax[sub * 4 + c], __float_as_int(mx[c]));
  }
  __syncthreads();
  float qscale[4], dscale[4];
  bool fixed_ok = true;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float mxc = __int_as_float(chmax[sub * 4 + c]);
    int ex;
    (void)frexpf(mxc, &ex);                       // mxc < 2^ex
    const bool fin = mxc > 0.f && mxc < 3.0e38f;  // finite, non-zero
    ex = fin ? min(max(ex, -100), 100) : 0;
    qscale[c] = ldexpf(1.f, 22 - ex);
    dscale[c] = ldexpf(1.f, ex - 22);
    fixed_ok = fixed_ok && (fin || mxc == 0.f);
  }
  // a non-finite grad_out anywhere in the tile: bypass the window (plain global atomics propagate it)
  fixed_ok = __all(fixed_ok);
  for (int j = 0; j < L; ++j) {
    const int Hj = (int)shapes[2 * j], Wj = (int)shapes[2 * j + 1];
    const int ry0 = ty * Hj / G, ry1 = (ty + 1) * Hj / G, rx0 = tx * Wj / G, rx1 = (tx + 1) * Wj / G;
    const int rw = rx1 - rx0, nq = (ry1 - ry0) * rw;
    const int qbase = (int)lvl_start[j];
    for (int i = grp; i < nq; i += NG) {
      const int q = qbase + (ry0 + i / rw) * Wj + rx0 + i % rw;
      const int64_t qm = ((int64_t)b * S + q) * M + m;
      const float4 *lp4 = reinterpret_cast<const float4 *>(loc + (qm * LP + l * P_) * 2);
      const float4 l01 = lp4[0], l23 = lp4[1];
      const float4 a4 = *reinterpret_cast<const float4 *>(attn + qm * LP + l * P_);
      const float4 go = *reinterpret_cast<const float4 *>(grad_out + qm * 32 + sub * 4);
      const float locs[8] = {l01.x, l01.y, l01.z, l01.w, l23.x, l23.y, l23.z, l23.w};
      const float aw[4] = {a4.x, a4.y, a4.z, a4.w};
      const float gs[4] = {go.x, go.y, go.z, go.w};
      float keep_a = 0.f, keep_x = 0.f, keep_y = 0.f;
#pragma unroll
      for (int p = 0; p < P_; ++p) {
        const float x = locs[2 * p], y = locs[2 * p + 1], a = aw[p];
        const float h_im = y * H - 0.5f, w_im = x * W - 0.5f;
        const bool in_range = h_im > -1 && w_im > -1 && h_im < H && w_im < W;
        int off[4]; bool ok[4]; float cw[4], lh, lw, hh, hw;
        corner_setup<float>(h_im, w_im, H, W, stride_w, in_range, off, ok, cw, lh, lw, hh, hw);
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float4 t = *reinterpret_cast<const float4 *>(vbase + off[k]);
          v[k] = ok[k] ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float pa = 0.f, ph = 0.f, pw = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float v1 = (&v[0].x)[c], v2 = (&v[1].x)[c], v3 = (&v[2].x)[c], v4 = (&v[3].x)[c];
          pa += gs[c] * (cw[0] * v1 + cw[1] * v2 + cw[2] * v3 + cw[3] * v4);
          ph += gs[c] * (-hw * v1 - lw * v2 + hw * v3 + lw * v4);
          pw += gs[c] * (-hh * v1 + hh * v2 - lh * v3 + lh * v4);
        }
        // window coordinates of the low corner (valid when in_range; clamped offsets otherwise unused)
        const int h_low = in_range ? (int)floorf(h_im) : 0, w_low = in_range ? (int)floorf(w_im) : 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (ok[k]) {
            const float sc = cw[k] * a;
            const int cy = h_low + (k >> 1) - wy0, cx = w_low + (k & 1) - wx0;
            if (fixed_ok && cy >= 0 && cy < wh && cx >= 0 && cx < ww) {
              const int cell = cy * ww + cx;
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const int fx = __float2int_rn(sc * gs[c] * qscale[c]);
                if (ABL & 1) asm volatile("" ::"v"(cell), "v"(fx));
                else atomicAdd(&win[lds_slot(cell, sub * 4 + c)], fx);   // ds_add_u32
              }
            } else {
              float *g = gbase + off[k];
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                if (ABL & 2) asm volatile("" ::"v"(g), "v"(sc * gs[c]));
                else unsafeAtomicAdd(g + c, sc * gs[c]);
              }
            }
          }
        }
        pa = group8_sum(pa);
        pw = group8_sum(pw) * (a * W);
        ph = group8_sum(ph) * (a * H);
        if (sub == p) { keep_a = pa; keep_x = pw; keep_y = ph; }
      }
      if (sub < P_) {
        grad_attn[qm * LP + l * P_ + sub] = keep_a;
        reinterpret_cast<float2 *>(grad_loc + (qm * LP + l * P_) * 2)[sub] = make_float2(keep_x, keep_y);
      }
    }
  }
  __syncthreads();
  // ---- flush: lane = channel, one full 128-byte line per cell per half-wave
  float *gl = grad_value + ((int64_t)b * S + lvl_start[l]) * stride_w + m * 32;
  for (int i = threadIdx.x; i < cells * 32; i += NT) {
    const int cell = i >> 5, ch = i & 31;
    const int acc = win[lds_slot(cell, ch)];
    int ex;
    (void)frexpf(__int_as_float(chmax[ch]), &ex);
    const float v = (float)acc * ldexpf(1.f, min(max(ex, -100), 100) - 22);
    if (acc != 0 && !(ABL & 4)) unsafeAtomicAdd(gl + ((int64_t)(wy0 + cell / ww) * W + wx0 + cell % ww) * stride_w + ch, v);
  }
}

// ----------------------------------------------------------------------------------------- owner backward (all levels per workgroup)
// What the per-destination-level kernel above was bound by, measured with its ablation switches (tools/bench_msda.py): with
// every atomic removed the pass still took 0.30 of its 0.38 ms, without the value gathers as well 0.27, without the stores
// 0.27, without the main loop 0.04 — the loop is bound by VECTOR INSTRUCTION ISSUE, not by memory or atomics: every one of
// the 8 lanes of a (query, head) repeats the coordinate / corner / window arithmetic of a point, every LDS add paid for
// its own slot computation and float -> int conversion, and grad_loc / grad_attn cost 12 FMAs per channel.  This kernel
//   * gives ONE workgroup per (batch, tile, head) the destination windows of ALL three levels (standard pyramid at G = 8
//     with a halo of 5 cells: 26x26 + 18x18 + 14x14 cells x 132 B = 154 KB of the 160 KB LDS), so a query's sampling
//     locations, attention weights and grad_out are read by one workgroup instead of three;
//   * gives a (query, head) to 4 lanes with 8 channels each (the lane-redundant part of the work halves, the cross-lane
//     sums are 2 DPP steps) and walks (query, level) units: 1008 units over 256 lane groups keep 98 % of the lanes busy,
//     and neighbouring lane groups work on different levels, i.e. on different windows (fewer same-cell collisions);
//   * computes grad_attn / grad_loc from the four corner dot products S_k = <grad_out, value row k> (they are linear in
//     them): 4 packed FMAs per corner and lane, then 12 operations per point;
//   * rounds a contribution to its fixed-point integer with ONE fused multiply-add (x + 1.5 * 2^23 lands on the integer
//     grid; bit pattern = 0x4B400000 + integer) and does NOT subtract the offset: the int32 slots wrap modulo 2^32 and the
//     flush removes count * 0x4B400000, a 33rd word per cell counting the corner contributions it received;
//   * addresses a cell as cell * 33 + channel: the lane's eight channels are immediate offsets of the ds_add, and the odd
//     pitch staggers neighbouring cells over the banks (the 16 lane groups of a wavefront meet 2 to a bank, the minimum
//     for 64 lanes; an XOR swizzle by multiples of 4 left them 8 to a bank);
//   * takes the common case — all four corners inside the map and inside the window — as straight-line code, per-corner
//     tests only otherwise, and moves the direct-global-atomic fallback (corner outside every window: large learned
//     offset) out of the loop into a rolled epilogue that re-reads the point, so it costs the loop no registers;
//   * uses 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate) for pixel / element indices — the host admits the
//     kernel only when they fit.
// Same fixed-point scheme and guarantees as above (order-independent, cannot overflow, result independent of tile / halo
// sizes).  Levels whose window does not fit the LDS budget (non-pyramid shapes) get no window: direct atomics.
// 0.378 -> 0.233 ms per launch at config 2 (tools/bench_msda.py, default spread).
constexpr int kHalo4 = 5, kWin4 = kTile + 2 * kHalo4, kOwnCells4 = 1200;
// HALF (round 3): the same kernel with the 32 channels of a head split over TWO workgroups.  A cell is then 17 words instead of
// 33, the same LDS holds 2 330 cells, and the halo grows from 5 to 9 cells (34 x 34 + 26 x 26 + 22 x 22 at the standard
// pyramid).  Every workgroup still forms the full 32-channel corner dot products (grad_attn / grad_loc need them; only the
// half-0 workgroup stores them) but accumulates just its 16 channels, so the gather / geometry work doubles: ~0.4 ms per launch
// instead of 0.23 — and stays there up to offsets of ~4 cells sigma, where the halo-5 kernel spends 2 ms in its global-atomic
// fallback (40 % of the corners miss its windows).  pd_msda_backward picks per launch from a sampled miss count (below).
constexpr int kHalo4H = 9, kWin4H = kTile + 2 * kHalo4H, kOwnCells4H = 2330;

__device__ __forceinline__ float group4_sum(float x)
{
  x = dpp_add<0xB1>(x);   // quad_perm [1,0,3,2]
  x = dpp_add<0x4E>(x);   // quad_perm [2,3,0,1]
  return x;
}

// FUSED (round 5, pd_msda_fused_backward): the module path's form of this kernel.  `loc` is the RAW projection output the fused forward
// read ([B S, ld_oa]: offsets | logits, see msda_fwd_d32), `attn` is unused; a unit re-forms its level's sampling locations
// (reference point + offset / (W, H)) and attention probabilities (exp(logit - max) / sum from the forward's 8-byte statistics) in
// registers, and instead of grad_sampling_loc / grad_attn_weight it writes the gradient of that projection output directly
// (`f.d_oa`, same column layout):
//   d offset = grad_loc / (W, H)                                       (ms_deform_attn.py:114-117 backwards)
//   d logit  = a (g - sum_j a_j g_j) with sum_j a_j g_j = <grad_out, out>   (softmax backwards; g_j = <grad_out, sampled value j>, and
//              out = sum_j a_j (sampled value j) is the forward's output row `f.fout`: one 32-channel dot product per unit, no pass
//              over the other levels' units)
// plus the rows' absolute maxima for the two-plane GEMM that reads d_oa (atomic max into the zero-filled f.doa_amax).  The
// msda_prep_bwd launch and the 144-byte-per-(query, head) round trip of grad_loc / grad_attn are gone.
struct FusedBwd {
  const float *ref = nullptr;          // [B S, L, 2]
  const float2 *stats = nullptr;       // [B S M] {max logit, 1 / sum exp}
  const float *fout = nullptr;         // [B S, M 32] the forward's output
  float *d_oa = nullptr;               // [B S, ld_doa]
  unsigned *doa_amax = nullptr;        // [B S] or null
  float *gdot = nullptr;               // [B S M L] scratch: sum over a level's points of a g per (query, head, level), written and read by the owning workgroup
  int ld_oa = 0, ld_doa = 0;
};

__device__ __forceinline__ float group4_max(float x)
{
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true)));
  x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true)));
  return x;
}

template <int ABL, bool HALF = false, bool FUSED = false>
__global__ __launch_bounds__(1024) void msda_bwd_owner4_d32(const float *__restrict__ value, const int64_t *__restrict__ shapes,
                                                            const int64_t *__restrict__ lvl_start, const float *__restrict__ loc,
                                                            const float *__restrict__ attn, const float *__restrict__ grad_out,
                                                            float *__restrict__ grad_value, float *__restrict__ grad_loc,
                                                            float *__restrict__ grad_attn, int S, int M, int B,
                                                            unsigned *__restrict__ cnt, unsigned long long *__restrict__ pub,
                                                            FusedBwd f = FusedBwd())
{
  constexpr int L = 3, P = 4, LP = L * P;
  constexpr int HALO = HALF ? kHalo4H : kHalo4, WIN = HALF ? kWin4H : kWin4, CELLS = HALF ? kOwnCells4H : kOwnCells4, CW = HALF ? 17 : 33;
  // cnt / pub (optional): the kernel counts the sample points that leave the halo-5 windows (cnt[0]) among those it looked at
  // (cnt[1]); the last workgroup to finish (ticket cnt[2]) publishes both to host-mapped memory and clears the slot.  The host
  // picks the variant of a LATER launch from what earlier launches published — no second launch, no synchronisation.
  extern __shared__ __attribute__((aligned(16))) int win[];   // CELLS cells x CW words + 32 max slots + 32 scales
  int Hs[L], Ws[L], ls[L], maxdim = 1;
#pragma unroll
  for (int j = 0; j < L; ++j) {
    Hs[j] = (int)shapes[2 * j]; Ws[j] = (int)shapes[2 * j + 1]; ls[j] = (int)lvl_start[j];
    maxdim = max(maxdim, max(Hs[j], Ws[j]));
  }
  const int G = min(kGmax, (maxdim + kTile - 1) / kTile);
  // block -> (batch, tile, head), heads fastest, over the ACTIVE tiles only (the host launches kGmax^2 per image: it cannot
  // see G) and XCD-chunked (block i runs on XCD i % 8): an XCD owns whole (batch, tile)s with all their heads, so a
  // query's loc / attn rows (all heads in one 768 / 384-byte run) are fetched into ONE L2
  const int nact = B * G * G * M * (HALF ? 2 : 1);
  int idx;
  if ((nact & 7) == 0) {
    const int per = nact >> 3, k = blockIdx.x & 7, j = blockIdx.x >> 3;
    if (j >= per) return;
    idx = k * per + j;
  } else {
    if ((int)blockIdx.x >= nact) return;
    idx = blockIdx.x;
  }
  const int half = HALF ? idx & 1 : 0;
  if (HALF) idx >>= 1;
  const int m = idx % M; idx /= M;
  const int tile = idx % (G * G);
  const int b = idx / (G * G);
  const int ty = tile / G, tx = tile % G;
  struct LvlP { int H, W, ls, wy0, wx0, wh, ww, wbase, dy5, dx5, wh5, ww5; };   // (dy5 ..: the halo-5 window inside a halo-9 one)
  __shared__ LvlP lvp[L];
  struct QLv { int rw, w, base, first, n, pad0, pad1, pad2; };   // the tile's queries of source level j: rows of rw pixels in a map of width w, from query `base`; `first` = queries of the levels before
  __shared__ QLv lq[L];
  __shared__ int s_used, s_fixed_ok, s_miss;
  if (threadIdx.x == 0) {
    s_miss = 0;
    int used = 0;
#pragma unroll
    for (int j = 0; j < L; ++j) {
      LvlP t;
      t.H = Hs[j]; t.W = Ws[j]; t.ls = ls[j];
      t.wy0 = max(0, ty * Hs[j] / G - HALO); t.wx0 = max(0, tx * Ws[j] / G - HALO);
      t.wh = max(0, min(min(Hs[j], (ty + 1) * Hs[j] / G + HALO) - t.wy0, WIN));
      t.ww = max(0, min(min(Ws[j], (tx + 1) * Ws[j] / G + HALO) - t.wx0, WIN));
      {
        const int y5 = max(0, ty * Hs[j] / G - kHalo4), x5 = max(0, tx * Ws[j] / G - kHalo4);
        t.dy5 = y5 - t.wy0; t.dx5 = x5 - t.wx0;
        t.wh5 = max(0, min(min(Hs[j], (ty + 1) * Hs[j] / G + kHalo4) - y5, kWin4));
        t.ww5 = max(0, min(min(Ws[j], (tx + 1) * Ws[j] / G + kHalo4) - x5, kWin4));
      }
      if (used + t.wh * t.ww > CELLS) { t.wh = 0; t.ww = 0; }     // no window: direct atomics for this level
      t.wbase = used;
      used += t.wh * t.ww;
      lvp[j] = t;
    }
    s_used = used;
    s_fixed_ok = 1;
    int first = 0;
#pragma unroll
    for (int j = 0; j < L; ++j) {
      const int ry0 = ty * Hs[j] / G, rx0 = tx * Ws[j] / G, rw = (tx + 1) * Ws[j] / G - rx0, n = ((ty + 1) * Hs[j] / G - ry0) * rw;
      lq[j] = QLv{rw, Ws[j], ls[j] + ry0 * Ws[j] + rx0, first, n, 0, 0, 0};
      first += n;
    }
  }
  int *chmax = win + CELLS * CW;
  float *chscale = reinterpret_cast<float *>(chmax + 32);
  const int NT = blockDim.x;
  if (threadIdx.x < 32) chmax[threadIdx.x] = 0;
  __syncthreads();
  {
    int4 *w4 = reinterpret_cast<int4 *>(win);
    const int n4 = (s_used * CW + 3) >> 2;
    for (int i = threadIdx.x; i < n4; i += NT) w4[i] = make_int4(0, 0, 0, 0);
  }
  const int stride_w = M * 32;
  // the tile's queries: source level j contributes the pixels [ry0, ry1) x [rx0, rx1)
  // (level constants of a query in LDS, not in arrays: of `j == 0 ? rw[0] : j == 1 ? rw[1] : rw[2]` the compiler made rw[j] on a STACK array — the
  // kernel sits at its limit of scalar registers as well as vector ones — i.e. a scratch load in front of every query's loads, in the pre-pass, the
  // main loop and the epilogue; as separate scalars they became a scratch table again)
  const int nq0 = __builtin_amdgcn_readfirstlane(lq[0].n), nq01 = nq0 + __builtin_amdgcn_readfirstlane(lq[1].n);
  const int nq_all = nq01 + __builtin_amdgcn_readfirstlane(lq[2].n);
  auto query_of = [&](int i) {
    const QLv t = lq[(i >= nq0 ? 1 : 0) + (i >= nq01 ? 1 : 0)];
    const int il = i - t.first, y = il / t.rw, x = il - y * t.rw;
    return t.base + y * t.w + x;
  };
  // ---- pre-pass: per-channel max |grad_out| over the tile's queries (8 lanes x float4 per query row) -> fixed-point scale
  {
    const int sub8 = threadIdx.x & 7, grp8 = threadIdx.x >> 3, NG8 = NT >> 3;
    float mx[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = grp8; i < nq_all; i += NG8) {
      const int q = query_of(i);
      const float4 go = *reinterpret_cast<const float4 *>(grad_out + (((int64_t)b * S + q) * M + m) * 32 + sub8 * 4);
      mx[0] = fmaxf(mx[0], fabsf(go.x)); mx[1] = fmaxf(mx[1], fabsf(go.y));
      mx[2] = fmaxf(mx[2], fabsf(go.z)); mx[3] = fmaxf(mx[3], fabsf(go.w));
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) atomicMax(&chmax[sub8 * 4 + c], __float_as_int(mx[c]));   // non-negative floats order like their bits
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const float mxc = __int_as_float(chmax[threadIdx.x]);
    int ex;
    (void)frexpf(mxc, &ex);                             // mxc < 2^ex
    const bool fin = mxc > 0.f && mxc < 3.0e38f;
    ex = fin ? min(max(ex, -100), 100) : 0;
    chscale[threadIdx.x] = ldexpf(1.f, 22 - ex);
    if (!(fin || mxc == 0.f)) s_fixed_ok = 0;           // a non-finite grad_out in the tile: bypass the windows
  }
  __syncthreads();
  const bool fixed_ok = s_fixed_ok != 0;
  // ---- main pass: a (query, level) unit per 4-lane group; lane `sub` owns channels [8 sub, 8 sub + 8)
  const int sub = threadIdx.x & 3, grp = threadIdx.x >> 2, NG = NT >> 2;
  const int nunits = nq_all * L;
  unsigned nmiss = 0;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  for (int u = grp; u < nunits; u += NG) {
    const int qi = u / L, l = u - qi * L;
    const int q = query_of(qi);
    int mv = m;                                            // (likewise the head index: its 64-bit copy was a spilled scalar pair)
    asm volatile("" : "+v"(mv));
    const int64_t qm = ((int64_t)b * S + q) * M + mv;
    const LvlP lv = lvp[l];
    const int H = lv.H, W = lv.W;
    // (the lane's channel offset made opaque per unit: `pointer + 8 sub` and `32 m + 8 sub` as loop-invariant 64-bit lane values were spilled and
    // reloaded at the top of every unit — the kernel has no register to carry them in)
    int subv = sub;
    asm volatile("" : "+v"(subv));
    const float4 g0 = *reinterpret_cast<const float4 *>(grad_out + qm * 32 + subv * 8), g1 = *reinterpret_cast<const float4 *>(grad_out + qm * 32 + subv * 8 + 4);
    const float gs[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const float4 q0 = *reinterpret_cast<const float4 *>(chscale + sub * 8), q1 = *reinterpret_cast<const float4 *>(chscale + sub * 8 + 4);
    const f32x2 gq2[4] = {{g0.x * q0.x, g0.y * q0.y}, {g0.z * q0.z, g0.w * q0.w}, {g1.x * q1.x, g1.y * q1.y}, {g1.z * q1.z, g1.w * q1.w}};
    const f32x2 gs2[4] = {{gs[0], gs[1]}, {gs[2], gs[3]}, {gs[4], gs[5]}, {gs[6], gs[7]}};
    const f32x2 magic2 = {12582912.f, 12582912.f};
    float4 l01, l23, a4;
    if constexpr (FUSED) {
      const int bq = b * S + q;
      const float *row = loc + (int64_t)bq * f.ld_oa + mv * LP + l * P;        // (+ the same again for the offsets: two floats per point)
      const float4 lg = *reinterpret_cast<const float4 *>(row + M * (LP * 2));
      const float2 st = f.stats[qm];
      a4 = make_float4(__expf(lg.x - st.x) * st.y, __expf(lg.y - st.x) * st.y, __expf(lg.z - st.x) * st.y, __expf(lg.w - st.x) * st.y);
      const float4 *op4 = reinterpret_cast<const float4 *>(row + mv * LP + l * P);
      l01 = op4[0]; l23 = op4[1];
      const float2 rf = *reinterpret_cast<const float2 *>(f.ref + ((int64_t)bq * L + l) * 2);
      const float fw = (float)W, fh = (float)H;
      if (((W & (W - 1)) | (H & (H - 1))) == 0) {          // powers of two: the product with 1 / W is the quotient, bit for bit
        const float inv_w = 1.f / fw, inv_h = 1.f / fh;
        l01 = make_float4(rf.x + l01.x * inv_w, rf.y + l01.y * inv_h, rf.x + l01.z * inv_w, rf.y + l01.w * inv_h);
        l23 = make_float4(rf.x + l23.x * inv_w, rf.y + l23.y * inv_h, rf.x + l23.z * inv_w, rf.y + l23.w * inv_h);
      } else {
        l01 = make_float4(rf.x + l01.x / fw, rf.y + l01.y / fh, rf.x + l01.z / fw, rf.y + l01.w / fh);
        l23 = make_float4(rf.x + l23.x / fw, rf.y + l23.y / fh, rf.x + l23.z / fw, rf.y + l23.w / fh);
      }
    } else {
      const float4 *lp4 = reinterpret_cast<const float4 *>(loc + (qm * LP + l * P) * 2);
      l01 = lp4[0]; l23 = lp4[1];
      a4 = *reinterpret_cast<const float4 *>(attn + qm * LP + l * P);
    }
    const float locs[8] = {l01.x, l01.y, l01.z, l01.w, l23.x, l23.y, l23.z, l23.w};
    const float aw[4] = {a4.x, a4.y, a4.z, a4.w};
    const int64_t voff = ((int64_t)b * S + lv.ls) * stride_w + mv * 32 + subv * 8;
    const float *vbase = value + voff;
    float *gbase = grad_value + voff;
    float keep_a = 0.f, keep_x = 0.f, keep_y = 0.f;
    unsigned slow = 0, out5 = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float a = aw[p];
      const float h_im = locs[2 * p + 1] * H - 0.5f, w_im = locs[2 * p] * W - 0.5f;
      const bool in_range = h_im > -1 && w_im > -1 && h_im < H && w_im < W;
      // reference .cuh:38-89 geometry, written for the instruction count (24-bit multiplies, one clamp per coordinate)
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h_low = in_range ? (int)hf : 0, w_low = in_range ? (int)wf : 0;
      float lh = h_im - hf, lw = w_im - wf;
      if (!in_range) { lh = 0.f; lw = 0.f; }
      float hh = 1.f - lh, hw = 1.f - lw;
      if (!in_range) { hh = 0.f; hw = 0.f; }              // NaN / inf / far locations contribute exactly nothing
      const bool hl = h_low >= 0, wl = w_low >= 0, hhi = h_low + 1 <= H - 1, whi = w_low + 1 <= W - 1;
      const bool ok[4] = {in_range && hl && wl, in_range && hl && whi, in_range && hhi && wl, in_range && hhi && whi};
      const int hlc = min(max(h_low, 0), H - 1), hhc = min(max(h_low + 1, 0), H - 1);
      const int wlc = min(max(w_low, 0), W - 1), whc = min(max(w_low + 1, 0), W - 1);
      const int r0 = __mul24(hlc, W), r1 = __mul24(hhc, W);
      const int off[4] = {__mul24(r0 + wlc, stride_w), __mul24(r0 + whc, stride_w), __mul24(r1 + wlc, stride_w), __mul24(r1 + whc, stride_w)};
      const float cw[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
      // a masked corner is dropped AFTER its dot product (select: a NaN in a row the reference would never read cannot leak)
      float Sk[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 t0 = *reinterpret_cast<const float4 *>(vbase + off[k]), t1 = *reinterpret_cast<const float4 *>(vbase + off[k] + 4);
        f32x2 d = gs2[0] * f32x2{t0.x, t0.y};
        d = __builtin_elementwise_fma(gs2[1], f32x2{t0.z, t0.w}, d);
        d = __builtin_elementwise_fma(gs2[2], f32x2{t1.x, t1.y}, d);
        d = __builtin_elementwise_fma(gs2[3], f32x2{t1.z, t1.w}, d);
        Sk[k] = ok[k] ? d.x + d.y : 0.f;
      }
      float pa = cw[0] * Sk[0] + cw[1] * Sk[1] + cw[2] * Sk[2] + cw[3] * Sk[3];
      float ph = hw * (Sk[2] - Sk[0]) + lw * (Sk[3] - Sk[1]);       // reference .cuh:122-158: d/dh, d/dw of the bilinear value
      float pw = hh * (Sk[1] - Sk[0]) + lh * (Sk[3] - Sk[2]);
      const int cy0 = h_low - lv.wy0, cx0 = w_low - lv.wx0;
      // does this point leave the halo-5 window?  (HALF: measured against the inner window; otherwise the `slow` bits say it)
      if (HALF && in_range && !((unsigned)(cy0 - lv.dy5) < (unsigned)max(lv.wh5 - 1, 0) && (unsigned)(cx0 - lv.dx5) < (unsigned)max(lv.ww5 - 1, 0))) out5 |= 1u << p;
      auto add_corner = [&](int *h0, float sc) {
        if (HALF && (sub >> 1) != half) return;               // the other half's channels belong to the sibling workgroup
        const f32x2 sc2 = {sc, sc};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x2 f = __builtin_elementwise_fma(sc2, gq2[c], magic2);
          if (ABL & 1) { asm volatile("" ::"v"(h0), "v"(f.x), "v"(f.y)); }
          else { atomicAdd(h0 + 2 * c, __float_as_int(f.x)); atomicAdd(h0 + 2 * c + 1, __float_as_int(f.y)); }   // ds_add_u32, immediate offsets
        }
        if ((HALF ? (sub & 1) == 0 : sub == 0) && !(ABL & 1)) atomicAdd(h0 + CW - 1, 1);   // (h0 of that lane is the cell's first word)
      };
      if (fixed_ok && ok[0] && ok[3] && (unsigned)cy0 < (unsigned)max(lv.wh - 1, 0) && (unsigned)cx0 < (unsigned)max(lv.ww - 1, 0)) {
        // all four corners inside the map and inside the cached window (the common case): straight-line code
        int *h00 = win + (lv.wbase + __mul24(cy0, lv.ww) + cx0) * CW + (HALF ? (sub & 1) : sub) * 8;
        int *h10 = h00 + lv.ww * CW;
        add_corner(h00, cw[0] * a);
        add_corner(h00 + CW, cw[1] * a);
        add_corner(h10, cw[2] * a);
        add_corner(h10 + CW, cw[3] * a);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float sc = cw[k] * a;
          if (ok[k] && !(fixed_ok && sc == 0.f)) {       // a zero weight (sample exactly on a pixel centre) adds nothing
            const int cy = cy0 + (k >> 1), cx = cx0 + (k & 1);
            if (fixed_ok && (unsigned)cy < (unsigned)lv.wh && (unsigned)cx < (unsigned)lv.ww)
              add_corner(win + (lv.wbase + __mul24(cy, lv.ww) + cx) * CW + (HALF ? (sub & 1) : sub) * 8, sc);
            else if (!HALF || (sub >> 1) == half) slow |= 1u << p;   // a corner outside the cached window (large learned offset): handled below
          }
        }
      }
      pa = group4_sum(pa);
      if constexpr (FUSED) {
        // a g (the softmax backward subtracts a sum_j a_j g_j in the closing pass); d offset = (pw a W) / W = pw a
        pa = a * pa;
        pw = group4_sum(pw) * a;
        ph = group4_sum(ph) * a;
      } else {
        pw = group4_sum(pw) * (a * W);
        ph = group4_sum(ph) * (a * H);
      }
      if (sub == p) { keep_a = pa; keep_x = pw; keep_y = ph; }
      __builtin_amdgcn_sched_barrier(0);          // one point's 8 rows in flight at a time: the next point's would not fit 128 VGPRs
    }
    if (sub == 0 && half == 0) nmiss += __popc(HALF ? out5 : slow);
    if (!HALF || half == 0) {
      if constexpr (FUSED) {
        const int bq = b * S + q;
        float *drow = f.d_oa + (int64_t)bq * f.ld_doa;
        drow[M * (LP * 2) + mv * LP + l * P + sub] = keep_a;
        reinterpret_cast<float2 *>(drow + mv * (LP * 2) + l * (P * 2))[sub] = make_float2(keep_x, keep_y);
        if (!(ABL & 64)) {
          const float part = group4_sum(keep_a);                     // this level's share of sum_j a_j g_j
          if (sub == 0) f.gdot[qm * L + l] = part;
        }
      } else {
        grad_attn[qm * LP + l * P + sub] = keep_a;
        reinterpret_cast<float2 *>(grad_loc + (qm * LP + l * P) * 2)[sub] = make_float2(keep_x, keep_y);
      }
    }
    if (slow) {
      // corners no window caches: direct global atomics (the result never depends on window / halo sizes).  Rolled, re-reading
      // the point from memory, so that this rare path costs the loop above neither registers nor code
#pragma unroll 1
      for (int p = 0; p < P; ++p) {
        if (!((slow >> p) & 1)) continue;
        float2 xy;
        float a;
        if constexpr (FUSED) {                                 // (rare path: the point re-formed from the raw projection row)
          const int bq = b * S + q;
          const float *row = loc + (int64_t)bq * f.ld_oa;
          const float2 of = *reinterpret_cast<const float2 *>(row + m * (LP * 2) + (l * P + p) * 2);
          const float2 rf = *reinterpret_cast<const float2 *>(f.ref + ((int64_t)bq * L + l) * 2);
          const float2 st = f.stats[qm];
          a = __expf(row[M * (LP * 2) + m * LP + l * P + p] - st.x) * st.y;
          xy = make_float2(rf.x + of.x / (float)W, rf.y + of.y / (float)H);
        } else {
          xy = *reinterpret_cast<const float2 *>(loc + (qm * LP + l * P + p) * 2);
          a = attn[qm * LP + l * P + p];
        }
        const float h_im = xy.y * H - 0.5f, w_im = xy.x * W - 0.5f;
        const bool in_range = h_im > -1 && w_im > -1 && h_im < H && w_im < W;
        int off[4]; bool ok[4]; float cw[4], lh, lw, hh, hw;
        corner_setup<float>(h_im, w_im, H, W, stride_w, in_range, off, ok, cw, lh, lw, hh, hw);
        const int cy0 = (in_range ? (int)floorf(h_im) : 0) - lv.wy0, cx0 = (in_range ? (int)floorf(w_im) : 0) - lv.wx0;
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
          const int cy = cy0 + (k >> 1), cx = cx0 + (k & 1);
          const bool cached = fixed_ok && (unsigned)cy < (unsigned)lv.wh && (unsigned)cx < (unsigned)lv.ww;
          const int o = k == 0 ? off[0] : k == 1 ? off[1] : k == 2 ? off[2] : off[3];
          const float w_ = k == 0 ? cw[0] : k == 1 ? cw[1] : k == 2 ? cw[2] : cw[3];
          const bool okk = k == 0 ? ok[0] : k == 1 ? ok[1] : k == 2 ? ok[2] : ok[3];
          const float sc = w_ * a;
          if (!okk || cached || (fixed_ok && sc == 0.f)) continue;
          float *g = gbase + o;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            if (ABL & 2) asm volatile("" ::"v"(g), "v"(sc * gs[c]));
            else unsafeAtomicAdd(g + c, sc * gs[c]);
          }
        }
      }
    }
  }
  if (cnt) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nmiss += __shfl_xor(nmiss, o, 64);
    if ((threadIdx.x & 63) == 0 && nmiss) atomicAdd(&s_miss, (int)nmiss);
  }
  __syncthreads();
  if constexpr (FUSED && !(ABL & 32)) {
    // closing pass of the softmax backward: d logit_j = a_j g_j - a_j sum_i a_i g_i.  The main pass left a_j g_j in d_oa and every
    // (query, level) unit's partial sum in scratch (plain stores of this workgroup, ordered by the barrier above; summed here in level
    // order: deterministic); a thread takes one (query, head): 12 probabilities re-formed from the logits, 48 bytes rewritten, and the
    // absolute maximum of the head's 36 gradient values joins the row maximum with ONE atomic.
    if (!HALF || half == 0) {
      for (int i = threadIdx.x; i < nq_all; i += NT) {
        const int q = query_of(i);
        const int64_t bq = (int64_t)b * S + q, qm = bq * M + m;
        const float sum = (f.gdot[qm * L] + f.gdot[qm * L + 1]) + f.gdot[qm * L + 2];
        const float2 st = f.stats[qm];
        const float4 *lg4 = reinterpret_cast<const float4 *>(loc + bq * f.ld_oa + M * (LP * 2) + m * LP);
        float4 *dl4 = reinterpret_cast<float4 *>(f.d_oa + bq * f.ld_doa + M * (LP * 2) + m * LP);
        const float4 *do4 = reinterpret_cast<const float4 *>(f.d_oa + bq * f.ld_doa + m * (LP * 2));
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < L; ++j) {
          const float4 lg = lg4[j];
          float4 t = dl4[j];
          t.x -= __expf(lg.x - st.x) * st.y * sum; t.y -= __expf(lg.y - st.x) * st.y * sum;
          t.z -= __expf(lg.z - st.x) * st.y * sum; t.w -= __expf(lg.w - st.x) * st.y * sum;
          dl4[j] = t;
          mx = fmaxf(mx, fmaxf(fmaxf(fabsf(t.x), fabsf(t.y)), fmaxf(fabsf(t.z), fabsf(t.w))));
        }
        if (f.doa_amax && !(ABL & 8)) {
#pragma unroll
          for (int j = 0; j < 2 * L; ++j) {
            const float4 o = do4[j];
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
          }
          if (mx > 0.f) atomicMax(f.doa_amax + bq, __float_as_uint(mx));                  // non-negative floats order like their bits
        }
      }
    }
  }
  // ---- flush: lane = channel, one full 128-byte line per cell per half-wave
#pragma unroll 1
  for (int l = 0; l < L; ++l) {
    const LvlP lv = lvp[l];
    float *gl = grad_value + ((int64_t)b * S + lv.ls) * stride_w + m * 32;
    const int cells = lv.wh * lv.ww;
    constexpr int NCH = HALF ? 16 : 32;                     // channels a cell holds; lane = channel
    for (int i = threadIdx.x; i < cells * NCH; i += NT) {
      const int cellw = i / NCH, cl = i % NCH, ch = half * 16 + cl;
      const int *cp = win + (lv.wbase + cellw) * CW;
      const int acc = (int)((unsigned)cp[cl] - (unsigned)cp[CW - 1] * 0x4B400000u);
      const float v = (float)acc / chscale[ch];           // chscale is a power of two: exact
      if (acc != 0 && !(ABL & 4)) unsafeAtomicAdd(gl + ((int64_t)(lv.wy0 + cellw / lv.ww) * lv.W + lv.wx0 + cellw % lv.ww) * stride_w + ch, v);
    }
  }
  // the launch's miss statistics (the gate of LATER launches) leave last: the ticket's agent-scope fence then has nothing of this
  // workgroup's closing pass / flush to wait out or to write back twice
  if (cnt && threadIdx.x == 0) {
    // relaxed agent-scope atomics, NO fence between the counts and the ticket: the three words share one 16-byte slot (one memory
    // channel; a wavefront's atomics to it are performed in issue order), and what they feed is a speed heuristic — a count that
    // missed a straggler would only shade a later launch's choice of window size.  A __threadfence() here made every workgroup wait
    // for the write-back of the lines it had just written, with its CU idle (one workgroup per CU): 6 us per launch in the operator
    // kernel, 19 us in the fused one.
    if (half == 0) {
      (void)__hip_atomic_fetch_add(cnt, (unsigned)s_miss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      (void)__hip_atomic_fetch_add(cnt + 1, (unsigned)(nq_all * LP), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (__hip_atomic_fetch_add(cnt + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nact - 1u) {   // the last active workgroup: publish and recycle the slot
      const unsigned mi = __hip_atomic_fetch_add(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned to = __hip_atomic_fetch_add(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (pub) { *pub = ((unsigned long long)mi << 32) | to; __threadfence_system(); }
      atomicExch(cnt, 0u); atomicExch(cnt + 1, 0u); atomicExch(cnt + 2, 0u);
    }
  }
}

// ----------------------------------------------------------------------------------------- generic forward
template <typename T>
__global__ __launch_bounds__(256) void msda_fwd_generic(const T *__restrict__ value, const int64_t *__restrict__ shapes,
                                                         const int64_t *__restrict__ lvl_start, const T *__restrict__ loc,
                                                         const T *__restrict__ attn, T *__restrict__ out,
                                                         int64_t n, int S, int M, int D, int L, int Lq, int P)
{
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % D);
    const int64_t qm = idx / D;
    const int m = (int)(qm % M);
    const int b = (int)((qm / M) / Lq);
    const int stride_w = M * D;
    int64_t wp = qm * L * P;
    T col = 0;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const T *vbase = value + ((int64_t)b * S + lvl_start[l]) * stride_w + m * D + c;
      for (int p = 0; p < P; ++p, ++wp) {
        const T x = loc[2 * wp], y = loc[2 * wp + 1], a = attn[wp];
        const T h_im = y * H - (T)0.5, w_im = x * W - (T)0.5;
        const bool in_range = h_im > -1 && w_im > -1 && h_im < H && w_im < W;
        int off[4]; bool ok[4]; T cw[4], lh, lw, hh, hw;
        corner_setup<T>(h_im, w_im, H, W, stride_w, in_range, off, ok, cw, lh, lw, hh, hw);
        T val = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) val += ok[k] ? cw[k] * vbase[off[k]] : (T)0;
        col += val * a;
      }
    }
    out[idx] = col;
  }
}

// ----------------------------------------------------------------------------------------- generic backward
template <typename T>
__device__ __forceinline__ T wave_sum(T x)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

// one wavefront per (b,q,m); lanes stride over channels
template <typename T>
__global__ __launch_bounds__(256) void msda_bwd_generic(const T *__restrict__ value, const int64_t *__restrict__ shapes,
                                                         const int64_t *__restrict__ lvl_start, const T *__restrict__ loc,
                                                         const T *__restrict__ attn, const T *__restrict__ grad_out,
                                                         T *__restrict__ grad_value, T *__restrict__ grad_loc,
                                                         T *__restrict__ grad_attn, int64_t total_qm, int S, int M, int D,
                                                         int L, int Lq, int P)
{
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t qm = wave0; qm < total_qm; qm += nwaves) {
    const int m = (int)(qm % M);
    const int b = (int)((qm / M) / Lq);
    const int stride_w = M * D;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const int64_t voff = ((int64_t)b * S + lvl_start[l]) * stride_w + m * D;
      for (int p = 0; p < P; ++p) {
        const int64_t wp = (qm * L + l) * P + p;
        const T x = loc[2 * wp], y = loc[2 * wp + 1], a = attn[wp];
        const T h_im = y * H - (T)0.5, w_im = x * W - (T)0.5;
        const bool in_range = h_im > -1 && w_im > -1 && h_im < H && w_im < W;
        int off[4]; bool ok[4]; T cw[4], lh, lw, hh, hw;
        corner_setup<T>(h_im, w_im, H, W, stride_w, in_range, off, ok, cw, lh, lw, hh, hw);
        T pa = 0, ph = 0, pw = 0;
        for (int c = lane; c < D; c += 64) {
          const T top = grad_out[qm * D + c];
          T v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = ok[k] ? value[voff + off[k] + c] : (T)0;
          const T gh = -hw * v[0] - lw * v[1] + hw * v[2] + lw * v[3];
          const T gw = -hh * v[0] + hh * v[1] - lh * v[2] + lh * v[3];
          const T val = cw[0] * v[0] + cw[1] * v[1] + cw[2] * v[2] + cw[3] * v[3];
          pa += top * val; ph += top * gh; pw += top * gw;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (ok[k]) unsafeAtomicAdd(grad_value + voff + off[k] + c, cw[k] * a * top);
        }
        pa = wave_sum(pa); ph = wave_sum(ph); pw = wave_sum(pw);
        if (lane == 0) {
          grad_attn[wp] = pa;
          grad_loc[2 * wp] = pw * a * W;
          grad_loc[2 * wp + 1] = ph * a * H;
        }
      }
    }
  }
}

inline int round_up8(int64_t x) { return (int)((x + 7) / 8 * 8); }

int check_common(const void *const *ptrs, int nptr, int batch, int spatial_size, int num_heads, int channels,
                 int num_levels, int num_query, int num_point, int im2col_step, int dtype)
{
  for (int i = 0; i < nptr; ++i)
    if (!ptrs[i]) return pd_set_error(PD_ERR_INVALID_ARG, "msda: null pointer argument (index %d)", i);
  if (batch < 0 || spatial_size < 0 || num_heads <= 0 || channels <= 0 || num_levels <= 0 || num_query < 0 || num_point <= 0)
    return pd_set_error(PD_ERR_INVALID_ARG, "msda: bad sizes batch=%d S=%d M=%d D=%d L=%d Lq=%d P=%d", batch, spatial_size,
                        num_heads, channels, num_levels, num_query, num_point);
  if (dtype != PD_F32 && dtype != PD_F64)
    return pd_set_error(PD_ERR_INVALID_ARG, "msda: dtype %d not supported (float32/float64 only, as the reference)", dtype);
  if (im2col_step <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "msda: im2col_step must be positive, got %d", im2col_step);
  if (batch > 0) {
    const int step = batch < im2col_step ? batch : im2col_step;
    if (batch % step != 0)
      return pd_set_error(PD_ERR_IM2COL_STEP, "batch(%d) must divide im2col_step(%d)", batch, step);
  }
  return PD_OK;
}

}  // namespace

static int msda_forward_launch(const void *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                               const void *sampling_loc, const void *attn_weight, void *output, int batch,
                               int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                               int num_point, int im2col_step, int dtype, void *stream_, float *row_amax)
{
  const void *ptrs[] = {value, spatial_shapes, level_start_index, sampling_loc, attn_weight, output};
  int rc = check_common(ptrs, 6, batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, im2col_step, dtype);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t total_qm = (int64_t)batch * num_query * num_heads;
  if (total_qm == 0) return PD_OK;
  // (the reference splits the batch into im2col_step chunks only to bound its launch size; one launch here)
  if (dtype == PD_F32 && channels == 32 && num_levels == 3 && num_point == 4 && total_qm < (1LL << 31)) {
    if (g_msda_fwd_q4) {                                  // four lanes per (query, head), a point's geometry once per quad (round 6)
      const int nblocks = round_up8((total_qm + 63) / 64);
      hipLaunchKernelGGL((msda_fwd_q4<3, 4, false>), dim3(nblocks), dim3(256), 0, stream, (const float *)value, spatial_shapes,
                         level_start_index, (const float *)sampling_loc, (const float *)attn_weight, (float *)output,
                         spatial_size, num_heads, num_query, (int)total_qm, reinterpret_cast<unsigned *>(row_amax));
      return pd_check_launch("pd_msda_forward");
    }
    const int nblocks = round_up8((total_qm + 31) / 32);
    hipLaunchKernelGGL((msda_fwd_d32<3, 4>), dim3(nblocks), dim3(256), 0, stream, (const float *)value, spatial_shapes,
                       level_start_index, (const float *)sampling_loc, (const float *)attn_weight, (float *)output,
                       spatial_size, num_heads, num_query, (int)total_qm, reinterpret_cast<unsigned *>(row_amax));
  } else {
    if (row_amax) return pd_set_error(PD_ERR_INVALID_ARG, "pd_msda_forward_amax: row maxima come from the fp32, D = 32, L = 3, P = 4 kernel only");
    const int64_t n = total_qm * channels;
    const int nblocks = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    if (dtype == PD_F32)
      hipLaunchKernelGGL(msda_fwd_generic<float>, dim3(nblocks), dim3(256), 0, stream, (const float *)value, spatial_shapes,
                         level_start_index, (const float *)sampling_loc, (const float *)attn_weight, (float *)output, n,
                         spatial_size, num_heads, channels, num_levels, num_query, num_point);
    else
      hipLaunchKernelGGL(msda_fwd_generic<double>, dim3(nblocks), dim3(256), 0, stream, (const double *)value, spatial_shapes,
                         level_start_index, (const double *)sampling_loc, (const double *)attn_weight, (double *)output, n,
                         spatial_size, num_heads, channels, num_levels, num_query, num_point);
  }
  return pd_check_launch("pd_msda_forward");
}

extern "C" int pd_msda_forward(const void *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                               const void *sampling_loc, const void *attn_weight, void *output, int batch,
                               int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                               int num_point, int im2col_step, int dtype, void *stream_)
{
  return msda_forward_launch(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, output, batch, spatial_size, num_heads, channels,
                             num_levels, num_query, num_point, im2col_step, dtype, stream_, nullptr);
}

extern "C" int pd_msda_forward_amax(const void *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                                    const void *sampling_loc, const void *attn_weight, void *output, float *row_amax, int batch,
                                    int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                    int num_point, int im2col_step, int dtype, void *stream_)
{
  return msda_forward_launch(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, output, batch, spatial_size, num_heads, channels,
                             num_levels, num_query, num_point, im2col_step, dtype, stream_, row_amax);
}

// Which LDS-window backward kernel a launch uses is decided on the HOST from what EARLIER launches measured: every gated launch
// counts the sample points that left the halo-5 windows and its last workgroup writes {missed, looked-at} into host-mapped
// memory (a ring of 64 slots per device; nothing waits for it).  A launch takes the halo-9 kernel when the most recent results that
// have ARRIVED (the host runs ahead of the device) show more than g_pd_dbg_msda_gate_pct per mille misses.  Both kernels are correct for any offsets — a stale
// or missing measurement (the first launches of a run) only costs speed.
#include <mutex>
static std::mutex g_gate_mu;
static unsigned *g_gate_cnt[16] = {nullptr};                 // device: 64 x {missed, looked-at, ticket, pad}
static unsigned long long *g_gate_pub[16] = {nullptr};       // host-mapped: 64 x (missed << 32 | looked-at)
static unsigned long long *g_gate_pub_dev[16] = {nullptr};
static unsigned g_gate_next[16] = {0};
static int g_gate_last_variant[16] = {0};
int g_pd_dbg_msda_gate_pct = 20;     // halo-9 kernel when more than this many PER MILLE of the sample points of recent launches left the halo-5 windows
                                     // (measured break-even: 0.9 % misses -> 0.28 vs 0.39 ms, 3.9 % -> 0.63 vs 0.40 ms)

static bool msda_gate_init(int dev, hipStream_t stream)
{
  if (g_gate_cnt[dev]) return true;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(stream, &cs);
  if (cs != hipStreamCaptureStatusNone) return false;        // no allocation inside a capture
  void *c = nullptr, *h = nullptr, *hd = nullptr;
  if (hipMalloc(&c, 64 * 16) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (hipMemset(c, 0, 64 * 16) != hipSuccess || hipHostMalloc(&h, 64 * 8, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer(&hd, h, 0) != hipSuccess) {
    (void)hipGetLastError(); (void)hipFree(c);
    if (h) (void)hipHostFree(h);
    return false;
  }
  memset(h, 0, 64 * 8);
  g_gate_cnt[dev] = (unsigned *)c; g_gate_pub[dev] = (unsigned long long *)h; g_gate_pub_dev[dev] = (unsigned long long *)hd;
  return true;
}

// -> slot index of this launch (or -1: ungated) and whether recent launches call for the halo-9 kernel
static int msda_gate_pick(hipStream_t stream, bool *want_half)
{
  int dev = 0;
  *want_half = false;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -1;
  std::lock_guard<std::mutex> lk(g_gate_mu);
  if (!msda_gate_init(dev, stream)) return -1;
  const unsigned n = g_gate_next[dev]++;
  // the host runs ahead of the device: the newest results are usually still in flight (slot == 0).  Take the 6 most recent that
  // HAVE arrived among the last 63 launches; with none (start of a run, or > 63 launches queued) keep the previous decision.
  // Hysteresis: up at > gate per-mille misses in any of them, back down only when all are below 0.7 of it.
  int seen = 0;
  bool any_high = false, all_low = true;
  for (unsigned k = 1; k <= 63 && k <= n && seen < 6; ++k) {
    const unsigned long long v = ((volatile unsigned long long *)g_gate_pub[dev])[(n - k) & 63];
    const unsigned long long mi = v >> 32, to = v & 0xffffffffull;
    if (!to) continue;
    ++seen;
    if (mi * 1000ull > to * (unsigned long long)g_pd_dbg_msda_gate_pct) any_high = true;
    if (mi * 10000ull > to * (unsigned long long)g_pd_dbg_msda_gate_pct * 7ull) all_low = false;
  }
  static bool state[16] = {false};
  if (seen) {
    if (any_high) state[dev] = true;
    else if (all_low) state[dev] = false;
  }
  *want_half = state[dev];
  g_gate_last_variant[dev] = *want_half ? 2 : 3;
  return (int)(n & 63);
}

// tools / bench: {missed, looked-at} sample points of the most recent gated backward launch whose result has arrived on the current
// device, and the variant (2 = halo 9, 3 = halo 5) the most recent launch took.  Reads host memory only.
extern "C" int pd_msda_backward_last_gate(unsigned *out3)
{
  int dev = 0;
  if (!out3 || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return pd_set_error(PD_ERR_INVALID_ARG, "pd_msda_backward_last_gate: bad argument");
  out3[0] = out3[1] = out3[2] = 0;
  std::lock_guard<std::mutex> lk(g_gate_mu);
  if (!g_gate_pub[dev]) return PD_OK;
  out3[2] = (unsigned)g_gate_last_variant[dev];
  const unsigned n = g_gate_next[dev];
  for (unsigned k = 1; k <= 64 && k <= n; ++k) {
    const unsigned long long v = ((volatile unsigned long long *)g_gate_pub[dev])[(n - k) & 63];
    if (v & 0xffffffffull) { out3[0] = (unsigned)(v >> 32); out3[1] = (unsigned)(v & 0xffffffffull); break; }
  }
  return PD_OK;
}

extern "C" int pd_msda_backward(const void *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                                const void *sampling_loc, const void *attn_weight, const void *grad_output,
                                void *grad_value, void *grad_sampling_loc, void *grad_attn_weight, int batch,
                                int spatial_size, int num_heads, int channels, int num_levels, int num_query,
                                int num_point, int im2col_step, int dtype, void *stream_)
{
  const void *ptrs[] = {value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                        grad_value, grad_sampling_loc, grad_attn_weight};
  int rc = check_common(ptrs, 9, batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, im2col_step, dtype);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const size_t esz = dtype == PD_F32 ? 4 : 8;
  const int64_t total_qm = (int64_t)batch * num_query * num_heads;
  const size_t nv = (size_t)batch * spatial_size * num_heads * channels;
  const size_t nw = (size_t)total_qm * num_levels * num_point;
  if (nv) (void)hipMemsetAsync(grad_value, 0, nv * esz, stream);
  if (total_qm == 0) return pd_check_launch("pd_msda_backward");
  if (!g_pd_dbg_force_generic && g_pd_dbg_atomic_scope == 0 && dtype == PD_F32 && channels == 32 && num_point == 4 &&
      num_levels <= 8 && num_query == spatial_size && total_qm < (1LL << 31)) {
    // self-attention geometry: LDS-windowed accumulation (every grad_loc / grad_attn slot is written once)
    if (num_levels == 3 && g_pd_dbg_bwd_variant != 1 && spatial_size < (1 << 23) && num_heads * 32 < (1 << 23)) {   // 24-bit index multiplies
      const size_t lds4 = ((size_t)kOwnCells4 * 33 + 64) * sizeof(int), lds4h = ((size_t)kOwnCells4H * 17 + 64) * sizeof(int);
      typedef void (*ofn)(const float *, const int64_t *, const int64_t *, const float *, const float *, const float *, float *, float *,
                          float *, int, int, int, unsigned *, unsigned long long *, FusedBwd);
#ifdef PD_PROBES                                                     // ablation instantiations: the probe library only (make probes)
      const int ai = g_pd_dbg_ablate == 1 ? 1 : g_pd_dbg_ablate == 4 ? 2 : g_pd_dbg_ablate == 7 ? 3 : 0;
      const ofn all4[4] = {msda_bwd_owner4_d32<0>, msda_bwd_owner4_d32<1>, msda_bwd_owner4_d32<4>, msda_bwd_owner4_d32<7>};
#else
      const int ai = 0;
      const ofn all4[1] = {msda_bwd_owner4_d32<0>};
#endif
      const ofn half4 = msda_bwd_owner4_d32<0, true>;
      static bool a4set[5] = {false, false, false, false, false};
      if (!a4set[ai]) { (void)hipFuncSetAttribute((const void *)all4[ai], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4); a4set[ai] = true; }
      if (!a4set[4]) { (void)hipFuncSetAttribute((const void *)half4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4h); a4set[4] = true; }
      const int64_t nb4 = (int64_t)batch * kGmax * kGmax * num_heads;
      // g_pd_dbg_bwd_variant (tools): 0 = picked from what recent launches measured, 2 = always halo 9, 3 = always halo 5
      bool want_half = g_pd_dbg_bwd_variant == 2;
      unsigned *cnt = nullptr;
      unsigned long long *pub = nullptr;
      if (g_pd_dbg_bwd_variant == 0 && g_pd_dbg_ablate == 0) {
        const int slot = msda_gate_pick(stream, &want_half);
        if (slot >= 0) {
          int dev = 0;
          (void)hipGetDevice(&dev);
          cnt = g_gate_cnt[dev] + 4 * slot; pub = g_gate_pub_dev[dev] + slot;
          ((volatile unsigned long long *)g_gate_pub[dev])[slot] = 0ull;      // "not measured yet" until this launch publishes
        }
      }
      if (!want_half)
        hipLaunchKernelGGL(all4[ai], dim3((unsigned)nb4), dim3(1024), lds4, stream, (const float *)value, spatial_shapes, level_start_index,
                           (const float *)sampling_loc, (const float *)attn_weight, (const float *)grad_output, (float *)grad_value,
                           (float *)grad_sampling_loc, (float *)grad_attn_weight, spatial_size, num_heads, batch, cnt, pub, FusedBwd());
      else
        hipLaunchKernelGGL(half4, dim3((unsigned)(2 * nb4)), dim3(1024), lds4h, stream, (const float *)value, spatial_shapes, level_start_index,
                           (const float *)sampling_loc, (const float *)attn_weight, (const float *)grad_output, (float *)grad_value,
                           (float *)grad_sampling_loc, (float *)grad_attn_weight, spatial_size, num_heads, batch, cnt, pub, FusedBwd());
      return pd_check_launch("pd_msda_backward");
    }
    const int64_t nblocks = (int64_t)batch * kGmax * kGmax * num_levels * num_heads;
    const size_t lds = ((size_t)kWin * kWin * 32 + 32) * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void *)msda_bwd_tiled_d32<4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#ifdef PD_PROBES
      (void)hipFuncSetAttribute((const void *)msda_bwd_tiled_d32<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void *)msda_bwd_tiled_d32<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void *)msda_bwd_tiled_d32<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void *)msda_bwd_tiled_d32<4, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#endif
      attr_set = true;
    }
#ifdef PD_PROBES
    auto tk = g_pd_dbg_ablate == 1 ? msda_bwd_tiled_d32<4, 1> : g_pd_dbg_ablate == 2 ? msda_bwd_tiled_d32<4, 2>
            : g_pd_dbg_ablate == 4 ? msda_bwd_tiled_d32<4, 4> : g_pd_dbg_ablate == 7 ? msda_bwd_tiled_d32<4, 7> : msda_bwd_tiled_d32<4, 0>;
#else
    auto tk = msda_bwd_tiled_d32<4, 0>;
#endif
    // 74 KB of LDS window -> 2 workgroups per CU whatever their size: 512 threads give the gathers 16 waves per CU to hide
    // their latency behind instead of 8 (measured 0.56 -> see DESIGN.md)
    const int nthreads = g_pd_dbg_bwd_threads > 0 ? g_pd_dbg_bwd_threads : 512;
    hipLaunchKernelGGL(tk, dim3((unsigned)nblocks), dim3(nthreads), lds, stream, (const float *)value,
                       spatial_shapes, level_start_index, (const float *)sampling_loc, (const float *)attn_weight,
                       (const float *)grad_output, (float *)grad_value, (float *)grad_sampling_loc,
                       (float *)grad_attn_weight, spatial_size, num_heads, num_levels);
  } else if (!g_pd_dbg_force_generic && dtype == PD_F32 && channels == 32 && num_levels == 3 && num_point == 4 && total_qm < (1LL << 31)) {
    // every (b,q,m,l,p) slot of grad_loc / grad_attn is written by the kernel: no memset needed
    const int nblocks = round_up8((total_qm + 31) / 32);
    auto kern = g_pd_dbg_atomic_scope == 1 ? msda_bwd_d32<3, 4, 1> : g_pd_dbg_atomic_scope == 2 ? msda_bwd_d32<3, 4, 2> : msda_bwd_d32<3, 4, 0>;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), 0, stream, (const float *)value, spatial_shapes,
                       level_start_index, (const float *)sampling_loc, (const float *)attn_weight,
                       (const float *)grad_output, (float *)grad_value, (float *)grad_sampling_loc,
                       (float *)grad_attn_weight, spatial_size, num_heads, num_query, (int)total_qm);
  } else {
    (void)nw;
    const int64_t nb = (total_qm + 3) / 4;
    const int nblocks = (int)(nb < 65536 ? nb : 65536);
    if (dtype == PD_F32)
      hipLaunchKernelGGL(msda_bwd_generic<float>, dim3(nblocks), dim3(256), 0, stream, (const float *)value, spatial_shapes,
                         level_start_index, (const float *)sampling_loc, (const float *)attn_weight,
                         (const float *)grad_output, (float *)grad_value, (float *)grad_sampling_loc,
                         (float *)grad_attn_weight, total_qm, spatial_size, num_heads, channels, num_levels, num_query,
                         num_point);
    else
      hipLaunchKernelGGL(msda_bwd_generic<double>, dim3(nblocks), dim3(256), 0, stream, (const double *)value,
                         spatial_shapes, level_start_index, (const double *)sampling_loc, (const double *)attn_weight,
                         (const double *)grad_output, (double *)grad_value, (double *)grad_sampling_loc,
                         (double *)grad_attn_weight, total_qm, spatial_size, num_heads, channels, num_levels, num_query,
                         num_point);
  }
  return pd_check_launch("pd_msda_backward");
}

// ----------------------------------------------------------------------------------------- fused module path (round 5)
// MSDeformAttn.forward (ms_deform_attn.py:86-131) hands the operator softmax(attention_weights(query)) and reference_points +
// sampling_offsets(query) / (W, H).  pd_msda_fused_* take the projections' RAW output instead and do those two steps in the kernels'
// registers (see msda_fwd_d32<.., FUSED> / msda_bwd_owner4_d32<.., FUSED>).  pd_msda_forward / pd_msda_backward — the reference's
// operator ABI — are unchanged; the fused pair is what the module path of this package calls when pd_msda_fused_supported says so.
extern "C" int pd_msda_fused_supported(int batch, int spatial_size, int num_heads, int channels, int num_levels, int num_query, int num_point)
{
  const int64_t total_qm = (int64_t)batch * num_query * num_heads;
  return (!g_pd_dbg_force_generic && g_pd_dbg_atomic_scope == 0 && g_pd_dbg_bwd_variant != 1 && channels == 32 && num_point == 4 && num_levels == 3 &&
          num_query == spatial_size && total_qm > 0 && total_qm < (1LL << 31) && spatial_size < (1 << 23) && num_heads * 32 < (1 << 23)) ? 1 : 0;
}

extern "C" int pd_msda_fused_forward(const float *value, const int64_t *spatial_shapes, const int64_t *level_start_index, const float *oa,
                                     int ld_oa, const float *ref, float *output, float *stats, float *row_amax, int batch, int spatial_size,
                                     int num_heads, int channels, int num_levels, int num_query, int num_point, void *stream_)
{
  const void *ptrs[] = {value, spatial_shapes, level_start_index, oa, ref, output, stats};
  int rc = check_common(ptrs, 7, batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, 1, PD_F32);
  if (rc) return rc;
  if (!pd_msda_fused_supported(batch, spatial_size, num_heads, channels, num_levels, num_query, num_point))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_msda_fused_forward: geometry not served (ask pd_msda_fused_supported; fp32, 32 channels, 3 levels, 4 points, queries == pixels)");
  if (ld_oa < 3 * num_heads * num_levels * num_point || (ld_oa & 3) || ((uintptr_t)oa & 15))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_msda_fused_forward: projection rows need >= 3 M L P columns, a stride that is a multiple of 4 and a 16-byte aligned base");
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t total_qm = (int64_t)batch * num_query * num_heads;
  static const bool band_env = []() { const char *e = getenv("PD_MSDA_BAND"); return !e || e[0] != '0'; }();   // A/B switch
  if (g_msda_fwd_q4) {
    const int nb4 = round_up8((total_qm + 63) / 64);
    hipLaunchKernelGGL((msda_fwd_q4<3, 4, true>), dim3(nb4), dim3(256), 0, stream, value, spatial_shapes, level_start_index, oa, ref, output,
                       spatial_size, num_heads, num_query, (int)total_qm, reinterpret_cast<unsigned *>(row_amax), ld_oa, reinterpret_cast<float2 *>(stats),
                       (g_pd_dbg_ablate == 128 || !band_env) ? 0 : 1);
    return pd_check_launch("pd_msda_fused_forward");
  }
  const int nblocks = round_up8((total_qm + 31) / 32);
  hipLaunchKernelGGL((msda_fwd_d32<3, 4, true>), dim3(nblocks), dim3(256), 0, stream, value, spatial_shapes, level_start_index, oa, ref, output,
                     spatial_size, num_heads, num_query, (int)total_qm, reinterpret_cast<unsigned *>(row_amax), ld_oa, reinterpret_cast<float2 *>(stats),
                     (g_pd_dbg_ablate == 128 || !band_env) ? 0 : 1);
  return pd_check_launch("pd_msda_fused_forward");
}

extern "C" int pd_msda_fused_backward(const float *value, const int64_t *spatial_shapes, const int64_t *level_start_index, const float *oa,
                                      int ld_oa, const float *ref, const float *stats, const float *fwd_output, const float *grad_output,
                                      float *grad_value, float *d_oa, int ld_doa, float *d_oa_amax, float *scratch, int batch, int spatial_size, int num_heads,
                                      int channels, int num_levels, int num_query, int num_point, void *stream_)
{
  const void *ptrs[] = {value, spatial_shapes, level_start_index, oa, ref, stats, fwd_output, grad_output, grad_value, d_oa, scratch};
  int rc = check_common(ptrs, 11, batch, spatial_size, num_heads, channels, num_levels, num_query, num_point, 1, PD_F32);
  if (rc) return rc;
  if (!pd_msda_fused_supported(batch, spatial_size, num_heads, channels, num_levels, num_query, num_point))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_msda_fused_backward: geometry not served (ask pd_msda_fused_supported)");
  const int n3 = 3 * num_heads * num_levels * num_point;
  if (ld_oa < n3 || ld_doa < n3 || ((ld_oa | ld_doa) & 3) || (((uintptr_t)oa | (uintptr_t)d_oa) & 15))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_msda_fused_backward: projection rows need >= 3 M L P columns, strides that are multiples of 4 and 16-byte aligned bases");
  hipStream_t stream = (hipStream_t)stream_;
  const size_t nv = (size_t)batch * spatial_size * num_heads * channels;
  if (d_oa_amax == grad_value + nv) {              // the module path allocates them back to back: one clear
    (void)hipMemsetAsync(grad_value, 0, (nv + (size_t)batch * num_query) * sizeof(float), stream);
  } else {
    (void)hipMemsetAsync(grad_value, 0, nv * sizeof(float), stream);
    if (d_oa_amax) (void)hipMemsetAsync(d_oa_amax, 0, (size_t)batch * num_query * sizeof(float), stream);
  }
  const size_t lds4 = ((size_t)kOwnCells4 * 33 + 64) * sizeof(int), lds4h = ((size_t)kOwnCells4H * 17 + 64) * sizeof(int);
#ifdef PD_PROBES
  const auto k5 = g_pd_dbg_ablate == 8 ? msda_bwd_owner4_d32<8, false, true> : g_pd_dbg_ablate == 32 ? msda_bwd_owner4_d32<32, false, true>
                : g_pd_dbg_ablate == 96 ? msda_bwd_owner4_d32<96, false, true> : msda_bwd_owner4_d32<0, false, true>;
#else
  const auto k5 = msda_bwd_owner4_d32<0, false, true>;
#endif
  const auto k9 = msda_bwd_owner4_d32<0, true, true>;
  static bool attr = false;
  if (g_pd_dbg_ablate) (void)hipFuncSetAttribute((const void *)k5, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)k5, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
    (void)hipFuncSetAttribute((const void *)k9, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4h);
    attr = true;
  }
  const int64_t nb4 = (int64_t)batch * kGmax * kGmax * num_heads;
  bool want_half = g_pd_dbg_bwd_variant == 2;
  unsigned *cnt = nullptr;
  unsigned long long *pub = nullptr;
  if (g_pd_dbg_bwd_variant == 0) {                           // the same measured choice between the halo-5 and halo-9 windows as pd_msda_backward
    const int slot = msda_gate_pick(stream, &want_half);
    if (slot >= 0) {
      int dev = 0;
      (void)hipGetDevice(&dev);
      cnt = g_gate_cnt[dev] + 4 * slot; pub = g_gate_pub_dev[dev] + slot;
      ((volatile unsigned long long *)g_gate_pub[dev])[slot] = 0ull;
    }
  }
  FusedBwd f;
  f.ref = ref; f.stats = reinterpret_cast<const float2 *>(stats); f.fout = fwd_output; f.d_oa = d_oa;
  f.doa_amax = reinterpret_cast<unsigned *>(d_oa_amax); f.ld_oa = ld_oa; f.ld_doa = ld_doa; f.gdot = scratch;
  if (!want_half)
    hipLaunchKernelGGL(k5, dim3((unsigned)nb4), dim3(1024), lds4, stream, value, spatial_shapes, level_start_index, oa, (const float *)nullptr,
                       grad_output, grad_value, (float *)nullptr, (float *)nullptr, spatial_size, num_heads, batch, cnt, pub, f);
  else
    hipLaunchKernelGGL(k9, dim3((unsigned)(2 * nb4)), dim3(1024), lds4h, stream, value, spatial_shapes, level_start_index, oa, (const float *)nullptr,
                       grad_output, grad_value, (float *)nullptr, (float *)nullptr, spatial_size, num_heads, batch, cnt, pub, f);
  return pd_check_launch("pd_msda_fused_backward");
}
