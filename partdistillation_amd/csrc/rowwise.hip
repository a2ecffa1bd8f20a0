// Token-wise kernels of the transformer layers (gfx950, wave64): one wavefront owns one row of C = NV4*256 channels,
// every lane moves 16-byte (fp32) / 8-byte (bf16) pieces so a row is read and written as whole 1 KiB / 512 B bursts,
// and the row statistics are 64-lane DPP reductions.  C-ABI in include/pd_rowwise.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "pd_common.h"
#include "pd_msda.h"
#include "pd_rowwise.h"

int g_ln_bwd_cap = 0;     // pd_debug_set "ln_bwd_cap" (tools/ only): workgroups of pd_add_layernorm_bwd at most (0 = 256; < 0: four-wavefront workgroups whatever the rows)
namespace {

typedef unsigned short bf16_t;

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ bf16_t f2bf(float f)
{
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);   // NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                          // round to nearest even
  return (bf16_t)(u >> 16);
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t *p)
{
  const ushort4 v = *reinterpret_cast<const ushort4 *>(p);
  return make_float4(bf2f(v.x), bf2f(v.y), bf2f(v.z), bf2f(v.w));
}
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ void st4(bf16_t *p, float4 v)
{
  ushort4 o;
  o.x = f2bf(v.x); o.y = f2bf(v.y); o.z = f2bf(v.z); o.w = f2bf(v.w);
  *reinterpret_cast<ushort4 *>(p) = o;
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// sum over the 64 lanes, result in every lane: quad_perm swaps, row_half_mirror, row_mirror inside each row of 16
// lanes (DPP, no LDS traffic), then two cross-row exchanges
__device__ __forceinline__ float wave_sum(float v)
{
  v += dpp_mov<0xB1>(v);       // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);      // row_half_mirror
  v += dpp_mov<0x140>(v);      // row_mirror
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v)
{
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
__device__ __forceinline__ float amax4(float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }

// ------------------------------------------------------------------------------------------------ add + LayerNorm
template <int NV4, typename XT, typename CT>
__global__ __launch_bounds__(256) void add_ln_fwd(const XT *__restrict__ x, const float *__restrict__ res,
                                                  const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                  float *__restrict__ z, float *__restrict__ y, CT *__restrict__ y_c,
                                                  const float *__restrict__ pos, int pos_div, CT *__restrict__ ypos_c,
                                                  float *__restrict__ mean, float *__restrict__ rstd, int rows,
                                                  float *__restrict__ y_amax, float *__restrict__ ypos_amax)
{
  constexpr int C = NV4 * 256;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4 gm[NV4], bt[NV4];
#pragma unroll
  for (int j = 0; j < NV4; ++j) {
    gm[j] = ld4(gamma + j * 256 + lane * 4);
    bt[j] = ld4(beta + j * 256 + lane * 4);
  }
  for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
    const int64_t base = (int64_t)r * C + lane * 4;
    float4 v[NV4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV4; ++j) {
      v[j] = x ? ld4(x + base + j * 256) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (res) v[j] = add4(v[j], ld4(res + base + j * 256));
      if (z) st4(z + base + j * 256, v[j]);
      s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    const float mu = wave_sum(s) * (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV4; ++j) {
      v[j].x -= mu; v[j].y -= mu; v[j].z -= mu; v[j].w -= mu;
      q += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
    }
    const float rs = rsqrtf(wave_sum(q) * (1.f / C) + eps);
    if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
    const float *prow = pos ? pos + (int64_t)(r / pos_div) * C + lane * 4 : nullptr;
    float my = 0.f, mp = 0.f;                                      // absolute row maxima of y and y + pos (the scaling input of pd_gemm_tn_f16x2)
#pragma unroll
    for (int j = 0; j < NV4; ++j) {
      float4 o;
      o.x = v[j].x * rs * gm[j].x + bt[j].x;
      o.y = v[j].y * rs * gm[j].y + bt[j].y;
      o.z = v[j].z * rs * gm[j].z + bt[j].z;
      o.w = v[j].w * rs * gm[j].w + bt[j].w;
      if (y) st4(y + base + j * 256, o);
      if (y_c) st4(y_c + base + j * 256, o);
      my = fmaxf(my, amax4(o));
      if (ypos_c) {
        const float4 op = add4(o, ld4(prow + j * 256));
        st4(ypos_c + base + j * 256, op);
        mp = fmaxf(mp, amax4(op));
      }
    }
    if (y_amax) { my = wave_max(my); if (lane == 0) y_amax[r] = my; }
    if (ypos_amax) { mp = wave_max(mp); if (lane == 0) ypos_amax[r] = mp; }
  }
}

// WAVES wavefronts per workgroup, each walking rows; their column sums meet in LDS and leave as 3 C atomic adds per workgroup.  Those
// adds are what bounds the launch when workgroups are many: the chip retires ~40 G fp32 atomics per second, 1 024 workgroups x 768 adds
// took 21 of the 42 us at [43 008, 256] (tools/bench_add_ln.py; spreading them over 16 copies of the accumulators changed nothing: it is
// the total, not the depth per address).  Hence FAT workgroups: 16 wavefronts (C = 256), a grid of a few hundred.
template <int NV4, typename CT, typename DT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void add_ln_bwd(const float *dy, const float *__restrict__ dy2,
                                                  const CT *__restrict__ dy_c, const CT *__restrict__ dypos_c,
                                                  const float *__restrict__ z, const float *__restrict__ mean,
                                                  const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                  float *dz, DT *__restrict__ dz_c, float *__restrict__ dgamma,
                                                  float *__restrict__ dbeta, float *__restrict__ dbias,
                                                  float *__restrict__ dpos_acc, int pos_div, int rows, float *__restrict__ dz_amax)
{
  constexpr int C = NV4 * 256;
  __shared__ float red[WAVES][3][C];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4 gm[NV4], ag[NV4], ab[NV4], ad[NV4];
#pragma unroll
  for (int j = 0; j < NV4; ++j) {
    gm[j] = ld4(gamma + j * 256 + lane * 4);
    ag[j] = ab[j] = ad[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int r = blockIdx.x * WAVES + wave; r < rows; r += gridDim.x * WAVES) {
    const int64_t base = (int64_t)r * C + lane * 4;
    const float mu = mean[r], rs = rstd[r];
    float4 g[NV4], xh[NV4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV4; ++j) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (dy) t = ld4(dy + base + j * 256);
      if (dy2) t = add4(t, ld4(dy2 + base + j * 256));
      if (dy_c) t = add4(t, ld4(dy_c + base + j * 256));
      if (dypos_c) {
        const float4 p = ld4(dypos_c + base + j * 256);
        t = add4(t, p);
        if (dpos_acc) {
          float *a = dpos_acc + (int64_t)(r / pos_div) * C + j * 256 + lane * 4;
          if (pos_div == 1) st4(a, add4(ld4(a), p));      // one row per positional row: plain read-modify-write
          else { atomicAdd(a, p.x); atomicAdd(a + 1, p.y); atomicAdd(a + 2, p.z); atomicAdd(a + 3, p.w); }
        }
      }
      float4 h = ld4(z + base + j * 256);
      h.x = (h.x - mu) * rs; h.y = (h.y - mu) * rs; h.z = (h.z - mu) * rs; h.w = (h.w - mu) * rs;
      ag[j].x += t.x * h.x; ag[j].y += t.y * h.y; ag[j].z += t.z * h.z; ag[j].w += t.w * h.w;
      ab[j] = add4(ab[j], t);
      t.x *= gm[j].x; t.y *= gm[j].y; t.z *= gm[j].z; t.w *= gm[j].w;
      s1 += (t.x + t.y) + (t.z + t.w);
      s2 += (t.x * h.x + t.y * h.y) + (t.z * h.z + t.w * h.w);
      g[j] = t; xh[j] = h;
    }
    const float m1 = wave_sum(s1) * (1.f / C), m2 = wave_sum(s2) * (1.f / C);
    float mz = 0.f;
#pragma unroll
    for (int j = 0; j < NV4; ++j) {
      float4 o;
      o.x = rs * (g[j].x - m1 - xh[j].x * m2);
      o.y = rs * (g[j].y - m1 - xh[j].y * m2);
      o.z = rs * (g[j].z - m1 - xh[j].z * m2);
      o.w = rs * (g[j].w - m1 - xh[j].w * m2);
      ad[j] = add4(ad[j], o);
      st4(dz + base + j * 256, o);
      if (dz_c) st4(dz_c + base + j * 256, o);
      mz = fmaxf(mz, amax4(o));
    }
    if (dz_amax) { mz = wave_max(mz); if (lane == 0) dz_amax[r] = mz; }
  }
  if (!dgamma && !dbeta && !dbias) return;
#pragma unroll
  for (int j = 0; j < NV4; ++j) {
    st4(&red[wave][0][j * 256 + lane * 4], ag[j]);
    st4(&red[wave][1][j * 256 + lane * 4], ab[j]);
    st4(&red[wave][2][j * 256 + lane * 4], ad[j]);
  }
  __syncthreads();
  float *const outs[3] = {dgamma, dbeta, dbias};
  for (int i = threadIdx.x; i < 3 * C; i += 64 * WAVES) {
    const int k = i / C, c = i - k * C;
    if (!outs[k]) continue;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) sum += red[w][k][c];
    atomicAdd(outs[k] + c, sum);
  }
}

// ------------------------------------------------------------------------------------------------ column sums
template <typename T, bool RELU>
__global__ __launch_bounds__(256) void colsum_acc(T *__restrict__ x, const T *__restrict__ h, int rows, int N, int rows_per_blk,
                                                  float *__restrict__ acc)
{
  __shared__ float red[4][128];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c0 = blockIdx.x * 128 + lane * 2;
  const int r_begin = blockIdx.y * rows_per_blk, r_end = min(rows, r_begin + rows_per_blk);
  float a0 = 0.f, a1 = 0.f;
  int r = r_begin + wave;
  if (!RELU && sizeof(T) == 4) {                                  // plain fp32 column sums: four rows' loads in flight per lane (a dependent
    for (; r + 12 < r_end; r += 16) {                             // load per iteration left the 32 768-row sums at 2.4 TB/s)
      const float *p = reinterpret_cast<const float *>(x) + (int64_t)r * N + c0;
      const float2 u0 = *reinterpret_cast<const float2 *>(p), u1 = *reinterpret_cast<const float2 *>(p + (int64_t)4 * N);
      const float2 u2 = *reinterpret_cast<const float2 *>(p + (int64_t)8 * N), u3 = *reinterpret_cast<const float2 *>(p + (int64_t)12 * N);
      a0 += (u0.x + u1.x) + (u2.x + u3.x);
      a1 += (u0.y + u1.y) + (u2.y + u3.y);
    }
  }
  for (; r < r_end; r += 4) {
    T *p = x + (int64_t)r * N + c0;
    float v0, v1;
    if constexpr (sizeof(T) == 2) {
      const unsigned u = *reinterpret_cast<const unsigned *>(p);
      v0 = bf2f((bf16_t)(u & 0xffffu)); v1 = bf2f((bf16_t)(u >> 16));
    } else {
      const float2 u = *reinterpret_cast<const float2 *>(p);
      v0 = u.x; v1 = u.y;
    }
    if (RELU) {
      float h0, h1;
      if constexpr (sizeof(T) == 2) {
        const unsigned u = *reinterpret_cast<const unsigned *>(h + (int64_t)r * N + c0);
        h0 = bf2f((bf16_t)(u & 0xffffu)); h1 = bf2f((bf16_t)(u >> 16));
      } else {
        const float2 u = *reinterpret_cast<const float2 *>(h + (int64_t)r * N + c0);
        h0 = u.x; h1 = u.y;
      }
      if (!(h0 > 0.f)) v0 = 0.f;
      if (!(h1 > 0.f)) v1 = 0.f;
      if constexpr (sizeof(T) == 2) *reinterpret_cast<unsigned *>(p) = (unsigned)f2bf(v0) | ((unsigned)f2bf(v1) << 16);
      else *reinterpret_cast<float2 *>(p) = make_float2(v0, v1);
    }
    a0 += v0; a1 += v1;
  }
  if (!acc) return;
  red[wave][lane * 2] = a0; red[wave][lane * 2 + 1] = a1;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int c = threadIdx.x;
    atomicAdd(acc + blockIdx.x * 128 + c, (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
  }
}

// ------------------------------------------------------------------------------------------------ decoder memory
template <typename CT>
__global__ __launch_bounds__(256) void mem_prep_fwd(const float *__restrict__ tok, int64_t bstride, const float *__restrict__ lvl,
                                                    const float *__restrict__ pos, CT *__restrict__ mem, CT *__restrict__ mempos,
                                                    int B, int HW, int C)
{
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rows = B * HW;
  for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
    const int p = r / B, b = r - p * B;
    const float *src = tok + (int64_t)b * bstride + (int64_t)p * C;
    for (int c = lane * 4; c < C; c += 256) {
      float4 v = ld4(src + c);
      if (lvl) v = add4(v, ld4(lvl + c));
      if (mem) st4(mem + (int64_t)r * C + c, v);
      if (mempos) st4(mempos + (int64_t)r * C + c, add4(v, ld4(pos + (int64_t)p * C + c)));
    }
  }
}

template <typename CT>
__global__ __launch_bounds__(256) void mem_prep_bwd(const CT *__restrict__ dmem, const CT *__restrict__ dmempos,
                                                    float *__restrict__ dtok, int64_t bstride, int B, int HW, int C)
{
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rows = B * HW;
  for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {      // r enumerates the OUTPUT rows (b, p): coalesced stores
    const int b = r / HW, p = r - b * HW;
    const int64_t srow = ((int64_t)p * B + b) * C;
    float *dst = dtok + (int64_t)b * bstride + (int64_t)p * C;
    for (int c = lane * 4; c < C; c += 256) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (dmem) v = ld4(dmem + srow + c);
      if (dmempos) v = add4(v, ld4(dmempos + srow + c));
      st4(dst + c, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------ attention mask
template <typename T>
__global__ __launch_bounds__(256) void attn_mask_u8(const T *__restrict__ logits, int n, uint8_t *__restrict__ mask)
{
  __shared__ int any_open;
  const T *row = logits + (int64_t)blockIdx.x * n;
  uint8_t *out = mask + (int64_t)blockIdx.x * n;
  if (threadIdx.x == 0) any_open = 0;
  __syncthreads();
  int open = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    float v;
    if constexpr (sizeof(T) == 2) v = bf2f(row[i]); else v = row[i];
    open |= !(v < 0.f);
  }
  if (__ballot(open) != 0ull && (threadIdx.x & 63) == 0) atomicOr(&any_open, 1);
  __syncthreads();
  const bool keep = any_open != 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    float v;
    if constexpr (sizeof(T) == 2) v = bf2f(row[i]); else v = row[i];
    out[i] = (keep && v < 0.f) ? 1 : 0;
  }
}

// The bf16 rows of the decoder (n % 8 == 0, n <= 16 384): 8 logits per 16-byte load, the row stays in registers between the "is any key open"
// vote and the byte stores (8 per lane) — the element-wise form above reads the row twice, 2 bytes per lane, and stores single bytes (15.7 us per
// launch at 128^2 keys, as long as the product that made the logits).  v < 0 on the bf16 bits: sign set, not -0, not NaN.
__device__ __forceinline__ unsigned neg2(unsigned w)         // -> bit 0 / bit 8: low / high half of the word is a bf16 below zero
{
  const unsigned lo = w & 0xffffu, hi = w >> 16;
  const unsigned bl = (lo & 0x8000u) && (lo & 0x7fffu) != 0u && (lo & 0x7fffu) <= 0x7f80u;
  const unsigned bh = (hi & 0x8000u) && (hi & 0x7fffu) != 0u && (hi & 0x7fffu) <= 0x7f80u;
  return bl | (bh << 8);
}
constexpr int AMV = 8;                                       // 16-byte pieces per thread: 256 x 8 x 8 = 16 384 logits
__global__ __launch_bounds__(256) void attn_mask_u8_bf16x8(const bf16_t *__restrict__ logits, int n, uint8_t *__restrict__ mask)
{
  __shared__ int any_open;
  const uint4 *row = reinterpret_cast<const uint4 *>(logits + (int64_t)blockIdx.x * n);
  uint2 *out = reinterpret_cast<uint2 *>(mask + (int64_t)blockIdx.x * n);
  const int nv = n >> 3;
  if (threadIdx.x == 0) any_open = 0;
  uint2 b[AMV];                                              // the eight mask bytes of each piece
  unsigned all = 0x01010101u;
#pragma unroll
  for (int k = 0; k < AMV; ++k) {
    const int i = threadIdx.x + 256 * k;
    b[k] = make_uint2(0x01010101u, 0x01010101u);
    if (i < nv) {
      const uint4 v = row[i];
      const unsigned x = neg2(v.x), y = neg2(v.y), z = neg2(v.z), w = neg2(v.w);
      b[k] = make_uint2(x | (y << 16), z | (w << 16));
    }
    all &= b[k].x & b[k].y;
  }
  __syncthreads();
  if (__ballot(all != 0x01010101u) != 0ull && (threadIdx.x & 63) == 0) atomicOr(&any_open, 1);
  __syncthreads();
  const bool keep = any_open != 0;
#pragma unroll
  for (int k = 0; k < AMV; ++k) {
    const int i = threadIdx.x + 256 * k;
    if (i < nv) out[i] = keep ? b[k] : make_uint2(0u, 0u);
  }
}

// ------------------------------------------------------------------------------------------------ matcher costs
// One pass over the matcher's point logits x [rows, n] (reference matcher.py:108-158 batch_sigmoid_ce_loss / batch_dice_loss on
// the sampled points): x as fp32 (the operand of the x . target product), sigmoid(x), and per row sum softplus(x) and
// sum sigmoid(x) — ATen ran a cast, softplus, sigmoid and two reductions over the 100 MB tensor (183 us).  softplus as torch's
// (beta 1, threshold 20: x above it is returned as is).  One workgroup per row; fp32 sums, lanes -> wavefront -> workgroup.
template <typename T>
__global__ __launch_bounds__(256) void matcher_point_terms(const T *__restrict__ x, int n, float *__restrict__ xf, float *__restrict__ sg,
                                                           float *__restrict__ sp_sum, float *__restrict__ sg_sum)
{
  __shared__ float red[2][4];
  const T *row = x + (int64_t)blockIdx.x * n;
  float *xo = xf ? xf + (int64_t)blockIdx.x * n : nullptr, *so = sg + (int64_t)blockIdx.x * n;
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    float v;
    if constexpr (sizeof(T) == 2) v = bf2f(row[i]); else v = row[i];
    const float e = expf(-fabsf(v));
    const float sp = v > 20.f ? v : fmaxf(v, 0.f) + log1pf(e);
    const float s = 1.f / (1.f + expf(-v));
    if (xo) xo[i] = v;
    so[i] = s;
    a += sp; b += s;
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    sp_sum[blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    sg_sum[blockIdx.x] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// ------------------------------------------------------------------------------------------------ MSDeformAttn prep
// one thread per (token, head): softmax over the L*P logits, sampling locations = reference point + offset / (W_l, H_l).
// LP = L*P values per thread, moved as float4 (LP % 4 == 0, P even): a thread owns 4*LP contiguous bytes of logits / attn
// and 8*LP of offsets / locations, so a wavefront covers one contiguous span with 16-byte lanes.
template <int LP4>      // LP / 4
__global__ __launch_bounds__(256) void msda_prep_fwd_v(const float *__restrict__ offs, const float *__restrict__ logits,
                                                       const float *__restrict__ ref, const int64_t *__restrict__ shapes,
                                                       float *__restrict__ loc, float *__restrict__ attn, int64_t total, int M,
                                                       int L, int P, int ldo, int ldl)
{
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int64_t bq = t / M;
  const int m = (int)(t - bq * M);
  offs += bq * ldo + (int64_t)m * LP4 * 8 - t * LP4 * 8;          // re-base so the dense indexing below lands in the strided rows
  logits += bq * ldl + (int64_t)m * LP4 * 4 - t * LP4 * 4;
  float4 lg[LP4];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < LP4; ++i) {
    lg[i] = ld4(logits + (t * LP4 + i) * 4);
    mx = fmaxf(fmaxf(fmaxf(mx, lg[i].x), fmaxf(lg[i].y, lg[i].z)), lg[i].w);
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LP4; ++i) {
    lg[i].x = expf(lg[i].x - mx); lg[i].y = expf(lg[i].y - mx); lg[i].z = expf(lg[i].z - mx); lg[i].w = expf(lg[i].w - mx);
    sum += lg[i].x; sum += lg[i].y; sum += lg[i].z; sum += lg[i].w;           // same left-to-right order as the scalar loop
  }
#pragma unroll
  for (int i = 0; i < LP4; ++i)
    st4(attn + (t * LP4 + i) * 4, make_float4(lg[i].x / sum, lg[i].y / sum, lg[i].z / sum, lg[i].w / sum));
  const int half = P / 2;                                  // float4 = two (x, y) points of the same level
#pragma unroll
  for (int i = 0; i < 2 * LP4; ++i) {
    const int l = i / half;
    const float w = (float)shapes[l * 2 + 1], h = (float)shapes[l * 2];
    const float rx = ref[(bq * L + l) * 2], ry = ref[(bq * L + l) * 2 + 1];
    const float4 o = ld4(offs + (t * 2 * LP4 + i) * 4);
    st4(loc + (t * 2 * LP4 + i) * 4, make_float4(rx + o.x / w, ry + o.y / h, rx + o.z / w, ry + o.w / h));
  }
}

template <int LP4>
__global__ __launch_bounds__(256) void msda_prep_bwd_v(const float *__restrict__ gloc, const float *__restrict__ gattn,
                                                       const float *__restrict__ attn, const int64_t *__restrict__ shapes,
                                                       float *__restrict__ d_offs, float *__restrict__ d_logits, int64_t total,
                                                       int M, int L, int P, int ldo, int ldl, float *__restrict__ row_amax)
{
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int64_t bq0 = t / M;
  {
    const int64_t bq = bq0;
    const int m = (int)(t - bq * M);
    d_offs += bq * ldo + (int64_t)m * LP4 * 8 - t * LP4 * 8;
    d_logits += bq * ldl + (int64_t)m * LP4 * 4 - t * LP4 * 4;
  }
  float amx = 0.f;                                               // row_amax (M == 8 only): max |.| over the token's d_offs and d_logits
  float4 a[LP4], g[LP4];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < LP4; ++i) {
    a[i] = ld4(attn + (t * LP4 + i) * 4);
    g[i] = ld4(gattn + (t * LP4 + i) * 4);
    dot += a[i].x * g[i].x; dot += a[i].y * g[i].y; dot += a[i].z * g[i].z; dot += a[i].w * g[i].w;
  }
#pragma unroll
  for (int i = 0; i < LP4; ++i) {
    const float4 o = make_float4(a[i].x * (g[i].x - dot), a[i].y * (g[i].y - dot), a[i].z * (g[i].z - dot), a[i].w * (g[i].w - dot));
    st4(d_logits + (t * LP4 + i) * 4, o);
    amx = fmaxf(amx, amax4(o));
  }
  const int half = P / 2;
#pragma unroll
  for (int i = 0; i < 2 * LP4; ++i) {
    const int l = i / half;
    const float w = (float)shapes[l * 2 + 1], h = (float)shapes[l * 2];
    const float4 o = ld4(gloc + (t * 2 * LP4 + i) * 4);
    const float4 r = make_float4(o.x / w, o.y / h, o.z / w, o.w / h);
    st4(d_offs + (t * 2 * LP4 + i) * 4, r);
    amx = fmaxf(amx, amax4(r));
  }
  if (row_amax) {                                                // the 8 heads of a token are 8 consecutive lanes
    amx = fmaxf(amx, __shfl_xor(amx, 1, 64)); amx = fmaxf(amx, __shfl_xor(amx, 2, 64)); amx = fmaxf(amx, __shfl_xor(amx, 4, 64));
    if ((threadIdx.x & 7) == 0) row_amax[bq0] = amx;
  }
}

// any L, P: scalar loops
__global__ __launch_bounds__(256) void msda_prep_fwd(const float *__restrict__ offs, const float *__restrict__ logits,
                                                     const float *__restrict__ ref, const int64_t *__restrict__ shapes,
                                                     float *__restrict__ loc, float *__restrict__ attn, int64_t total, int M, int L,
                                                     int P, int ldo, int ldl)
{
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int LP = L * P;
  const int64_t bq = t / M;
  const int m = (int)(t - bq * M);
  const float *lg = logits + bq * ldl + (int64_t)m * LP;
  float mx = lg[0];
  for (int i = 1; i < LP; ++i) mx = fmaxf(mx, lg[i]);
  float sum = 0.f;
  for (int i = 0; i < LP; ++i) sum += expf(lg[i] - mx);
  float *ao = attn + t * LP;
  for (int i = 0; i < LP; ++i) ao[i] = expf(lg[i] - mx) / sum;
  const float *of = offs + bq * ldo + (int64_t)m * LP * 2;
  float *lo = loc + t * LP * 2;
  for (int l = 0; l < L; ++l) {
    const float w = (float)shapes[l * 2 + 1], h = (float)shapes[l * 2];
    const float rx = ref[(bq * L + l) * 2], ry = ref[(bq * L + l) * 2 + 1];
    for (int p = 0; p < P; ++p) {
      const int i = (l * P + p) * 2;
      lo[i] = rx + of[i] / w;
      lo[i + 1] = ry + of[i + 1] / h;
    }
  }
}

__global__ __launch_bounds__(256) void msda_prep_bwd(const float *__restrict__ gloc, const float *__restrict__ gattn,
                                                     const float *__restrict__ attn, const int64_t *__restrict__ shapes,
                                                     float *__restrict__ d_offs, float *__restrict__ d_logits, int64_t total,
                                                     int M, int L, int P, int ldo, int ldl)
{
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int LP = L * P;
  const int64_t bq = t / M;
  const int m = (int)(t - bq * M);
  const float *a = attn + t * LP, *g = gattn + t * LP;
  float dot = 0.f;
  for (int i = 0; i < LP; ++i) dot += a[i] * g[i];
  float *dl = d_logits + bq * ldl + (int64_t)m * LP;
  for (int i = 0; i < LP; ++i) dl[i] = a[i] * (g[i] - dot);
  const float *gl = gloc + t * LP * 2;
  float *dof = d_offs + bq * ldo + (int64_t)m * LP * 2;
  for (int l = 0; l < L; ++l) {
    const float w = (float)shapes[l * 2 + 1], h = (float)shapes[l * 2];
    for (int p = 0; p < P; ++p) {
      const int i = (l * P + p) * 2;
      dof[i] = gl[i] / w;
      dof[i + 1] = gl[i + 1] / h;
    }
  }
}

// ------------------------------------------------------------------------------------------------ point sampling
// F.grid_sample(bilinear, zeros, align_corners=False) of a channels-last map at points that are shared by all C channels:
// one wavefront per point, lanes across channels (16-byte pieces), so each corner is one contiguous C*4-byte burst.
template <typename TO>
__global__ __launch_bounds__(256) void point_sample_nhwc(const float *__restrict__ in, const float *__restrict__ coords,
                                                         TO *__restrict__ out, int B, int H, int W, int C, int P)
{
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t total = (int64_t)B * P;
  for (int64_t pt = (int64_t)blockIdx.x * 4 + wave; pt < total; pt += (int64_t)gridDim.x * 4) {
    const int b = (int)(pt / P);
    // the exact arithmetic of the torch path: g = 2*c - 1, then ((g + 1) * size - 1) / 2
    const float gx = 2.0f * coords[pt * 2] - 1.0f, gy = 2.0f * coords[pt * 2 + 1] - 1.0f;
    const float ix = ((gx + 1.f) * W - 1.f) / 2.f, iy = ((gy + 1.f) * H - 1.f) / 2.f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wnw = (x1 - ix) * (y1 - iy), wne = (ix - x0) * (y1 - iy), wsw = (x1 - ix) * (iy - y0), wse = (ix - x0) * (iy - y0);
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    const float *base = in + (int64_t)b * H * W * C;
    for (int c = lane * 4; c < C; c += 256) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (vy0 && vx0) { const float4 v = ld4(base + ((int64_t)y0 * W + x0) * C + c); acc.x += v.x * wnw; acc.y += v.y * wnw; acc.z += v.z * wnw; acc.w += v.w * wnw; }
      if (vy0 && vx1) { const float4 v = ld4(base + ((int64_t)y0 * W + x1) * C + c); acc.x += v.x * wne; acc.y += v.y * wne; acc.z += v.z * wne; acc.w += v.w * wne; }
      if (vy1 && vx0) { const float4 v = ld4(base + ((int64_t)y1 * W + x0) * C + c); acc.x += v.x * wsw; acc.y += v.y * wsw; acc.z += v.z * wsw; acc.w += v.w * wsw; }
      if (vy1 && vx1) { const float4 v = ld4(base + ((int64_t)y1 * W + x1) * C + c); acc.x += v.x * wse; acc.y += v.y * wse; acc.z += v.z * wse; acc.w += v.w * wse; }
      st4(out + pt * C + c, acc);
    }
  }
}

// The same sampling of PLANAR maps [N, C, H, W] at per-map points coords[N, P, 2] -> out[N, C, P] (the criterion samples
// single-channel mask logits and target masks at 10^5..10^6 points: torch's generic grid_sampler takes 50 us for what is a
// few MB of gathers), and its gradient with respect to the maps (atomic scatter into a zero-filled gradient).
__global__ __launch_bounds__(256) void point_sample_planar_fwd(const float *__restrict__ in, const float *__restrict__ coords,
                                                               float *__restrict__ out, int N, int C, int H, int W, int P)
{
  const int64_t total = (int64_t)N * P;
  for (int64_t pt = (int64_t)blockIdx.x * 256 + threadIdx.x; pt < total; pt += (int64_t)gridDim.x * 256) {
    const int n = (int)(pt / P), p = (int)(pt - (int64_t)n * P);
    const float gx = 2.0f * coords[pt * 2] - 1.0f, gy = 2.0f * coords[pt * 2 + 1] - 1.0f;
    const float ix = ((gx + 1.f) * W - 1.f) / 2.f, iy = ((gy + 1.f) * H - 1.f) / 2.f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wnw = (x1 - ix) * (y1 - iy), wne = (ix - x0) * (y1 - iy), wsw = (x1 - ix) * (iy - y0), wse = (ix - x0) * (iy - y0);
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    // (the four corners as straight-line loads from clamped addresses, an outside corner's VALUE replaced by 0 before the multiply — exact, and
    //  no longer four dependent memory round trips per point behind `if (valid)`)
    const int xa = min(max(x0, 0), W - 1), xb = min(max(x1, 0), W - 1), ya = min(max(y0, 0), H - 1), yb = min(max(y1, 0), H - 1);
    for (int c = 0; c < C; ++c) {
      const float *m = in + ((int64_t)n * C + c) * H * W;
      const float v00 = m[(int64_t)ya * W + xa], v01 = m[(int64_t)ya * W + xb], v10 = m[(int64_t)yb * W + xa], v11 = m[(int64_t)yb * W + xb];
      float acc = 0.f;
      if (vy0 && vx0) acc += v00 * wnw;
      if (vy0 && vx1) acc += v01 * wne;
      if (vy1 && vx0) acc += v10 * wsw;
      if (vy1 && vx1) acc += v11 * wse;
      out[((int64_t)n * C + c) * P + p] = acc;
    }
  }
}

__global__ __launch_bounds__(256) void point_sample_planar_bwd(const float *__restrict__ gout, const float *__restrict__ coords,
                                                               float *__restrict__ gin, int N, int C, int H, int W, int P)
{
  const int64_t total = (int64_t)N * P;
  for (int64_t pt = (int64_t)blockIdx.x * 256 + threadIdx.x; pt < total; pt += (int64_t)gridDim.x * 256) {
    const int n = (int)(pt / P), p = (int)(pt - (int64_t)n * P);
    const float gx = 2.0f * coords[pt * 2] - 1.0f, gy = 2.0f * coords[pt * 2 + 1] - 1.0f;
    const float ix = ((gx + 1.f) * W - 1.f) / 2.f, iy = ((gy + 1.f) * H - 1.f) / 2.f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wnw = (x1 - ix) * (y1 - iy), wne = (ix - x0) * (y1 - iy), wsw = (x1 - ix) * (iy - y0), wse = (ix - x0) * (iy - y0);
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    for (int c = 0; c < C; ++c) {
      float *m = gin + ((int64_t)n * C + c) * H * W;
      const float g = gout[((int64_t)n * C + c) * P + p];
      if (vy0 && vx0) unsafeAtomicAdd(m + (int64_t)y0 * W + x0, g * wnw);
      if (vy0 && vx1) unsafeAtomicAdd(m + (int64_t)y0 * W + x1, g * wne);
      if (vy1 && vx0) unsafeAtomicAdd(m + (int64_t)y1 * W + x0, g * wsw);
      if (vy1 && vx1) unsafeAtomicAdd(m + (int64_t)y1 * W + x1, g * wse);
    }
  }
}

// Single-channel maps small enough to tile through LDS (the matched mask logits, [80, 1, 256, 256]): the random 4-byte gathers /
// atomics above run at ~0.2 G points/ms whatever the kernel (193 us for the gradient of 1 M points).  Here a workgroup owns one
// T x T tile of one map, scans ALL points of that map (coordinates are 8 bytes, the scan is cheap) and serves the ones whose
// clamped top-left corner (forward) / whose corner pixels (backward) fall inside its tile from LDS: no global atomics, no
// zero-fill of the gradient, every output written exactly once.
constexpr int PST = 112;                                 // tile edge; (T + 1)^2 floats = 51 KB with the forward's right / bottom halo
__global__ __launch_bounds__(256) void point_sample_tiled_fwd(const float *__restrict__ in, const float *__restrict__ coords,
                                                              float *__restrict__ out, int H, int W, int P, int tiles_x)
{
  extern __shared__ float tile[];                        // [(T + 1)][(T + 1)]
  const int n = blockIdx.y, ty0 = (blockIdx.x / tiles_x) * PST, tx0 = (blockIdx.x % tiles_x) * PST;
  const float *m = in + (int64_t)n * H * W;
  for (int i = threadIdx.x; i < (PST + 1) * (PST + 1); i += 256) {
    const int y = ty0 + i / (PST + 1), x = tx0 + i % (PST + 1);
    tile[i] = (y < H && x < W) ? m[(int64_t)y * W + x] : 0.f;
  }
  __syncthreads();
  const float *cn = coords + (int64_t)n * P * 2;
  for (int p = threadIdx.x; p < P; p += 256) {
    const float2 c = *reinterpret_cast<const float2 *>(cn + 2 * p);
    const float gx = 2.0f * c.x - 1.0f, gy = 2.0f * c.y - 1.0f;
    const float ix = ((gx + 1.f) * W - 1.f) / 2.f, iy = ((gy + 1.f) * H - 1.f) / 2.f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)W), y0 = (int)fminf(fmaxf(fy, -2.f), (float)H);   // far-out points: any invalid corner
    const int xc = min(max(x0, 0), W - 1), yc = min(max(y0, 0), H - 1);
    if (xc < tx0 || xc >= tx0 + PST || yc < ty0 || yc >= ty0 + PST) continue;        // another tile's point
    const int x1 = x0 + 1, y1 = y0 + 1;
    const float wnw = (x1 - ix) * (y1 - iy), wne = (ix - x0) * (y1 - iy), wsw = (x1 - ix) * (iy - y0), wse = (ix - x0) * (iy - y0);
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    const int lx = x0 - tx0, ly = y0 - ty0;
    float acc = 0.f;
    if (vy0 && vx0) acc += tile[ly * (PST + 1) + lx] * wnw;
    if (vy0 && vx1) acc += tile[ly * (PST + 1) + lx + 1] * wne;
    if (vy1 && vx0) acc += tile[(ly + 1) * (PST + 1) + lx] * wsw;
    if (vy1 && vx1) acc += tile[(ly + 1) * (PST + 1) + lx + 1] * wse;
    out[(int64_t)n * P + p] = acc;
  }
}

__global__ __launch_bounds__(256) void point_sample_tiled_bwd(const float *__restrict__ gout, const float *__restrict__ coords,
                                                              float *__restrict__ gin, int H, int W, int P, int tiles_x)
{
  extern __shared__ float tile[];                        // [T][T] accumulators
  const int n = blockIdx.y, ty0 = (blockIdx.x / tiles_x) * PST, tx0 = (blockIdx.x % tiles_x) * PST;
  for (int i = threadIdx.x; i < PST * PST; i += 256) tile[i] = 0.f;
  __syncthreads();
  const float *cn = coords + (int64_t)n * P * 2, *gn = gout + (int64_t)n * P;
  for (int p = threadIdx.x; p < P; p += 256) {
    const float2 c = *reinterpret_cast<const float2 *>(cn + 2 * p);
    const float gx = 2.0f * c.x - 1.0f, gy = 2.0f * c.y - 1.0f;
    const float ix = ((gx + 1.f) * W - 1.f) / 2.f, iy = ((gy + 1.f) * H - 1.f) / 2.f;
    const float fx = floorf(ix), fy = floorf(iy);
    if (fx < tx0 - 1.f || fx >= tx0 + PST || fy < ty0 - 1.f || fy >= ty0 + PST) continue;    // no corner of this point in the tile
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float g = gn[p];
    const float wnw = (x1 - ix) * (y1 - iy), wne = (ix - x0) * (y1 - iy), wsw = (x1 - ix) * (iy - y0), wse = (ix - x0) * (iy - y0);
    const int lx0 = x0 - tx0, ly0 = y0 - ty0, lx1 = lx0 + 1, ly1 = ly0 + 1;
    const bool ax0 = lx0 >= 0 && lx0 < PST && x0 < W, ax1 = lx1 >= 0 && lx1 < PST && x1 < W;
    const bool ay0 = ly0 >= 0 && ly0 < PST && y0 < H, ay1 = ly1 >= 0 && ly1 < PST && y1 < H;
    if (ay0 && ax0) atomicAdd(&tile[ly0 * PST + lx0], g * wnw);
    if (ay0 && ax1) atomicAdd(&tile[ly0 * PST + lx1], g * wne);
    if (ay1 && ax0) atomicAdd(&tile[ly1 * PST + lx0], g * wsw);
    if (ay1 && ax1) atomicAdd(&tile[ly1 * PST + lx1], g * wse);
  }
  __syncthreads();
  float *m = gin + (int64_t)n * H * W;
  for (int i = threadIdx.x; i < PST * PST; i += 256) {
    const int y = ty0 + i / PST, x = tx0 + i % PST;
    if (y < H && x < W) m[(int64_t)y * W + x] = tile[i];
  }
}

// ------------------------------------------------------------------------------------------------ FPN top-down step
// y = cur + bilinear_upsample(lo) (align_corners=False, torch's upsample_bilinear2d arithmetic), channels-last fp32;
// the backward for the exact-2x case in GATHER form: every low-resolution pixel sums its <= 4 x 4 contributing outputs.
__device__ __forceinline__ void up_src(int dst, float scale, int in_size, int &i0, int &ip, float &l0, float &l1)
{
  float src = scale * (dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  ip = (i0 < in_size - 1) ? 1 : 0;
  l1 = src - i0;
  l0 = 1.f - l1;
}

__global__ __launch_bounds__(256) void upsample_add_nhwc(const float *__restrict__ lo, const float *__restrict__ cur,
                                                         float *__restrict__ y, int B, int h, int w, int H, int W, int C4,
                                                         float sh, float sw, float *__restrict__ amax, int64_t lo_bs4)
{
  const int64_t total = (int64_t)B * H * W * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    int64_t t = i / C4;
    const int ox = (int)(t % W); t /= W;
    const int oy = (int)(t % H);
    const int b = (int)(t / H);
    int y0, yp, x0, xp; float hy0, hy1, wx0, wx1;
    up_src(oy, sh, h, y0, yp, hy0, hy1);
    up_src(ox, sw, w, x0, xp, wx0, wx1);
    const float4 *L = reinterpret_cast<const float4 *>(lo) + (int64_t)b * lo_bs4;         // lo_bs4: batch stride of lo in float4 (a level of a token tensor)
    const float4 a = L[((int64_t)y0 * w + x0) * C4 + c], bq = L[((int64_t)y0 * w + x0 + xp) * C4 + c];
    const float4 cq = L[((int64_t)(y0 + yp) * w + x0) * C4 + c], d = L[((int64_t)(y0 + yp) * w + x0 + xp) * C4 + c];
    const float4 u = reinterpret_cast<const float4 *>(cur)[i];
    float4 o;
    o.x = u.x + (hy0 * (wx0 * a.x + wx1 * bq.x) + hy1 * (wx0 * cq.x + wx1 * d.x));
    o.y = u.y + (hy0 * (wx0 * a.y + wx1 * bq.y) + hy1 * (wx0 * cq.y + wx1 * d.y));
    o.z = u.z + (hy0 * (wx0 * a.z + wx1 * bq.z) + hy1 * (wx0 * cq.z + wx1 * d.z));
    o.w = u.w + (hy0 * (wx0 * a.w + wx1 * bq.w) + hy1 * (wx0 * cq.w + wx1 * d.w));
    reinterpret_cast<float4 *>(y)[i] = o;
    if (amax) {                                                    // C4 == 64: a wavefront holds one output pixel
      float m = amax4(o);
#pragma unroll
      for (int of = 32; of; of >>= 1) m = fmaxf(m, __shfl_xor(m, of, 64));
      if ((threadIdx.x & 63) == 0) amax[i >> 6] = m;
    }
  }
}

// F.interpolate(x, size, mode="bilinear", align_corners=False) of a channels-last fp32 map to up to PD_RESIZE_MAX sizes in ONE launch, the
// results written as [B, h w, C] rows in bf16 or fp32 (the decoder's three pooled copies of the mask features, reference
// mask2former_transformer_decoder.py:452: three ATen resizes + casts).  One lane per 4 channels of an output pixel; torch's arithmetic.
struct ResizeLevels { void *out[PD_RESIZE_MAX]; int h[PD_RESIZE_MAX], w[PD_RESIZE_MAX]; int64_t first[PD_RESIZE_MAX + 1]; int count; };

template <bool BF16>
__global__ __launch_bounds__(256) void resize_bilinear_nhwc_multi(const float *__restrict__ x, int B, int H, int W, int C4, ResizeLevels lv)
{
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < lv.first[lv.count]; i += (int64_t)gridDim.x * 256) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < PD_RESIZE_MAX; ++k) l += (k < lv.count && i >= lv.first[k]) ? 1 : 0;
    const int h = lv.h[l], w = lv.w[l];
    int64_t t = i - lv.first[l];
    const int c = (int)(t % C4); t /= C4;
    const int ox = (int)(t % w); t /= w;
    const int oy = (int)(t % h);
    const int b = (int)(t / h);
    int y0, yp, x0, xp; float hy0, hy1, wx0, wx1;
    up_src(oy, (float)H / (float)h, H, y0, yp, hy0, hy1);
    up_src(ox, (float)W / (float)w, W, x0, xp, wx0, wx1);
    const float4 *L = reinterpret_cast<const float4 *>(x) + (int64_t)b * H * W * C4;
    const float4 a = L[((int64_t)y0 * W + x0) * C4 + c], bq = L[((int64_t)y0 * W + x0 + xp) * C4 + c];
    const float4 cq = L[((int64_t)(y0 + yp) * W + x0) * C4 + c], d = L[((int64_t)(y0 + yp) * W + x0 + xp) * C4 + c];
    float4 o;
    o.x = hy0 * (wx0 * a.x + wx1 * bq.x) + hy1 * (wx0 * cq.x + wx1 * d.x);
    o.y = hy0 * (wx0 * a.y + wx1 * bq.y) + hy1 * (wx0 * cq.y + wx1 * d.y);
    o.z = hy0 * (wx0 * a.z + wx1 * bq.z) + hy1 * (wx0 * cq.z + wx1 * d.z);
    o.w = hy0 * (wx0 * a.w + wx1 * bq.w) + hy1 * (wx0 * cq.w + wx1 * d.w);
    const int64_t oi = (((int64_t)b * h + oy) * w + ox) * C4 + c;
    if (BF16) st4(reinterpret_cast<bf16_t *>(lv.out[l]) + oi * 4, o);
    else st4(reinterpret_cast<float *>(lv.out[l]) + oi * 4, o);
  }
}

__global__ __launch_bounds__(256) void upsample2x_bwd_nhwc(const float *__restrict__ dy, float *__restrict__ dlo, int B, int h,
                                                           int w, int C4)
{
  const int H = 2 * h, W = 2 * w;
  const int64_t total = (int64_t)B * h * w * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    int64_t t = i / C4;
    const int ix = (int)(t % w); t /= w;
    const int iy = (int)(t % h);
    const int b = (int)(t / h);
    float wy[4], wx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int oy = 2 * iy - 1 + k, ox = 2 * ix - 1 + k;
      wy[k] = 0.f; wx[k] = 0.f;
      if (oy >= 0 && oy < H) {
        int i0, ip; float l0, l1;
        up_src(oy, 0.5f, h, i0, ip, l0, l1);
        if (i0 == iy) wy[k] += l0;
        if (i0 + ip == iy) wy[k] += l1;
      }
      if (ox >= 0 && ox < W) {
        int i0, ip; float l0, l1;
        up_src(ox, 0.5f, w, i0, ip, l0, l1);
        if (i0 == ix) wx[k] += l0;
        if (i0 + ip == ix) wx[k] += l1;
      }
    }
    const float4 *G = reinterpret_cast<const float4 *>(dy) + (int64_t)b * H * W * C4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // all sixteen taps are LOADED, from clamped coordinates, and dropped by value where their weight is zero (image border): behind `if (weight == 0)
    // continue` every load waited for the one before it (15 s_waitcnt vmcnt(0) between 16 loads in the ISA; 52.7 us for 168 MB at config 2)
    float4 gt[4][4];
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int oy = min(max(2 * iy - 1 + ky, 0), H - 1);
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) gt[ky][kx] = G[((int64_t)oy * W + min(max(2 * ix - 1 + kx, 0), W - 1)) * C4 + c];
    }
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const float wgt = wy[ky] * wx[kx];
        const bool on = wy[ky] != 0.f && wx[kx] != 0.f;
        const float4 g = gt[ky][kx];
        acc.x = on ? fmaf(wgt, g.x, acc.x) : acc.x; acc.y = on ? fmaf(wgt, g.y, acc.y) : acc.y;      // (the fused form the `+=` of the branchy loop contracted to)
        acc.z = on ? fmaf(wgt, g.z, acc.z) : acc.z; acc.w = on ? fmaf(wgt, g.w, acc.w) : acc.w;
      }
    }
    reinterpret_cast<float4 *>(dlo)[i] = acc;
  }
}

int grid_rows(int rows, int cap) { return max(1, min(cap, (rows + 3) / 4)); }

// ------------------------------------------------------------------------------------------------ batched fp32 transposes
// dst_i [batch, cols, rows] = src_i [batch, rows, cols]^T for up to PD_TRANSPOSE_MAX problems in ONE launch (the encoder backward's five
// transposed weight stacks: five strided ATen copies of ~12 us each for 18 MB).  32 x 32 tiles through LDS, coalesced both ways.
struct TrBatch { PdTransposeProblem p[PD_TRANSPOSE_MAX]; int first_tile[PD_TRANSPOSE_MAX + 1]; int count; };

__global__ __launch_bounds__(256) void transpose_batched_f32(TrBatch tb)
{
  __shared__ float tile[32][33];
  int pi = 0;
#pragma unroll
  for (int i = 1; i < PD_TRANSPOSE_MAX; ++i) pi += (i < tb.count && (int)blockIdx.x >= tb.first_tile[i]) ? 1 : 0;
  const PdTransposeProblem &pr = tb.p[pi];
  const int tr = (pr.rows + 31) / 32, tc = (pr.cols + 31) / 32;
  int t = blockIdx.x - tb.first_tile[pi];
  const int b = t / (tr * tc); t -= b * tr * tc;
  const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float *src = (const float *)pr.src + (int64_t)b * pr.src_batch_stride;
  float *dst = (float *)pr.dst + (int64_t)b * pr.rows * pr.cols;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    if (r < pr.rows && c < pr.cols) tile[ty + 8 * k][tx] = src[(int64_t)r * pr.src_row_stride + c];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (r < pr.rows && c < pr.cols) dst[(int64_t)c * pr.rows + r] = tile[tx][ty + 8 * k];
  }
}

// out1 = (a + b) + c and out2 = d + c in one pass (the end of the encoder's backward: d(src) from its three fp32 terms, d(pos) from its

// the encoder's entry in one pass (was an ATen add, two device-to-device copies and two row-maxima launches): q = a + b, an optional copy
// of a (the recorded region needs its first operand at an address of its own), and the absolute maxima of the rows of a and of q — the
// row scales of the two-plane GEMMs that read them.  One wavefront per row, 16 bytes per lane and pass.
__global__ __launch_bounds__(256) void add_rows_amax_f32(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ q,
                                                         float *__restrict__ a_copy, float *__restrict__ a_amax, float *__restrict__ q_amax,
                                                         int rows, int cols)
{
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int64_t base = (int64_t)row * cols;
  float ma = 0.f, mq = 0.f;
  for (int c = lane * 4; c < cols; c += 256) {
    const float4 x = ld4(a + base + c), y = ld4(b + base + c);
    const float4 s = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    st4(q + base + c, s);
    if (a_copy) st4(a_copy + base + c, x);
    ma = fmaxf(ma, amax4(x)); mq = fmaxf(mq, amax4(s));
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) { ma = fmaxf(ma, __shfl_xor(ma, o, 64)); mq = fmaxf(mq, __shfl_xor(mq, o, 64)); }
  if (lane == 0) { a_amax[row] = ma; q_amax[row] = mq; }
}

// accumulator and the last term — three ATen launches over [43 008, 256] before)
__global__ __launch_bounds__(256) void sum3_sum2_f32(const float4 *__restrict__ a, const float4 *__restrict__ b, const float4 *__restrict__ c,
                                                     const float4 *__restrict__ d, float4 *__restrict__ o1, float4 *__restrict__ o2, int64_t n4)
{
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 x = a[i], y = b[i], z = c[i], w = d[i];
    o1[i] = make_float4((x.x + y.x) + z.x, (x.y + y.y) + z.y, (x.z + y.z) + z.z, (x.w + y.w) + z.w);
    o2[i] = make_float4(w.x + z.x, w.y + z.y, w.z + z.z, w.w + z.w);
  }
}

bool dt_ok(int dt) { return dt == PD_F32 || dt == PD_BF16; }

}  // namespace

#define NV4_SWITCH(C, BODY)                                                                                        \
  switch ((C) / 256) {                                                                                             \
    case 1: { constexpr int NV4 = 1; BODY; } break;                                                                \
    case 2: { constexpr int NV4 = 2; BODY; } break;                                                                \
    case 3: { constexpr int NV4 = 3; BODY; } break;                                                                \
    case 4: { constexpr int NV4 = 4; BODY; } break;                                                                \
    default: return pd_set_error(PD_ERR_INVALID_ARG, "C=%d: supported widths are 256, 512, 768, 1024", (C)); \
  }

static int add_layernorm_fwd(const void *x, int x_dtype, const float *res, const float *gamma, const float *beta, float eps,
                             float *z, float *y, void *y_c, const float *pos, int pos_div, void *ypos_c, int c_dtype,
                             float *mean, float *rstd, int rows, int C, void *stream_, float *y_amax, float *ypos_amax)
{
  if (rows < 0 || C <= 0 || (C % 256)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_add_layernorm_fwd: rows=%d C=%d", rows, C);
  if (rows == 0) return PD_OK;
  if ((!x && !res) || !gamma || !beta || !mean || !rstd) return pd_set_error(PD_ERR_INVALID_ARG, "pd_add_layernorm_fwd: null pointer");
  if ((x && !dt_ok(x_dtype)) || ((y_c || ypos_c) && !dt_ok(c_dtype))) return pd_set_error(PD_ERR_INVALID_ARG, "pd_add_layernorm_fwd: dtype");
  if (ypos_c && (!pos || pos_div <= 0)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_add_layernorm_fwd: ypos_c needs pos and pos_div > 0");
  hipStream_t s = (hipStream_t)stream_;
  dim3 g(grid_rows(rows, 2048)), b(256);
  const bool xb = x && x_dtype == PD_BF16, cb = c_dtype == PD_BF16;
#define LAUNCH(XT, CT) hipLaunchKernelGGL((add_ln_fwd<NV4, XT, CT>), g, b, 0, s, (const XT *)x, res, gamma, beta, eps, z, y, (CT *)y_c, pos, pos_div, (CT *)ypos_c, mean, rstd, rows, y_amax, ypos_amax)
  NV4_SWITCH(C, {
    if (xb) { if (cb) LAUNCH(bf16_t, bf16_t); else LAUNCH(bf16_t, float); }
    else { if (cb) LAUNCH(float, bf16_t); else LAUNCH(float, float); }
  })
#undef LAUNCH
  return pd_check_launch("pd_add_layernorm_fwd");
}

extern "C" int pd_add_layernorm_fwd(const void *x, int x_dtype, const float *res, const float *gamma, const float *beta, float eps,
                                    float *z, float *y, void *y_c, const float *pos, int pos_div, void *ypos_c, int c_dtype,
                                    float *mean, float *rstd, int rows, int C, void *stream_)
{
  return add_layernorm_fwd(x, x_dtype, res, gamma, beta, eps, z, y, y_c, pos, pos_div, ypos_c, c_dtype, mean, rstd, rows, C, stream_, nullptr, nullptr);
}

extern "C" int pd_add_layernorm_fwd_amax(const void *x, int x_dtype, const float *res, const float *gamma, const float *beta, float eps,
                                         float *z, float *y, void *y_c, const float *pos, int pos_div, void *ypos_c, int c_dtype,
                                         float *mean, float *rstd, float *y_amax, float *ypos_amax, int rows, int C, void *stream_)
{
  if (ypos_amax && !ypos_c) return pd_set_error(PD_ERR_INVALID_ARG, "pd_add_layernorm_fwd_amax: ypos_amax needs ypos_c");
  return add_layernorm_fwd(x, x_dtype, res, gamma, beta, eps, z, y, y_c, pos, pos_div, ypos_c, c_dtype, mean, rstd, rows, C, stream_, y_amax, ypos_amax);
}

static int add_layernorm_bwd(const float *dy, const float *dy2, const void *dy_c, const void *dypos_c, int c_dtype,
                             const float *z, const float *mean, const float *rstd, const float *gamma, float *dz,
                             void *dz_c, int dzc_dtype, float *dgamma, float *dbeta, float *dbias, float *dpos_acc,
                             int pos_div, int rows, int C, void *stream_, float *dz_amax)
{
  if (rows < 0 || C <= 0 || (C % 256)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_add_layernorm_bwd: rows=%d C=%d", rows, C);
  if (rows == 0) return PD_OK;
  if ((!dy && !dy2 && !dy_c && !dypos_c) || !z || !mean || !rstd || !gamma || !dz)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_add_layernorm_bwd: null pointer");
  if (((dy_c || dypos_c) && !dt_ok(c_dtype)) || (dz_c && !dt_ok(dzc_dtype))) return pd_set_error(PD_ERR_INVALID_ARG, "pd_add_layernorm_bwd: dtype");
  if (dpos_acc && pos_div <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_add_layernorm_bwd: pos_div");
  hipStream_t s = (hipStream_t)stream_;
  // fat workgroups, few of them (see the kernel): 16 / 8 / 4 wavefronts by width (48 KB of column sums in LDS), >= 4 rows per wavefront,
  // at most 256 workgroups (one per CU: 23.6 us at [43 008, 256] against 26.4 with 512 and 42.0 with 1 024 four-wavefront workgroups)
  const bool cb = c_dtype == PD_BF16, db = dzc_dtype == PD_BF16;
#define LAUNCH(CT, DT, WV) hipLaunchKernelGGL((add_ln_bwd<NV4, CT, DT, WV>), g, b, 0, s, dy, dy2, (const CT *)dy_c, (const CT *)dypos_c, z, mean, rstd, gamma, dz, (DT *)dz_c, dgamma, dbeta, dbias, dpos_acc, pos_div, rows, dz_amax)
#define LAUNCH_W(WV)                                                                    \
  if (cb) { if (db) LAUNCH(bf16_t, bf16_t, WV); else LAUNCH(bf16_t, float, WV); }      \
  else { if (db) LAUNCH(float, bf16_t, WV); else LAUNCH(float, float, WV); }
  NV4_SWITCH(C, {
    constexpr int FAT = NV4 == 1 ? 16 : NV4 == 2 ? 8 : 4;
    if (rows >= 8192 && FAT > 4 && g_ln_bwd_cap >= 0) {
      const int cap = g_ln_bwd_cap > 0 ? g_ln_bwd_cap : 256;
      const dim3 g(max(1, min(cap, (rows + 4 * FAT - 1) / (4 * FAT))));
      const dim3 b(64 * FAT);
      LAUNCH_W(FAT)
    } else {                                             // few rows (the decoder's 200): one row per wavefront at a time, as many workgroups as rows / 4
      const dim3 g(grid_rows(rows, 1024));
      const dim3 b(256);
      LAUNCH_W(4)
    }
  })
#undef LAUNCH_W
#undef LAUNCH
  return pd_check_launch("pd_add_layernorm_bwd");
}

extern "C" int pd_add_layernorm_bwd(const float *dy, const float *dy2, const void *dy_c, const void *dypos_c, int c_dtype,
                                    const float *z, const float *mean, const float *rstd, const float *gamma, float *dz,
                                    void *dz_c, int dzc_dtype, float *dgamma, float *dbeta, float *dbias, float *dpos_acc,
                                    int pos_div, int rows, int C, void *stream_)
{
  return add_layernorm_bwd(dy, dy2, dy_c, dypos_c, c_dtype, z, mean, rstd, gamma, dz, dz_c, dzc_dtype, dgamma, dbeta, dbias, dpos_acc, pos_div, rows, C,
                           stream_, nullptr);
}

extern "C" int pd_add_layernorm_bwd_amax(const float *dy, const float *dy2, const void *dy_c, const void *dypos_c, int c_dtype,
                                         const float *z, const float *mean, const float *rstd, const float *gamma, float *dz,
                                         void *dz_c, int dzc_dtype, float *dgamma, float *dbeta, float *dbias, float *dpos_acc,
                                         int pos_div, float *dz_amax, int rows, int C, void *stream_)
{
  return add_layernorm_bwd(dy, dy2, dy_c, dypos_c, c_dtype, z, mean, rstd, gamma, dz, dz_c, dzc_dtype, dgamma, dbeta, dbias, dpos_acc, pos_div, rows, C,
                           stream_, dz_amax);
}

static int colsum_launch(void *x, const void *h, int dtype, int rows, int N, float *acc, bool relu, hipStream_t s, const char *who)
{
  if (rows < 0 || N <= 0 || (N % 128) || !dt_ok(dtype)) return pd_set_error(PD_ERR_INVALID_ARG, "%s: rows=%d N=%d dtype=%d", who, rows, N, dtype);
  if (rows == 0) return PD_OK;
  if (!x || (relu && !h) || (!relu && !acc)) return pd_set_error(PD_ERR_INVALID_ARG, "%s: null pointer", who);
  const int rpb = rows <= 4096 ? 64 : 256;
  dim3 g(N / 128, (rows + rpb - 1) / rpb), b(256);
  if (dtype == PD_BF16) {
    if (relu) hipLaunchKernelGGL((colsum_acc<bf16_t, true>), g, b, 0, s, (bf16_t *)x, (const bf16_t *)h, rows, N, rpb, acc);
    else hipLaunchKernelGGL((colsum_acc<bf16_t, false>), g, b, 0, s, (bf16_t *)x, (const bf16_t *)h, rows, N, rpb, acc);
  } else {
    if (relu) hipLaunchKernelGGL((colsum_acc<float, true>), g, b, 0, s, (float *)x, (const float *)h, rows, N, rpb, acc);
    else hipLaunchKernelGGL((colsum_acc<float, false>), g, b, 0, s, (float *)x, (const float *)h, rows, N, rpb, acc);
  }
  return pd_check_launch(who);
}

extern "C" int pd_colsum_acc(const void *x, int dtype, int rows, int N, float *acc, void *stream_)
{
  return colsum_launch(const_cast<void *>(x), nullptr, dtype, rows, N, acc, false, (hipStream_t)stream_, "pd_colsum_acc");
}

extern "C" int pd_relu_bwd_colsum(void *dh, const void *h, int dtype, int rows, int N, float *acc, void *stream_)
{
  return colsum_launch(dh, h, dtype, rows, N, acc, true, (hipStream_t)stream_, "pd_relu_bwd_colsum");
}

extern "C" int pd_mem_prep_fwd(const float *tok, int64_t tok_batch_stride, const float *level_embed, const float *pos, void *mem_c,
                               void *mempos_c, int c_dtype, int B, int HW, int C, void *stream_)
{
  if (B < 0 || HW < 0 || C <= 0 || (C % 256) || !dt_ok(c_dtype)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_mem_prep_fwd: B=%d HW=%d C=%d", B, HW, C);
  if (B == 0 || HW == 0) return PD_OK;
  if (!tok || (!mem_c && !mempos_c) || (mempos_c && !pos)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_mem_prep_fwd: null pointer");
  dim3 g(grid_rows(B * HW, 4096)), b(256);
  if (c_dtype == PD_BF16) hipLaunchKernelGGL((mem_prep_fwd<bf16_t>), g, b, 0, (hipStream_t)stream_, tok, tok_batch_stride, level_embed, pos, (bf16_t *)mem_c, (bf16_t *)mempos_c, B, HW, C);
  else hipLaunchKernelGGL((mem_prep_fwd<float>), g, b, 0, (hipStream_t)stream_, tok, tok_batch_stride, level_embed, pos, (float *)mem_c, (float *)mempos_c, B, HW, C);
  return pd_check_launch("pd_mem_prep_fwd");
}

extern "C" int pd_mem_prep_bwd(const void *dmem_c, const void *dmempos_c, int c_dtype, float *dtok, int64_t dtok_batch_stride, int B,
                               int HW, int C, void *stream_)
{
  if (B < 0 || HW < 0 || C <= 0 || (C % 256) || !dt_ok(c_dtype)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_mem_prep_bwd: B=%d HW=%d C=%d", B, HW, C);
  if (B == 0 || HW == 0) return PD_OK;
  if (!dtok || (!dmem_c && !dmempos_c)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_mem_prep_bwd: null pointer");
  dim3 g(grid_rows(B * HW, 4096)), b(256);
  if (c_dtype == PD_BF16) hipLaunchKernelGGL((mem_prep_bwd<bf16_t>), g, b, 0, (hipStream_t)stream_, (const bf16_t *)dmem_c, (const bf16_t *)dmempos_c, dtok, dtok_batch_stride, B, HW, C);
  else hipLaunchKernelGGL((mem_prep_bwd<float>), g, b, 0, (hipStream_t)stream_, (const float *)dmem_c, (const float *)dmempos_c, dtok, dtok_batch_stride, B, HW, C);
  return pd_check_launch("pd_mem_prep_bwd");
}

// ------------------------------------------------------------------------------------------------ grouped copies
// Up to PD_COPY_MAX_SEGS dense byte ranges copied by ONE launch (the table travels as the kernel argument): concatenating slices of several
// parameter tensors — the decoder's per-level key / value weights — was a torch.cat (a 5 us launch) per destination.
struct CopySegs {
  const unsigned char *src[PD_COPY_MAX_SEGS];
  unsigned char *dst[PD_COPY_MAX_SEGS];
  int64_t bytes[PD_COPY_MAX_SEGS];
};
__global__ __launch_bounds__(256) void copy_segments(const CopySegs sg)
{
  const int seg = blockIdx.y;
  const unsigned char *src = sg.src[seg];
  unsigned char *dst = sg.dst[seg];
  const int64_t n = sg.bytes[seg];
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    const int64_t n16 = n >> 4;
    for (int64_t i = tid; i < n16; i += nth) reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
    for (int64_t i = (n16 << 4) + tid; i < n; i += nth) dst[i] = src[i];
  } else {
    for (int64_t i = tid; i < n; i += nth) dst[i] = src[i];
  }
}

extern "C" int pd_copy_segments(const PdCopySeg *segs, int count, void *stream_)
{
  if (count < 0 || count > PD_COPY_MAX_SEGS || (count > 0 && !segs)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_copy_segments: count=%d (<= %d)", count, PD_COPY_MAX_SEGS);
  if (count == 0) return PD_OK;
  CopySegs sg;
  int64_t longest = 0;
  for (int i = 0; i < count; ++i) {
    if (segs[i].bytes < 0 || (segs[i].bytes > 0 && (!segs[i].src || !segs[i].dst))) return pd_set_error(PD_ERR_INVALID_ARG, "pd_copy_segments: segment %d", i);
    sg.src[i] = (const unsigned char *)segs[i].src; sg.dst[i] = (unsigned char *)segs[i].dst; sg.bytes[i] = segs[i].bytes;
    if (segs[i].bytes > longest) longest = segs[i].bytes;
  }
  for (int i = count; i < PD_COPY_MAX_SEGS; ++i) { sg.src[i] = nullptr; sg.dst[i] = nullptr; sg.bytes[i] = 0; }
  int bx = (int)((longest + 256 * 16 * 4 - 1) / (256 * 16 * 4));      // ~4 pieces of 16 bytes per thread of the longest segment
  bx = bx < 1 ? 1 : bx > 64 ? 64 : bx;
  hipLaunchKernelGGL(copy_segments, dim3((unsigned)bx, (unsigned)count), dim3(256), 0, (hipStream_t)stream_, sg);
  return pd_check_launch("pd_copy_segments");
}

extern "C" int pd_attn_mask_u8(const void *logits, int dtype, int rows, int n, uint8_t *mask, void *stream_)
{
  if (rows < 0 || n < 0 || !dt_ok(dtype)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_attn_mask_u8: rows=%d n=%d dtype=%d", rows, n, dtype);
  if (rows == 0 || n == 0) return PD_OK;
  if (!logits || !mask) return pd_set_error(PD_ERR_INVALID_ARG, "pd_attn_mask_u8: null pointer");
  if (dtype == PD_BF16 && !(n & 7) && n <= 256 * 8 * AMV && !((uintptr_t)logits & 15) && !((uintptr_t)mask & 7))
    hipLaunchKernelGGL(attn_mask_u8_bf16x8, dim3(rows), dim3(256), 0, (hipStream_t)stream_, (const bf16_t *)logits, n, mask);
  else if (dtype == PD_BF16) hipLaunchKernelGGL((attn_mask_u8<bf16_t>), dim3(rows), dim3(256), 0, (hipStream_t)stream_, (const bf16_t *)logits, n, mask);
  else hipLaunchKernelGGL((attn_mask_u8<float>), dim3(rows), dim3(256), 0, (hipStream_t)stream_, (const float *)logits, n, mask);
  return pd_check_launch("pd_attn_mask_u8");
}

extern "C" int pd_matcher_point_terms(const void *x, int dtype, int rows, int n, float *x_f32, float *sigmoid_x, float *softplus_sum,
                                      float *sigmoid_sum, void *stream_)
{
  if (rows < 0 || n < 0 || !dt_ok(dtype)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_matcher_point_terms: rows=%d n=%d dtype=%d", rows, n, dtype);
  if (rows == 0) return PD_OK;
  if (!x || !sigmoid_x || !softplus_sum || !sigmoid_sum) return pd_set_error(PD_ERR_INVALID_ARG, "pd_matcher_point_terms: null pointer");
  if (dtype == PD_BF16) hipLaunchKernelGGL((matcher_point_terms<bf16_t>), dim3(rows), dim3(256), 0, (hipStream_t)stream_, (const bf16_t *)x, n, x_f32, sigmoid_x, softplus_sum, sigmoid_sum);
  else hipLaunchKernelGGL((matcher_point_terms<float>), dim3(rows), dim3(256), 0, (hipStream_t)stream_, (const float *)x, n, x_f32, sigmoid_x, softplus_sum, sigmoid_sum);
  return pd_check_launch("pd_matcher_point_terms");
}

extern "C" int pd_msda_prep_fwd(const float *offs, const float *logits, const float *ref, const int64_t *spatial_shapes, float *loc,
                                float *attn, int64_t tokens, int M, int L, int P, int ld_offs, int ld_logits, void *stream_)
{
  if (ld_offs < M * L * P * 2 || ld_logits < M * L * P || (ld_offs & 3) || (ld_logits & 3))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_msda_prep_fwd: row strides %d / %d (>= row length, multiples of 4)", ld_offs, ld_logits);
  if (tokens < 0 || M <= 0 || L <= 0 || P <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_msda_prep_fwd: tokens=%lld M=%d L=%d P=%d", (long long)tokens, M, L, P);
  if (tokens == 0) return PD_OK;
  if (!offs || !logits || !ref || !spatial_shapes || !loc || !attn) return pd_set_error(PD_ERR_INVALID_ARG, "pd_msda_prep_fwd: null pointer");
  const int64_t total = tokens * M;
  const dim3 g((unsigned)((total + 255) / 256)), b(256);
  hipStream_t s = (hipStream_t)stream_;
  const int LP = L * P;
#define LAUNCH(KER) hipLaunchKernelGGL(KER, g, b, 0, s, offs, logits, ref, spatial_shapes, loc, attn, total, M, L, P, ld_offs, ld_logits)
  if ((P & 1) == 0 && LP == 12) LAUNCH(msda_prep_fwd_v<3>);
  else if ((P & 1) == 0 && LP == 16) LAUNCH(msda_prep_fwd_v<4>);
  else if ((P & 1) == 0 && LP == 8) LAUNCH(msda_prep_fwd_v<2>);
  else LAUNCH(msda_prep_fwd);
#undef LAUNCH
  return pd_check_launch("pd_msda_prep_fwd");
}

static int msda_prep_bwd_launch(const float *gloc, const float *gattn, const float *attn, const int64_t *spatial_shapes, float *d_offs,
                                float *d_logits, int64_t tokens, int M, int L, int P, int ld_offs, int ld_logits, void *stream_, float *row_amax)
{
  if (row_amax && (M != 8 || (P & 1) || (L * P != 12 && L * P != 16 && L * P != 8)))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_msda_prep_bwd_amax: row maxima need 8 heads and L P in {8, 12, 16} (the vector kernels)");
  if (ld_offs < M * L * P * 2 || ld_logits < M * L * P || (ld_offs & 3) || (ld_logits & 3))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_msda_prep_bwd: row strides %d / %d (>= row length, multiples of 4)", ld_offs, ld_logits);
  if (tokens < 0 || M <= 0 || L <= 0 || P <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_msda_prep_bwd: tokens=%lld M=%d L=%d P=%d", (long long)tokens, M, L, P);
  if (tokens == 0) return PD_OK;
  if (!gloc || !gattn || !attn || !spatial_shapes || !d_offs || !d_logits) return pd_set_error(PD_ERR_INVALID_ARG, "pd_msda_prep_bwd: null pointer");
  const int64_t total = tokens * M;
  const dim3 g((unsigned)((total + 255) / 256)), b(256);
  hipStream_t s = (hipStream_t)stream_;
  const int LP = L * P;
#define LAUNCH(KER) hipLaunchKernelGGL(KER, g, b, 0, s, gloc, gattn, attn, spatial_shapes, d_offs, d_logits, total, M, L, P, ld_offs, ld_logits, row_amax)
  if ((P & 1) == 0 && LP == 12) LAUNCH(msda_prep_bwd_v<3>);
  else if ((P & 1) == 0 && LP == 16) LAUNCH(msda_prep_bwd_v<4>);
  else if ((P & 1) == 0 && LP == 8) LAUNCH(msda_prep_bwd_v<2>);
  else hipLaunchKernelGGL(msda_prep_bwd, g, b, 0, s, gloc, gattn, attn, spatial_shapes, d_offs, d_logits, total, M, L, P, ld_offs, ld_logits);
#undef LAUNCH
  return pd_check_launch("pd_msda_prep_bwd");
}

extern "C" int pd_msda_prep_bwd(const float *gloc, const float *gattn, const float *attn, const int64_t *spatial_shapes, float *d_offs,
                                float *d_logits, int64_t tokens, int M, int L, int P, int ld_offs, int ld_logits, void *stream_)
{
  return msda_prep_bwd_launch(gloc, gattn, attn, spatial_shapes, d_offs, d_logits, tokens, M, L, P, ld_offs, ld_logits, stream_, nullptr);
}

extern "C" int pd_msda_prep_bwd_amax(const float *gloc, const float *gattn, const float *attn, const int64_t *spatial_shapes, float *d_offs,
                                     float *d_logits, float *row_amax, int64_t tokens, int M, int L, int P, int ld_offs, int ld_logits, void *stream_)
{
  return msda_prep_bwd_launch(gloc, gattn, attn, spatial_shapes, d_offs, d_logits, tokens, M, L, P, ld_offs, ld_logits, stream_, row_amax);
}

extern "C" int pd_point_sample_nhwc_f32(const float *in, const float *coords, float *out, int B, int H, int W, int C, int P,
                                        void *stream_)
{
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || P < 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_point_sample_nhwc_f32: B=%d H=%d W=%d C=%d P=%d", B, H, W, C, P);
  if (B == 0 || P == 0) return PD_OK;
  if (!in || !coords || !out) return pd_set_error(PD_ERR_INVALID_ARG, "pd_point_sample_nhwc_f32: null pointer");
  const int64_t total = (int64_t)B * P;
  const unsigned grid = (unsigned)((total + 3) / 4 < 16384 ? (total + 3) / 4 : 16384);
  hipLaunchKernelGGL(point_sample_nhwc<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream_, in, coords, out, B, H, W, C, P);
  return pd_check_launch("pd_point_sample_nhwc_f32");
}

extern "C" int pd_point_sample_nhwc_f32_bf16(const float *in, const float *coords, void *out, int B, int H, int W, int C, int P,
                                             void *stream_)
{
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || P < 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_point_sample_nhwc_f32_bf16: B=%d H=%d W=%d C=%d P=%d", B, H, W, C, P);
  if (B == 0 || P == 0) return PD_OK;
  if (!in || !coords || !out) return pd_set_error(PD_ERR_INVALID_ARG, "pd_point_sample_nhwc_f32_bf16: null pointer");
  const int64_t total = (int64_t)B * P;
  const unsigned grid = (unsigned)((total + 3) / 4 < 16384 ? (total + 3) / 4 : 16384);
  hipLaunchKernelGGL(point_sample_nhwc<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream_, in, coords, (bf16_t *)out, B, H, W, C, P);
  return pd_check_launch("pd_point_sample_nhwc_f32_bf16");
}

static int upsample_add_launch(const float *lo, int64_t lo_bs, const float *cur, float *y, int B, int h, int w, int H, int W, int C, void *stream_, float *amax)
{
  if (lo_bs == 0) lo_bs = (int64_t)h * w * C;
  if (lo_bs < (int64_t)h * w * C || (lo_bs & 3)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_upsample_add_nhwc_f32: lo_batch_stride %lld (>= h w C, a multiple of 4)", (long long)lo_bs);
  if (amax && C != 256) return pd_set_error(PD_ERR_INVALID_ARG, "pd_upsample_add_amax_nhwc_f32: pixel maxima need C == 256");
  if (B < 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_upsample_add_nhwc_f32: bad sizes");
  if (B == 0) return PD_OK;
  if (!lo || !cur || !y) return pd_set_error(PD_ERR_INVALID_ARG, "pd_upsample_add_nhwc_f32: null pointer");
  const int64_t total = (int64_t)B * H * W * (C / 4);
  const unsigned grid = (unsigned)((total + 255) / 256 < 32768 ? (total + 255) / 256 : 32768);
  hipLaunchKernelGGL(upsample_add_nhwc, dim3(grid), dim3(256), 0, (hipStream_t)stream_, lo, cur, y, B, h, w, H, W, C / 4,
                     (float)h / (float)H, (float)w / (float)W, amax, lo_bs / 4);
  return pd_check_launch("pd_upsample_add_nhwc_f32");
}

extern "C" int pd_upsample_add_nhwc_f32(const float *lo, int64_t lo_batch_stride, const float *cur, float *y, int B, int h, int w, int H, int W, int C,
                                        void *stream_)
{
  return upsample_add_launch(lo, lo_batch_stride, cur, y, B, h, w, H, W, C, stream_, nullptr);
}

extern "C" int pd_upsample_add_amax_nhwc_f32(const float *lo, int64_t lo_batch_stride, const float *cur, float *y, float *amax, int B, int h, int w, int H,
                                             int W, int C, void *stream_)
{
  return upsample_add_launch(lo, lo_batch_stride, cur, y, B, h, w, H, W, C, stream_, amax);
}

extern "C" int pd_upsample2x_bwd_nhwc_f32(const float *dy, float *dlo, int B, int h, int w, int C, void *stream_)
{
  if (B < 0 || h <= 0 || w <= 0 || C <= 0 || (C & 3)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_upsample2x_bwd_nhwc_f32: bad sizes");
  if (B == 0) return PD_OK;
  if (!dy || !dlo) return pd_set_error(PD_ERR_INVALID_ARG, "pd_upsample2x_bwd_nhwc_f32: null pointer");
  const int64_t total = (int64_t)B * h * w * (C / 4);
  const unsigned grid = (unsigned)((total + 255) / 256 < 32768 ? (total + 255) / 256 : 32768);
  hipLaunchKernelGGL(upsample2x_bwd_nhwc, dim3(grid), dim3(256), 0, (hipStream_t)stream_, dy, dlo, B, h, w, C / 4);
  return pd_check_launch("pd_upsample2x_bwd_nhwc_f32");
}

extern "C" int pd_point_sample_planar_f32(const float *in, const float *coords, float *out, int N, int C, int H, int W, int P, void *stream_)
{
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || P < 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_point_sample_planar_f32: N=%d C=%d H=%d W=%d P=%d", N, C, H, W, P);
  if ((int64_t)N * P == 0) return PD_OK;
  if (!in || !coords || !out) return pd_set_error(PD_ERR_INVALID_ARG, "pd_point_sample_planar_f32: null pointer");
  const int tx = (W + PST - 1) / PST, ty = (H + PST - 1) / PST;
  if (C == 1 && tx * ty <= 1 && N <= 65535) {             // one-tile maps only: with 9 tiles per 256^2 map the scan of all points by every
                                                          // tile costs more than the gathers it saves (112 vs 57 us at 80 x 37 632 points)
    hipLaunchKernelGGL(point_sample_tiled_fwd, dim3(tx * ty, N), dim3(256), (PST + 1) * (PST + 1) * sizeof(float), (hipStream_t)stream_, in,
                       coords, out, H, W, P, tx);
    return pd_check_launch("pd_point_sample_planar_f32");
  }
  const int64_t blocks = ((int64_t)N * P + 255) / 256;
  hipLaunchKernelGGL(point_sample_planar_fwd, dim3((unsigned)(blocks > 65536 ? 65536 : blocks)), dim3(256), 0, (hipStream_t)stream_, in, coords, out,
                     N, C, H, W, P);
  return pd_check_launch("pd_point_sample_planar_f32");
}

extern "C" int pd_point_sample_planar_bwd_f32(const float *grad_out, const float *coords, float *grad_in, int N, int C, int H, int W, int P,
                                              void *stream_)
{
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || P < 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_point_sample_planar_bwd_f32: N=%d C=%d H=%d W=%d P=%d", N, C, H, W, P);
  if ((int64_t)N * P == 0) return PD_OK;
  if (!grad_out || !coords || !grad_in) return pd_set_error(PD_ERR_INVALID_ARG, "pd_point_sample_planar_bwd_f32: null pointer");
  const int tx = (W + PST - 1) / PST, ty = (H + PST - 1) / PST;
  if (C == 1 && tx * ty <= 16 && N <= 65535) {            // writes EVERY element of grad_in (no zero-fill needed, harmless if done)
    hipLaunchKernelGGL(point_sample_tiled_bwd, dim3(tx * ty, N), dim3(256), PST * PST * sizeof(float), (hipStream_t)stream_, grad_out, coords,
                       grad_in, H, W, P, tx);
    return pd_check_launch("pd_point_sample_planar_bwd_f32");
  }
  const int64_t blocks = ((int64_t)N * P + 255) / 256;
  hipLaunchKernelGGL(point_sample_planar_bwd, dim3((unsigned)(blocks > 65536 ? 65536 : blocks)), dim3(256), 0, (hipStream_t)stream_, grad_out, coords,
                     grad_in, N, C, H, W, P);
  return pd_check_launch("pd_point_sample_planar_bwd_f32");
}

// mirrors the dispatch of pd_point_sample_planar_bwd_f32 EXACTLY (the tiled kernel writes every element, the scatter kernel accumulates):
// with N maps > 65 535 the scatter kernel runs whatever the tile count
extern "C" int pd_point_sample_planar_bwd_needs_zero_n(int N, int C, int H, int W)
{
  return !(C == 1 && ((W + PST - 1) / PST) * ((H + PST - 1) / PST) <= 16 && N <= 65535);
}
extern "C" int pd_point_sample_planar_bwd_needs_zero(int C, int H, int W) { return pd_point_sample_planar_bwd_needs_zero_n(1, C, H, W); }

extern "C" int pd_transpose_batched_f32(const PdTransposeProblem *problems, int count, void *stream_)
{
  if (count < 0 || count > PD_TRANSPOSE_MAX || (count && !problems)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_transpose_batched_f32: count=%d (0..%d)", count, PD_TRANSPOSE_MAX);
  TrBatch tb;
  memset(&tb, 0, sizeof(tb));
  int tiles = 0;
  for (int i = 0; i < count; ++i) {
    const PdTransposeProblem &p = problems[i];
    if (p.batch < 0 || p.rows < 0 || p.cols < 0 || p.src_row_stride < p.cols) return pd_set_error(PD_ERR_INVALID_ARG, "pd_transpose_batched_f32: problem %d", i);
    if (p.batch && p.rows && p.cols && (!p.src || !p.dst)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_transpose_batched_f32: null pointer");
    tb.p[i] = p;
    tb.first_tile[i] = tiles;
    tiles += p.batch * ((p.rows + 31) / 32) * ((p.cols + 31) / 32);
  }
  tb.first_tile[count] = tiles;
  tb.count = count;
  if (tiles == 0) return PD_OK;
  hipLaunchKernelGGL(transpose_batched_f32, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream_, tb);
  return pd_check_launch("pd_transpose_batched_f32");
}

extern "C" int pd_resize_bilinear_nhwc_f32(const float *x, int B, int H, int W, int C, const int *heights, const int *widths, void *const *outs,
                                           int count, int out_dtype, void *stream_)
{
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || count < 0 || count > PD_RESIZE_MAX || (out_dtype != PD_F32 && out_dtype != PD_BF16))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_resize_bilinear_nhwc_f32: B=%d H=%d W=%d C=%d count=%d dtype=%d", B, H, W, C, count, out_dtype);
  if (B == 0 || count == 0) return PD_OK;
  if (!x || !heights || !widths || !outs) return pd_set_error(PD_ERR_INVALID_ARG, "pd_resize_bilinear_nhwc_f32: null pointer");
  ResizeLevels lv;
  memset(&lv, 0, sizeof(lv));
  int64_t total = 0;
  for (int i = 0; i < count; ++i) {
    if (heights[i] <= 0 || widths[i] <= 0 || !outs[i]) return pd_set_error(PD_ERR_INVALID_ARG, "pd_resize_bilinear_nhwc_f32: level %d", i);
    lv.out[i] = outs[i]; lv.h[i] = heights[i]; lv.w[i] = widths[i]; lv.first[i] = total;
    total += (int64_t)B * heights[i] * widths[i] * (C / 4);
  }
  lv.first[count] = total; lv.count = count;
  const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  if (out_dtype == PD_BF16) hipLaunchKernelGGL(resize_bilinear_nhwc_multi<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream_, x, B, H, W, C / 4, lv);
  else hipLaunchKernelGGL(resize_bilinear_nhwc_multi<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream_, x, B, H, W, C / 4, lv);
  return pd_check_launch("pd_resize_bilinear_nhwc_f32");
}

// ------------------------------------------------------------------------------------------------ skinny linear (class head)
// y = x w^T + b for K <= 8 output columns (the decoder's class head: [heads B Q = 2 000, 256] x [K + 1 = 2, 256]^T, reference
// mask2former_transformer_decoder.py:223, 446).  The library serves the forward in 5 us and the weight gradient [2, 2000] x [2000, 256] in
// 41 us; here a wavefront per row forward, and ONE backward pass per 16 rows that forms d x (added to the mask-embedding MLP's d x, which
// shares the input), and per-workgroup partial d w / d b that skinny_linear_reduce adds in workgroup order.
namespace {
constexpr int SK_MAXK = 8, SK_RB = 32;
__device__ __forceinline__ float4 ldw4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 ldw4(const bf16_t *p) { return ld4(p); }
__device__ __forceinline__ float ldw1(const float *p) { return *p; }
__device__ __forceinline__ float ldw1(const bf16_t *p) { return bf2f(*p); }
__device__ __forceinline__ void stw1(float *p, float v) { *p = v; }
__device__ __forceinline__ void stw1(bf16_t *p, float v) { *p = f2bf(v); }

template <typename WT>
__global__ __launch_bounds__(256) void skinny_linear_fwd(const bf16_t *__restrict__ x, const WT *__restrict__ w, const WT *__restrict__ b,
                                                         float *__restrict__ y, int R, int C, int K)
{
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  float acc[SK_MAXK];
#pragma unroll
  for (int k = 0; k < SK_MAXK; ++k) acc[k] = 0.f;
  for (int c = 4 * lane; c < C; c += 256) {
    const float4 xv = ld4(x + (int64_t)row * C + c);
#pragma unroll
    for (int k = 0; k < SK_MAXK; ++k)
      if (k < K) {
        const float4 wv = ldw4(w + (int64_t)k * C + c);
        acc[k] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
      }
  }
#pragma unroll
  for (int k = 0; k < SK_MAXK; ++k)
    if (k < K) {
      const float v = wave_sum(acc[k]);
      if (lane == 0) y[(int64_t)row * K + k] = v + (b ? ldw1(b + k) : 0.f);
    }
}

// partial [workgroups][K][C + 1]: column C holds the bias partial
template <typename WT, typename DXT>
__global__ __launch_bounds__(256) void skinny_linear_bwd(const bf16_t *__restrict__ x, const WT *__restrict__ w, const float *__restrict__ dy,
                                                         const bf16_t *__restrict__ dx_in, DXT *__restrict__ dx_out, float *__restrict__ partial,
                                                         int R, int C, int K)
{
  __shared__ float dys[SK_RB][SK_MAXK];
  const int r0 = blockIdx.x * SK_RB, nr = min(SK_RB, R - r0);
  if (threadIdx.x < SK_RB * SK_MAXK) {
    const int r = threadIdx.x / SK_MAXK, k = threadIdx.x % SK_MAXK;
    dys[r][k] = (r < nr && k < K) ? dy[(int64_t)(r0 + r) * K + k] : 0.f;
  }
  __syncthreads();
  float *pw = partial + (int64_t)blockIdx.x * K * (C + 1);
  for (int c = threadIdx.x; c < C; c += 256) {
    float wk[SK_MAXK], dw[SK_MAXK];
#pragma unroll
    for (int k = 0; k < SK_MAXK; ++k) wk[k] = k < K ? ldw1(w + (int64_t)k * C + c) : 0.f, dw[k] = 0.f;
    for (int r = 0; r < nr; ++r) {
      const int64_t at = (int64_t)(r0 + r) * C + c;
      const float xv = bf2f(x[at]);
      float d = dx_in ? bf2f(dx_in[at]) : 0.f;
#pragma unroll
      for (int k = 0; k < SK_MAXK; ++k) d += dys[r][k] * wk[k], dw[k] += dys[r][k] * xv;
      if (dx_out) stw1(dx_out + at, d);
    }
#pragma unroll
    for (int k = 0; k < SK_MAXK; ++k)
      if (k < K) pw[k * (C + 1) + c] = dw[k];
  }
  if (threadIdx.x < K) {
    float s = 0.f;
    for (int r = 0; r < nr; ++r) s += dys[r][threadIdx.x];
    pw[threadIdx.x * (C + 1) + C] = s;
  }
}

template <typename WT, typename BT>
__global__ __launch_bounds__(256) void skinny_linear_reduce(const float *__restrict__ partial, WT *__restrict__ dw, BT *__restrict__ db, int groups, int C, int K)
{
  const int i = blockIdx.x * 256 + threadIdx.x, n = K * (C + 1);
  if (i >= n) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int g = 0;
  for (; g + 3 < groups; g += 4) {
    s0 += partial[(int64_t)g * n + i], s1 += partial[(int64_t)(g + 1) * n + i];
    s2 += partial[(int64_t)(g + 2) * n + i], s3 += partial[(int64_t)(g + 3) * n + i];
  }
  for (; g < groups; ++g) s0 += partial[(int64_t)g * n + i];
  const float s = (s0 + s1) + (s2 + s3);
  const int k = i / (C + 1), c = i - k * (C + 1);
  if (c < C) stw1(dw + (int64_t)k * C + c, s);
  else if (db) stw1(db + k, s);
}
}  // namespace

extern "C" int pd_skinny_linear_fwd(const void *x_bf16, const void *w, const void *b, int wb_dtype, float *y, int R, int C, int K, void *stream_)
{
  if (R < 0 || C <= 0 || (C & 3) || K <= 0 || K > SK_MAXK || !dt_ok(wb_dtype))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_skinny_linear_fwd: R=%d C=%d (a multiple of 4) K=%d (<= %d) dtype=%d", R, C, K, SK_MAXK, wb_dtype);
  if (R == 0) return PD_OK;
  if (!x_bf16 || !w || !y || (((uintptr_t)x_bf16 | (uintptr_t)w) & 7)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_skinny_linear_fwd: null / misaligned pointer");
  const dim3 g((unsigned)((R + 3) / 4)), t(256);
  if (wb_dtype == PD_BF16)
    hipLaunchKernelGGL(skinny_linear_fwd<bf16_t>, g, t, 0, (hipStream_t)stream_, (const bf16_t *)x_bf16, (const bf16_t *)w, (const bf16_t *)b, y, R, C, K);
  else
    hipLaunchKernelGGL(skinny_linear_fwd<float>, g, t, 0, (hipStream_t)stream_, (const bf16_t *)x_bf16, (const float *)w, (const float *)b, y, R, C, K);
  return pd_check_launch("pd_skinny_linear_fwd");
}

extern "C" int64_t pd_skinny_linear_partial_floats(int R, int C, int K) { return (int64_t)((R + SK_RB - 1) / SK_RB) * K * (C + 1); }

extern "C" int pd_skinny_linear_bwd(const void *x_bf16, const void *w, int w_dtype, const float *dy, const void *dx_in_bf16, void *dx_out, int dx_dtype,
                                    float *partial, void *dw, void *db, int b_dtype, int R, int C, int K, void *stream_)
{
  if (R < 0 || C <= 0 || K <= 0 || K > SK_MAXK || !dt_ok(w_dtype) || !dt_ok(dx_dtype) || !dt_ok(b_dtype))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_skinny_linear_bwd: R=%d C=%d K=%d (<= %d) dtypes %d %d %d", R, C, K, SK_MAXK, w_dtype, dx_dtype, b_dtype);
  if (!x_bf16 || !w || !dy || !partial || !dw) return pd_set_error(PD_ERR_INVALID_ARG, "pd_skinny_linear_bwd: null pointer");
  const int groups = (R + SK_RB - 1) / SK_RB;
  hipStream_t s = (hipStream_t)stream_;
  if (groups) {
    const dim3 g((unsigned)groups), t(256);
#define PD_SK(WT, DXT) hipLaunchKernelGGL((skinny_linear_bwd<WT, DXT>), g, t, 0, s, (const bf16_t *)x_bf16, (const WT *)w, dy, (const bf16_t *)dx_in_bf16, \
                                          (DXT *)dx_out, partial, R, C, K)
    if (w_dtype == PD_BF16) { if (dx_dtype == PD_BF16) PD_SK(bf16_t, bf16_t); else PD_SK(bf16_t, float); }
    else { if (dx_dtype == PD_BF16) PD_SK(float, bf16_t); else PD_SK(float, float); }
#undef PD_SK
  }
  const dim3 g2((unsigned)((K * (C + 1) + 255) / 256)), t2(256);
#define PD_SKR(WT, BT) hipLaunchKernelGGL((skinny_linear_reduce<WT, BT>), g2, t2, 0, s, partial, (WT *)dw, (BT *)db, groups, C, K)
  if (w_dtype == PD_BF16) { if (b_dtype == PD_BF16) PD_SKR(bf16_t, bf16_t); else PD_SKR(bf16_t, float); }
  else { if (b_dtype == PD_BF16) PD_SKR(float, bf16_t); else PD_SKR(float, float); }
#undef PD_SKR
  return pd_check_launch("pd_skinny_linear_bwd");
}

extern "C" int pd_add_rows_amax_f32(const float *a, const float *b, float *q, float *a_copy, float *a_amax, float *q_amax, int rows, int cols,
                                    void *stream_)
{
  if (rows < 0 || cols <= 0 || (cols & 3)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_add_rows_amax_f32: rows=%d cols=%d (a multiple of 4)", rows, cols);
  if (rows == 0) return PD_OK;
  if (!a || !b || !q || !a_amax || !q_amax || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)q | (uintptr_t)a_copy) & 15))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_add_rows_amax_f32: null / misaligned pointer");
  hipLaunchKernelGGL(add_rows_amax_f32, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, a, b, q, a_copy, a_amax, q_amax, rows, cols);
  return pd_check_launch("pd_add_rows_amax_f32");
}

extern "C" int pd_sum3_sum2_f32(const float *a, const float *b, const float *c, const float *d, float *out1, float *out2, int64_t n, void *stream_)
{
  if (n < 0 || (n & 3)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_sum3_sum2_f32: n=%lld (a multiple of 4)", (long long)n);
  if (n == 0) return PD_OK;
  if (!a || !b || !c || !d || !out1 || !out2 || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)out1 | (uintptr_t)out2) & 15))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_sum3_sum2_f32: null / misaligned pointer");
  const int64_t n4 = n / 4;
  const unsigned grid = (unsigned)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
  hipLaunchKernelGGL(sum3_sum2_f32, dim3(grid), dim3(256), 0, (hipStream_t)stream_, (const float4 *)a, (const float4 *)b, (const float4 *)c,
                     (const float4 *)d, (float4 *)out1, (float4 *)out2, n4);
  return pd_check_launch("pd_sum3_sum2_f32");
}
