// Fused bandwidth-bound kernels (gfx950): affine+residual+ReLU epilogue of the frozen-BN backbone convolutions
// (bf16 NHWC, 16-byte lanes) and the multi-tensor gradient gather with fused sum of squares.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "pd_common.h"
#include "pd_fused.h"
#include "pd_msda.h"

namespace {

typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f2bf(float f)
{
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                                  // round to nearest even
  return (unsigned short)(u >> 16);
}

// ------------------------------------------------------------------------------------------------ stem max pooling
// 3 x 3, stride 2, pad 1 max pooling of the R50 stem on bf16 NHWC maps (reference: detectron2 BasicStem, F.max_pool2d): the forward
// also leaves the window position of the maximum (0..8, the FIRST maximum in row-major window order like ATen's kernel: after a ReLU
// most windows hold ties at 0) as one byte per output element, and the backward is a GATHER — every input pixel looks at the <= 4
// windows that cover it — instead of ATen's zero-fill + scatter through 8-byte indices (67 MB of indices at 2 x 64 x 512^2).
__global__ __launch_bounds__(256) void maxpool3s2_fwd(const u16x8 *__restrict__ x, u16x8 *__restrict__ y, unsigned long long *__restrict__ arg,
                                                      int B, int H, int W, int OH, int OW, int c8)
{
  const int64_t total = (int64_t)B * OH * OW * c8, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % c8);
    int64_t t = i / c8;
    const int ox = (int)(t % OW); t /= OW;
    const int oy = (int)(t % OH);
    const int b = (int)(t / OH);
    float best[8];
    unsigned pos[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; pos[e] = 0; }
    bool first = true;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int iy = 2 * oy - 1 + dy, ix = 2 * ox - 1 + dx;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        const u16x8 v = x[(((int64_t)b * H + iy) * W + ix) * c8 + c];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = bf2f(v[e]);
          if (first || f > best[e] || f != f) { best[e] = f; pos[e] = dy * 3 + dx; }       // ATen: (val > max) || isnan(val); first valid tap seeds
        }
        first = false;
      }
    u16x8 o;
    unsigned long long a = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { o[e] = f2bf(best[e]); a |= (unsigned long long)pos[e] << (8 * e); }
    y[i] = o;
    arg[i] = a;
  }
}

__global__ __launch_bounds__(256) void maxpool3s2_bwd(const u16x8 *__restrict__ dy, const unsigned long long *__restrict__ arg, u16x8 *__restrict__ dx,
                                                      int B, int H, int W, int OH, int OW, int c8)
{
  const int64_t total = (int64_t)B * H * W * c8, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % c8);
    int64_t t = i / c8;
    const int ix = (int)(t % W); t /= W;
    const int iy = (int)(t % H);
    const int b = (int)(t / H);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // windows oy with 2 oy - 1 <= iy <= 2 oy + 1
    const int oy0 = iy >> 1, oy1 = (iy + 1) >> 1, ox0 = ix >> 1, ox1 = (ix + 1) >> 1;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int oy = a ? oy1 : oy0;
      if ((a && oy1 == oy0) || oy >= OH) continue;
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const int ox = bb ? ox1 : ox0;
        if ((bb && ox1 == ox0) || ox >= OW) continue;
        const unsigned want = (unsigned)((iy - (2 * oy - 1)) * 3 + (ix - (2 * ox - 1)));
        const int64_t o = (((int64_t)b * OH + oy) * OW + ox) * c8 + c;
        const unsigned long long ar = arg[o];
        const u16x8 g = dy[o];
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (((ar >> (8 * e)) & 0xffu) == want) acc[e] += bf2f(g[e]);
      }
    }
    u16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = f2bf(acc[e]);
    dx[i] = r;
  }
}

template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void affine_act_fwd(const u16x8 *__restrict__ x, const u16x8 *__restrict__ res,
                                                       const float *__restrict__ scale, const float *__restrict__ bias,
                                                       u16x8 *__restrict__ y, int64_t n8, int c8)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const int c = (int)(i % c8) * 8;
    const u16x8 v = x[i];
    u16x8 r;
    if (RES) r = res[i];
    const float4 s0 = *reinterpret_cast<const float4 *>(scale + c), s1 = *reinterpret_cast<const float4 *>(scale + c + 4);
    const float4 b0 = *reinterpret_cast<const float4 *>(bias + c), b1 = *reinterpret_cast<const float4 *>(bias + c + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float bi[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    u16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      // fp32 inside, one rounding at the end (the unfused half-precision chain rounds after every op)
      float f = bf2f(v[k]) * sc[k] + bi[k];
      if (RES) f += bf2f(r[k]);
      if (RELU) f = fmaxf(f, 0.f);
      o[k] = f2bf(f);
    }
    y[i] = o;
  }
}

template <bool RES, bool RELU, bool TWO>
__global__ __launch_bounds__(256) void affine_act_bwd(const u16x8 *__restrict__ gy, const u16x8 *__restrict__ gy2, const u16x8 *__restrict__ y,
                                                       const float *__restrict__ scale, u16x8 *__restrict__ gx,
                                                       u16x8 *__restrict__ gres, int64_t n8, int c8)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const int c = (int)(i % c8) * 8;
    u16x8 g = gy[i];
    if (TWO) {                                            // the output fed two consumers: their gradients are summed here (bf16 sum,
      const u16x8 h = gy2[i];                             // rounded like the separate add kernel autograd would have launched)
#pragma unroll
      for (int k = 0; k < 8; ++k) g[k] = f2bf(bf2f(g[k]) + bf2f(h[k]));
    }
    u16x8 yy;
    if (RELU) yy = y[i];
    const float4 s0 = *reinterpret_cast<const float4 *>(scale + c), s1 = *reinterpret_cast<const float4 *>(scale + c + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    u16x8 o, orr;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool on = !RELU || ((yy[k] & 0x7fff) != 0 && !(yy[k] & 0x8000));   // y > 0
      const unsigned short gk = on ? g[k] : (unsigned short)0;
      if (RES) orr[k] = gk;
      o[k] = f2bf(bf2f(gk) * sc[k]);
    }
    gx[i] = o;
    if (RES) gres[i] = orr;
  }
}

__global__ __launch_bounds__(256) void multi_gather_sumsq(const int64_t *__restrict__ src_ptrs, const int32_t *__restrict__ is_bf16,
                                                           const int32_t *__restrict__ blk_tensor, const int64_t *__restrict__ blk_start,
                                                           const int64_t *__restrict__ blk_dst, const int32_t *__restrict__ blk_len,
                                                           float *__restrict__ dst, double *__restrict__ sumsq, int block_begin)
{
  const int b = block_begin + blockIdx.x;
  const int t = blk_tensor[b];
  const int64_t start = blk_start[b];
  const int len = blk_len[b];
  float *d = dst + blk_dst[b];
  const int64_t base = src_ptrs[t];
  float acc = 0.f;
  if (base == 0) {
    for (int i = threadIdx.x; i < len; i += 256) d[i] = 0.f;
  } else if (is_bf16[t]) {
    const unsigned short *s = pd_as_global(reinterpret_cast<const unsigned short *>(base)) + start;   // (pd_common.h: a pointer read from memory would be FLAT)
    if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {               // eight bf16 per lane-step: one 16-byte load, two 16-byte stores
      const int l8 = len >> 3;
      for (int i = threadIdx.x; i < l8; i += 256) {
        const uint4 u = reinterpret_cast<const uint4 *>(s)[i];
        const float4 a = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
        const float4 c = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u));
        reinterpret_cast<float4 *>(d)[2 * i] = a;
        reinterpret_cast<float4 *>(d)[2 * i + 1] = c;
        acc += (a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w) + (c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w);
      }
      for (int i = (l8 << 3) + threadIdx.x; i < len; i += 256) { const float v = bf2f(s[i]); d[i] = v; acc += v * v; }
    } else {
      for (int i = threadIdx.x; i < len; i += 256) { const float v = bf2f(s[i]); d[i] = v; acc += v * v; }
    }
  } else {
    const float *s = pd_as_global(reinterpret_cast<const float *>(base)) + start;
    if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
      const int l4 = len >> 2;
      for (int i = threadIdx.x; i < l4; i += 256) {
        const float4 v = reinterpret_cast<const float4 *>(s)[i];
        reinterpret_cast<float4 *>(d)[i] = v;
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      }
      for (int i = (l4 << 2) + threadIdx.x; i < len; i += 256) { const float v = s[i]; d[i] = v; acc += v * v; }
    } else {
      for (int i = threadIdx.x; i < len; i += 256) { const float v = s[i]; d[i] = v; acc += v * v; }
    }
  }
  if (sumsq) {
    double a = (double)acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(sumsq, part[0] + part[1] + part[2] + part[3]);
  }
}

// ---------------------------------------------------------------------------------------------- NHWC GroupNorm pieces
// GroupNorm over channels-last fp32 maps [N, P pixels, C] is built from three streaming kernels with per-(image,
// channel) coefficients; the O(N*C) algebra between them (group means, rstd, coefficients) is done by the caller.
//   nc_sums<0>   S0[n,c] = sum_p x,            S1[n,c] = sum_p x*x                      (fp64 accumulation across blocks)
//   nc_sums<1>   S0[n,c] = sum_p dy*(x*a+b),   S1[n,c] = sum_p dy     (dy masked by y > 0 when relu)
//   nc_affine    y  = x*a[n,c] + b[n,c] (ReLU)
//   nc_affine2   dx = dy(masked)*a[n,c] + x*p[n,c] + r[n,c]
// Thread t owns channel quad t % (C/4) of pixels (t / (C/4)) + k*(256/(C/4)): one wave instruction reads whole pixels.
template <int MODE>
__global__ __launch_bounds__(256) void nc_sums(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ y,
                                                const float *__restrict__ a, const float *__restrict__ b, double *__restrict__ out,
                                                int P, int C, int pix_per_block, int relu)
{
  __shared__ float red[2][256][4];
  const int n = blockIdx.y, cq = C >> 2, q = threadIdx.x % cq, po = threadIdx.x / cq, pstride = 256 / cq;
  const int p0 = blockIdx.x * pix_per_block, p1 = min(P, p0 + pix_per_block);
  const int64_t base = (int64_t)n * P * C + q * 4;
  float4 s0 = make_float4(0, 0, 0, 0), s1 = s0, av = s0, bv = s0;
  if (MODE == 1) { av = *reinterpret_cast<const float4 *>(a + n * C + q * 4); bv = *reinterpret_cast<const float4 *>(b + n * C + q * 4); }
#pragma unroll 4
  for (int p = p0 + po; p < p1; p += pstride) {
    const float4 xv = *reinterpret_cast<const float4 *>(x + base + (int64_t)p * C);
    if (MODE == 0) {
      s0.x += xv.x; s0.y += xv.y; s0.z += xv.z; s0.w += xv.w;
      s1.x += xv.x * xv.x; s1.y += xv.y * xv.y; s1.z += xv.z * xv.z; s1.w += xv.w * xv.w;
    } else {
      float4 g = *reinterpret_cast<const float4 *>(dy + base + (int64_t)p * C);
      if (relu) {
        const float4 yv = *reinterpret_cast<const float4 *>(y + base + (int64_t)p * C);
        g.x = yv.x > 0 ? g.x : 0; g.y = yv.y > 0 ? g.y : 0; g.z = yv.z > 0 ? g.z : 0; g.w = yv.w > 0 ? g.w : 0;
      }
      s0.x += g.x * (xv.x * av.x + bv.x); s0.y += g.y * (xv.y * av.y + bv.y);
      s0.z += g.z * (xv.z * av.z + bv.z); s0.w += g.w * (xv.w * av.w + bv.w);
      s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
    }
  }
  red[0][threadIdx.x][0] = s0.x; red[0][threadIdx.x][1] = s0.y; red[0][threadIdx.x][2] = s0.z; red[0][threadIdx.x][3] = s0.w;
  red[1][threadIdx.x][0] = s1.x; red[1][threadIdx.x][1] = s1.y; red[1][threadIdx.x][2] = s1.z; red[1][threadIdx.x][3] = s1.w;
  __syncthreads();
  if (threadIdx.x < C) {                      // one thread per channel sums the pixel-offset copies
    const int c = threadIdx.x, qq = c >> 2, e = c & 3;
    double t0 = 0, t1 = 0;
    for (int k = 0; k < pstride; ++k) { t0 += red[0][k * cq + qq][e]; t1 += red[1][k * cq + qq][e]; }
    unsafeAtomicAdd(out + ((int64_t)n * C + c) * 2, t0);
    unsafeAtomicAdd(out + ((int64_t)n * C + c) * 2 + 1, t1);
  }
}

// amax (nullable; C == 256 only: the 64 lanes of a wavefront then hold exactly one pixel): absolute maximum over the channels of every
// output pixel — the row-scaling input of pd_gemm_tn_f16x2 / pd_conv3x3_nhwc_f16x2 for the convolution that reads this map
__device__ __forceinline__ void pixel_amax_store(float4 v, float *__restrict__ amax, int64_t i)
{
  float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
#pragma unroll
  for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) amax[i >> 6] = m;
}

template <bool RELU>
__global__ __launch_bounds__(256) void nc_affine(const float4 *__restrict__ x, const float *__restrict__ a, const float *__restrict__ b,
                                                  float4 *__restrict__ y, int64_t n4, int P, int C, float *__restrict__ amax)
{
  const int cq = C >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, per_img = (int64_t)P * cq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int n = (int)(i / per_img), q = (int)(i % cq);
    const float4 av = *reinterpret_cast<const float4 *>(a + n * C + q * 4), bv = *reinterpret_cast<const float4 *>(b + n * C + q * 4);
    float4 v = x[i];
    v.x = v.x * av.x + bv.x; v.y = v.y * av.y + bv.y; v.z = v.z * av.z + bv.z; v.w = v.w * av.w + bv.w;
    if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    y[i] = v;
    if (amax) pixel_amax_store(v, amax, i);
  }
}

template <bool RELU>
__global__ __launch_bounds__(256) void nc_affine2(const float4 *__restrict__ dy, const float4 *__restrict__ x, const float4 *__restrict__ y,
                                                   const float *__restrict__ a, const float *__restrict__ pc, const float *__restrict__ rc,
                                                   float4 *__restrict__ dx, int64_t n4, int P, int C, float *__restrict__ amax)
{
  const int cq = C >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, per_img = (int64_t)P * cq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int n = (int)(i / per_img), q = (int)(i % cq);
    const float4 av = *reinterpret_cast<const float4 *>(a + n * C + q * 4), pv = *reinterpret_cast<const float4 *>(pc + n * C + q * 4),
                 rv = *reinterpret_cast<const float4 *>(rc + n * C + q * 4);
    float4 g = dy[i];
    const float4 xv = x[i];
    if (RELU) {
      const float4 yv = y[i];
      g.x = yv.x > 0 ? g.x : 0; g.y = yv.y > 0 ? g.y : 0; g.z = yv.z > 0 ? g.z : 0; g.w = yv.w > 0 ? g.w : 0;
    }
    const float4 o = make_float4(g.x * av.x + xv.x * pv.x + rv.x, g.y * av.y + xv.y * pv.y + rv.y, g.z * av.z + xv.z * pv.z + rv.z,
                                 g.w * av.w + xv.w * pv.w + rv.w);
    dx[i] = o;
    if (amax) pixel_amax_store(o, amax, i);
  }
}

inline int grid_for(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }


// group statistics -> per-(n, c) coefficients; one thread per (n, c), each re-reads its group's <= 64 channel sums
__global__ __launch_bounds__(256) void gn_coeffs_fwd(const double *__restrict__ sums, const float *__restrict__ weight,
                                                     const float *__restrict__ bias, int N, int C, int G, double m, double eps,
                                                     float *__restrict__ a, float *__restrict__ b, float *__restrict__ mean_c,
                                                     float *__restrict__ rstd_c, float *__restrict__ xb)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i - n * C, cpg = C / G, c0 = (c / cpg) * cpg;
  double s0 = 0.0, s1 = 0.0;
  for (int j = 0; j < cpg; ++j) {
    s0 += sums[((int64_t)n * C + c0 + j) * 2];
    s1 += sums[((int64_t)n * C + c0 + j) * 2 + 1];
  }
  const double mean = s0 / m;
  double var = s1 / m - mean * mean;
  var = var < 0.0 ? 0.0 : var;
  const float rstd = (float)(1.0 / sqrt(var + eps)), mu = (float)mean;
  const float av = rstd * weight[c];
  a[i] = av;
  b[i] = bias[c] - mu * av;
  mean_c[i] = mu;
  rstd_c[i] = rstd;
  xb[i] = -mu * rstd;
}

__global__ __launch_bounds__(256) void gn_coeffs_bwd(const double *__restrict__ sums, const float *__restrict__ weight,
                                                     const float *__restrict__ mean_c, const float *__restrict__ rstd_c, int N,
                                                     int C, int G, double m, float *__restrict__ a, float *__restrict__ pcoef,
                                                     float *__restrict__ rcoef, float *__restrict__ gw, float *__restrict__ gb)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i - n * C, cpg = C / G, c0 = (c / cpg) * cpg;
  double S1 = 0.0, S2 = 0.0;
  for (int j = 0; j < cpg; ++j) {
    const double w = (double)weight[c0 + j];
    S1 += sums[((int64_t)n * C + c0 + j) * 2] * w;           // sum_p gamma * dy * x_hat
    S2 += sums[((int64_t)n * C + c0 + j) * 2 + 1] * w;       // sum_p gamma * dy
  }
  S1 /= m; S2 /= m;
  const double r = (double)rstd_c[i], mu = (double)mean_c[i];
  a[i] = rstd_c[i] * weight[c];
  pcoef[i] = (float)(-(r * r) * S1);
  rcoef[i] = (float)((r * r) * S1 * mu - r * S2);
  if (n == 0) {
    double w0 = 0.0, w1 = 0.0;
    for (int k = 0; k < N; ++k) {
      w0 += sums[((int64_t)k * C + c) * 2];
      w1 += sums[((int64_t)k * C + c) * 2 + 1];
    }
    gw[c] = (float)w0;
    gb[c] = (float)w1;
  }
}

}  // namespace

extern "C" int pd_maxpool3s2_fwd_bf16(const void *x, void *y, void *argmax, int B, int H, int W, int C, void *stream_)
{
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_maxpool3s2_fwd_bf16: B=%d H=%d W=%d C=%d (%% 8)", B, H, W, C);
  if (B == 0) return PD_OK;
  if (!x || !y || !argmax) return pd_set_error(PD_ERR_INVALID_ARG, "pd_maxpool3s2_fwd_bf16: null pointer");
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const int64_t total = (int64_t)B * OH * OW * (C / 8);
  const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  hipLaunchKernelGGL(maxpool3s2_fwd, dim3(grid), dim3(256), 0, (hipStream_t)stream_, (const u16x8 *)x, (u16x8 *)y, (unsigned long long *)argmax, B, H, W,
                     OH, OW, C / 8);
  return pd_check_launch("pd_maxpool3s2_fwd_bf16");
}

extern "C" int pd_maxpool3s2_bwd_bf16(const void *dy, const void *argmax, void *dx, int B, int H, int W, int C, void *stream_)
{
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_maxpool3s2_bwd_bf16: B=%d H=%d W=%d C=%d (%% 8)", B, H, W, C);
  if (B == 0) return PD_OK;
  if (!dy || !dx || !argmax) return pd_set_error(PD_ERR_INVALID_ARG, "pd_maxpool3s2_bwd_bf16: null pointer");
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const int64_t total = (int64_t)B * H * W * (C / 8);
  const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  hipLaunchKernelGGL(maxpool3s2_bwd, dim3(grid), dim3(256), 0, (hipStream_t)stream_, (const u16x8 *)dy, (const unsigned long long *)argmax, (u16x8 *)dx, B, H, W,
                     OH, OW, C / 8);
  return pd_check_launch("pd_maxpool3s2_bwd_bf16");
}

extern "C" int pd_affine_act_fwd_bf16(const void *x, const void *residual, const float *scale, const float *bias, void *y,
                                      int64_t n, int channels, int relu, void *stream_)
{
  if (n < 0 || channels <= 0 || (channels & 7) || (n % channels)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_affine_act_fwd_bf16: n=%lld channels=%d", (long long)n, channels);
  if (n == 0) return PD_OK;
  if (!x || !scale || !bias || !y) return pd_set_error(PD_ERR_INVALID_ARG, "pd_affine_act_fwd_bf16: null pointer");
  const int64_t n8 = n / 8;
  const int c8 = channels / 8;
  hipStream_t s = (hipStream_t)stream_;
  dim3 g(grid_for(n8)), b(256);
#define LAUNCH(R, A) hipLaunchKernelGGL((affine_act_fwd<R, A>), g, b, 0, s, (const u16x8 *)x, (const u16x8 *)residual, scale, bias, (u16x8 *)y, n8, c8)
  if (residual) { if (relu) LAUNCH(true, true); else LAUNCH(true, false); }
  else { if (relu) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
  return pd_check_launch("pd_affine_act_fwd_bf16");
}

extern "C" int pd_affine_act_bwd2_bf16(const void *gy, const void *gy2, const void *y, const float *scale, void *gx, void *gres, int64_t n,
                                       int channels, int relu, void *stream_)
{
  if (n < 0 || channels <= 0 || (channels & 7) || (n % channels)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_affine_act_bwd_bf16: n=%lld channels=%d", (long long)n, channels);
  if (n == 0) return PD_OK;
  if (!gy || !scale || !gx || (relu && !y)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_affine_act_bwd_bf16: null pointer");
  const int64_t n8 = n / 8;
  const int c8 = channels / 8;
  hipStream_t s = (hipStream_t)stream_;
  dim3 g(grid_for(n8)), b(256);
#define LAUNCH(R, A, T) hipLaunchKernelGGL((affine_act_bwd<R, A, T>), g, b, 0, s, (const u16x8 *)gy, (const u16x8 *)gy2, (const u16x8 *)y, scale, (u16x8 *)gx, (u16x8 *)gres, n8, c8)
#define LAUNCH2(R, A) do { if (gy2) LAUNCH(R, A, true); else LAUNCH(R, A, false); } while (0)
  if (gres) { if (relu) LAUNCH2(true, true); else LAUNCH2(true, false); }
  else { if (relu) LAUNCH2(false, true); else LAUNCH2(false, false); }
#undef LAUNCH2
#undef LAUNCH
  return pd_check_launch("pd_affine_act_bwd_bf16");
}

extern "C" int pd_affine_act_bwd_bf16(const void *gy, const void *y, const float *scale, void *gx, void *gres, int64_t n,
                                      int channels, int relu, void *stream_)
{
  return pd_affine_act_bwd2_bf16(gy, nullptr, y, scale, gx, gres, n, channels, relu, stream_);
}

extern "C" int pd_multi_gather_sumsq(const int64_t *src_ptrs, const int32_t *src_is_bf16, const int32_t *blk_tensor,
                                     const int64_t *blk_start, const int64_t *blk_dst, const int32_t *blk_len, float *dst,
                                     double *sumsq, int block_begin, int block_end, void *stream_)
{
  if (block_end < block_begin) return pd_set_error(PD_ERR_INVALID_ARG, "pd_multi_gather_sumsq: bad block range");
  if (block_end == block_begin) return PD_OK;
  if (!src_ptrs || !src_is_bf16 || !blk_tensor || !blk_start || !blk_dst || !blk_len || !dst)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_multi_gather_sumsq: null pointer");
  hipLaunchKernelGGL(multi_gather_sumsq, dim3(block_end - block_begin), dim3(256), 0, (hipStream_t)stream_, src_ptrs, src_is_bf16,
                     blk_tensor, blk_start, blk_dst, blk_len, dst, sumsq, block_begin);
  return pd_check_launch("pd_multi_gather_sumsq");
}

static int nc_check(int N, int P, int C, const char *who)
{
  if (N < 0 || P < 0 || C <= 0 || (C & 3) || (256 % (C >> 2)) != 0 || C > 256)
    return pd_set_error(PD_ERR_INVALID_ARG, "%s: N=%d P=%d C=%d (C must be 4*2^k <= 256)", who, N, P, C);
  return PD_OK;
}

extern "C" int pd_nc_sums_f32(const float *x, const float *dy, const float *y, const float *a, const float *b, double *out,
                              int N, int P, int C, int mode, int relu, void *stream_)
{
  int rc = nc_check(N, P, C, "pd_nc_sums_f32");
  if (rc) return rc;
  if (!out) return pd_set_error(PD_ERR_INVALID_ARG, "pd_nc_sums_f32: null output");
  hipStream_t s = (hipStream_t)stream_;
  (void)hipMemsetAsync(out, 0, (size_t)N * C * 2 * sizeof(double), s);
  if ((int64_t)N * P == 0) return pd_check_launch("pd_nc_sums_f32");
  if (!x || (mode == 1 && (!dy || !a || !b || (relu && !y)))) return pd_set_error(PD_ERR_INVALID_ARG, "pd_nc_sums_f32: null input");
  // pixels per block (tools/debug/nc_sums_sweep.py, round 6).  Every block ends in 2 C fp64 atomics on the SAME 2 C addresses per image, and those
  // serialise in L2: at 2 x 256^2 x 256 the launch costs ~15 us + 29 ns per block (16 / 128 / 512 pixels per block: 252 / 45 / 25 us), so the large
  // maps take few, long blocks; the small maps were the opposite problem — 4 or 16 blocks of 512 pixels are 128 dependent iterations per thread
  // on an empty machine (the per-launch trace showed 14-27 us for 2-8 MB) — and take many short ones.  Backward mode streams three operands:
  // its optimum sits at shorter blocks.
  static const int ppb_env = []() { const char *e = getenv("PD_NC_PPB"); return e ? atoi(e) : 0; }();      // tools/ A/B only
  const int64_t np = (int64_t)N * P;
  int ppb = mode == 0 ? (np >= 131072 ? 512 : np >= 32768 ? 256 : np >= 8192 ? 128 : 64) : (np >= 131072 ? 256 : np >= 32768 ? 128 : 32);
  if (ppb_env > 0) ppb = ppb_env;
  dim3 grid((P + ppb - 1) / ppb, N);
  if (mode == 0) hipLaunchKernelGGL(nc_sums<0>, grid, dim3(256), 0, s, x, dy, y, a, b, out, P, C, ppb, relu);
  else hipLaunchKernelGGL(nc_sums<1>, grid, dim3(256), 0, s, x, dy, y, a, b, out, P, C, ppb, relu);
  return pd_check_launch("pd_nc_sums_f32");
}

static int nc_affine_launch(const float *x, const float *a, const float *b, float *y, int N, int P, int C, int relu, void *stream_, float *amax)
{
  int rc = nc_check(N, P, C, "pd_nc_affine_f32");
  if (amax && C != 256) return pd_set_error(PD_ERR_INVALID_ARG, "pd_nc_affine_amax_f32: pixel maxima need C == 256");
  if (rc) return rc;
  const int64_t n4 = (int64_t)N * P * C / 4;
  if (n4 == 0) return PD_OK;
  if (!x || !a || !b || !y) return pd_set_error(PD_ERR_INVALID_ARG, "pd_nc_affine_f32: null pointer");
  hipStream_t s = (hipStream_t)stream_;
  if (relu) hipLaunchKernelGGL(nc_affine<true>, dim3(grid_for(n4)), dim3(256), 0, s, (const float4 *)x, a, b, (float4 *)y, n4, P, C, amax);
  else hipLaunchKernelGGL(nc_affine<false>, dim3(grid_for(n4)), dim3(256), 0, s, (const float4 *)x, a, b, (float4 *)y, n4, P, C, amax);
  return pd_check_launch("pd_nc_affine_f32");
}

extern "C" int pd_nc_affine_f32(const float *x, const float *a, const float *b, float *y, int N, int P, int C, int relu, void *stream_)
{
  return nc_affine_launch(x, a, b, y, N, P, C, relu, stream_, nullptr);
}

extern "C" int pd_nc_affine_amax_f32(const float *x, const float *a, const float *b, float *y, float *amax, int N, int P, int C, int relu, void *stream_)
{
  return nc_affine_launch(x, a, b, y, N, P, C, relu, stream_, amax);
}

static int nc_affine2_launch(const float *dy, const float *x, const float *y, const float *a, const float *p, const float *r,
                             float *dx, int N, int P, int C, int relu, void *stream_, float *amax)
{
  int rc = nc_check(N, P, C, "pd_nc_affine2_f32");
  if (amax && C != 256) return pd_set_error(PD_ERR_INVALID_ARG, "pd_nc_affine2_amax_f32: pixel maxima need C == 256");
  if (rc) return rc;
  const int64_t n4 = (int64_t)N * P * C / 4;
  if (n4 == 0) return PD_OK;
  if (!dy || !x || !a || !p || !r || !dx || (relu && !y)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_nc_affine2_f32: null pointer");
  hipStream_t s = (hipStream_t)stream_;
  if (relu) hipLaunchKernelGGL(nc_affine2<true>, dim3(grid_for(n4)), dim3(256), 0, s, (const float4 *)dy, (const float4 *)x, (const float4 *)y, a, p, r, (float4 *)dx, n4, P, C, amax);
  else hipLaunchKernelGGL(nc_affine2<false>, dim3(grid_for(n4)), dim3(256), 0, s, (const float4 *)dy, (const float4 *)x, (const float4 *)y, a, p, r, (float4 *)dx, n4, P, C, amax);
  return pd_check_launch("pd_nc_affine2_f32");
}

extern "C" int pd_nc_affine2_f32(const float *dy, const float *x, const float *y, const float *a, const float *p, const float *r,
                                 float *dx, int N, int P, int C, int relu, void *stream_)
{
  return nc_affine2_launch(dy, x, y, a, p, r, dx, N, P, C, relu, stream_, nullptr);
}

extern "C" int pd_nc_affine2_amax_f32(const float *dy, const float *x, const float *y, const float *a, const float *p, const float *r,
                                      float *dx, float *amax, int N, int P, int C, int relu, void *stream_)
{
  return nc_affine2_launch(dy, x, y, a, p, r, dx, N, P, C, relu, stream_, amax);
}

extern "C" int pd_gn_coeffs_fwd(const double *sums, const float *weight, const float *bias, int N, int C, int G, int P, float eps,
                                float *a, float *b, float *mean_c, float *rstd_c, float *xb, void *stream_)
{
  if (N < 0 || C <= 0 || G <= 0 || (C % G) || P <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_gn_coeffs_fwd: N=%d C=%d G=%d P=%d", N, C, G, P);
  if (N == 0) return PD_OK;
  if (!sums || !weight || !bias || !a || !b || !mean_c || !rstd_c || !xb) return pd_set_error(PD_ERR_INVALID_ARG, "pd_gn_coeffs_fwd: null pointer");
  hipLaunchKernelGGL(gn_coeffs_fwd, dim3((N * C + 255) / 256), dim3(256), 0, (hipStream_t)stream_, sums, weight, bias, N, C, G,
                     (double)P * (C / G), (double)eps, a, b, mean_c, rstd_c, xb);
  return pd_check_launch("pd_gn_coeffs_fwd");
}

extern "C" int pd_gn_coeffs_bwd(const double *sums, const float *weight, const float *mean_c, const float *rstd_c, int N, int C,
                                int G, int P, float *a, float *pcoef, float *rcoef, float *gw, float *gb, void *stream_)
{
  if (N < 0 || C <= 0 || G <= 0 || (C % G) || P <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_gn_coeffs_bwd: N=%d C=%d G=%d P=%d", N, C, G, P);
  if (N == 0) return PD_OK;
  if (!sums || !weight || !mean_c || !rstd_c || !a || !pcoef || !rcoef || !gw || !gb) return pd_set_error(PD_ERR_INVALID_ARG, "pd_gn_coeffs_bwd: null pointer");
  hipLaunchKernelGGL(gn_coeffs_bwd, dim3((N * C + 255) / 256), dim3(256), 0, (hipStream_t)stream_, sums, weight, mean_c, rstd_c, N, C,
                     G, (double)P * (C / G), a, pcoef, rcoef, gw, gb);
  return pd_check_launch("pd_gn_coeffs_bwd");
}
