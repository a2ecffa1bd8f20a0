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
