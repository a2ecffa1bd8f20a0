// fp32 GEMMs on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD).
//
// Used for the fp32 nn.Linear layers of the deformable-attention encoder (tokens x 256/1024), where the library
// heuristic picks kernels that reach ~45 TFLOP/s on these tall-skinny shapes (profiles/r01_*).  Two kernels:
//   gemm_tn_f32     C[M,N] = A[M,K] B[N,K]^T (+bias)(ReLU)    128x128x32 tiles, 4 waves (2x2) x (2x2) MFMA tiles
//   gemm_wgrad_f32  dW[N,K] += dY[M,N]^T X[M,K]                128x128 output tiles, contraction split over blocks
// LDS rows are padded (+4 floats) so the 16-lane groups of a ds_read_b128 touch 64 distinct banks; global loads
// are 16-byte lanes covering 128-byte row segments; the next tile's global loads are issued before the MFMAs of
// the current one and written to the other LDS buffer afterwards (register-staged double buffering).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pd_common.h"
#include "pd_gemm.h"
#include "pd_msda.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, LDSROW = BK + 4;

__device__ __forceinline__ int xcd_chunk(int bid, int nb)
{
  // contiguous logical ranges per XCD when the grid is a multiple of 8, identity otherwise
  return (nb & 7) == 0 ? (bid & 7) * (nb >> 3) + (bid >> 3) : bid;
}

template <bool RELU>
__global__ __launch_bounds__(256, 2) void gemm_tn_f32(const float *__restrict__ A, const float *__restrict__ B,
                                                       const float *__restrict__ bias, float *__restrict__ C, int M, int N,
                                                       int K, int lda, int ldb, int ldc, int ntiles_n)
{
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float(*As)[BM][LDSROW] = reinterpret_cast<float(*)[BM][LDSROW]>(smem);
  float(*Bs)[BN][LDSROW] = reinterpret_cast<float(*)[BN][LDSROW]>(smem + 2 * BM * LDSROW);
  const int lb = xcd_chunk(blockIdx.x, gridDim.x);
  const int m0 = (lb / ntiles_n) * BM, n0 = (lb % ntiles_n) * BN;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  // global -> register staging: thread covers rows lr + 32*j, columns lk..lk+3 of both tiles
  const int lr = t >> 3, lk = (t & 7) * 4;
  float4 ra[4], rb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = lr + 32 * j, k = k0 + lk;
      ra[j] = (m0 + r < M && k < K) ? *reinterpret_cast<const float4 *>(A + (int64_t)(m0 + r) * lda + k) : make_float4(0, 0, 0, 0);
      rb[j] = (n0 + r < N && k < K) ? *reinterpret_cast<const float4 *>(B + (int64_t)(n0 + r) * ldb + k) : make_float4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *reinterpret_cast<float4 *>(&As[buf][lr + 32 * j][lk]) = ra[j];
      *reinterpret_cast<float4 *>(&Bs[buf][lr + 32 * j][lk]) = rb[j];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int KT = (K + BK - 1) / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  const int fr = lane & 31, fk = (lane >> 5) * 4;
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) gload((kt + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < BK / 8; ++ks) {
      float4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const float4 *>(&As[buf][wm + i * 32 + fr][ks * 8 + fk]);
        b[i] = *reinterpret_cast<const float4 *>(&Bs[buf][wn + i * 32 + fr][ks * 8 + fk]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32((&a[i].x)[e], (&b[j].x)[e], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < KT) lstore(buf ^ 1);
    __syncthreads();
  }
  // epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn + j * 32 + (lane & 31);
    if (col >= N) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < M) {
          float v = acc[i][j][e] + bv;
          if (RELU) v = fmaxf(v, 0.f);
          C[(int64_t)row * ldc + col] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- weight gradient
constexpr int WM = 16;   // contraction rows per LDS stage

__global__ __launch_bounds__(256, 2) void gemm_wgrad_f32(const float *__restrict__ dY, const float *__restrict__ X,
                                                          float *__restrict__ dW, float *__restrict__ dB, int M, int N, int K,
                                                          int ldy, int ldx, int ldw, int tiles_k, int tiles, int m_chunk)
{
  __shared__ __attribute__((aligned(16))) float Ys[2][WM][BN];
  __shared__ __attribute__((aligned(16))) float Xs[2][WM][BM];
  const int tile = blockIdx.x % tiles, split = blockIdx.x / tiles;
  const int n0 = (tile / tiles_k) * BN, k0 = (tile % tiles_k) * BM;
  const int mb = split * m_chunk, me = min(M, mb + m_chunk);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;
  const int lr = t >> 5, lc = (t & 31) * 4;     // rows lr, lr+8; 4 columns each
  float4 ry[2], rx[2];
  auto gload = [&](int m) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = m + lr + 8 * j;
      ry[j] = (r < me && n0 + lc < N) ? *reinterpret_cast<const float4 *>(dY + (int64_t)r * ldy + n0 + lc) : make_float4(0, 0, 0, 0);
      rx[j] = (r < me && k0 + lc < K) ? *reinterpret_cast<const float4 *>(X + (int64_t)r * ldx + k0 + lc) : make_float4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      *reinterpret_cast<float4 *>(&Ys[buf][lr + 8 * j][lc]) = ry[j];
      *reinterpret_cast<float4 *>(&Xs[buf][lr + 8 * j][lc]) = rx[j];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int steps = (me - mb + WM - 1) / WM;
  if (steps > 0) {
    gload(mb);
    lstore(0);
  }
  __syncthreads();
  const int fc = lane & 31, fm = lane >> 5;
  // bias gradient dB[n] = sum_m dY[m,n]: the k-tile-0 workgroups add up the dY tile they stage anyway
  const bool do_bias = dB != nullptr && k0 == 0;
  const int bc = t & 127, br = (t >> 7) * 8;
  float bsum = 0.f;
  for (int s = 0; s < steps; ++s) {
    const int buf = s & 1;
    if (s + 1 < steps) gload(mb + (s + 1) * WM);
    if (do_bias) {
#pragma unroll
      for (int r = 0; r < 8; ++r) bsum += Ys[buf][br + r][bc];
    }
#pragma unroll
    for (int kk = 0; kk < WM / 2; ++kk) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = Ys[buf][2 * kk + fm][wn + i * 32 + fc];
        b[i] = Xs[buf][2 * kk + fm][wk + i * 32 + fc];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (s + 1 < steps) lstore(buf ^ 1);
    __syncthreads();
  }
  if (do_bias && n0 + bc < N) unsafeAtomicAdd(dB + n0 + bc, bsum);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = k0 + wk + j * 32 + (lane & 31);
    if (col >= K) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = n0 + wn + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < N) unsafeAtomicAdd(dW + (int64_t)row * ldw + col, acc[i][j][e]);
      }
  }
}

bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int pd_gemm_tn_f32(const float *A, const float *B, const float *bias, float *C, int M, int N, int K, int lda,
                              int ldb, int ldc, int relu, void *stream_)
{
  if (M < 0 || N < 0 || K < 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f32: negative size");
  if (M == 0 || N == 0) return PD_OK;
  if (!A || !B || !C) return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f32: null pointer");
  if ((K & 3) || (lda & 3) || (ldb & 3) || !aligned16(A) || !aligned16(B))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_gemm_tn_f32: K, lda, ldb must be multiples of 4 and A, B 16-byte aligned");
  const int tn = (N + BN - 1) / BN, tm = (M + BM - 1) / BM;
  const size_t lds = (size_t)2 * (BM + BN) * LDSROW * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)gemm_tn_f32<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void *)gemm_tn_f32<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  auto k = relu ? gemm_tn_f32<true> : gemm_tn_f32<false>;
  hipLaunchKernelGGL(k, dim3((unsigned)((int64_t)tm * tn)), dim3(256), lds, (hipStream_t)stream_, A, B, bias, C, M, N, K, lda,
                     ldb, ldc, tn);
  return pd_check_launch("pd_gemm_tn_f32");
}

int g_pd_dbg_wgrad_wgs = 0;          // experiment knob (pd_debug_set "wgrad_wgs"): > 0 overrides the split policy

static int wgrad_launch(const float *dY, const float *X, float *dW, float *dB, int M, int N, int K, int ldy, int ldx, int ldw,
                        bool zero_first, hipStream_t s, const char *who)
{
  if (M < 0 || N < 0 || K < 0) return pd_set_error(PD_ERR_INVALID_ARG, "%s: negative size", who);
  if (N == 0 || K == 0) return PD_OK;
  if (!dW || (M > 0 && (!dY || !X))) return pd_set_error(PD_ERR_INVALID_ARG, "%s: null pointer", who);
  if ((N & 3) || (K & 3) || (ldy & 3) || (ldx & 3) || !aligned16(dY) || !aligned16(X))
    return pd_set_error(PD_ERR_INVALID_ARG, "%s: N, K, ldy, ldx must be multiples of 4, 16-byte aligned", who);
  if (zero_first) {
    (void)hipMemset2DAsync(dW, (size_t)ldw * sizeof(float), 0, (size_t)K * sizeof(float), (size_t)N, s);
    if (dB) (void)hipMemsetAsync(dB, 0, (size_t)N * sizeof(float), s);
  }
  if (M == 0) return pd_check_launch(who);
  const int tk = (K + BM - 1) / BM, tn = (N + BN - 1) / BN, tiles = tk * tn;
  // split the contraction so the chip is full: ~4 workgroups per CU when the output has many tiles; one per CU when it has
  // only a few (each split ends in a tile of atomics, and with <= 4 tiles they would otherwise dominate: measured 87 -> 78 us
  // at 256x256, 57 -> 47 us at 96x256, M = 43008)
  const int target = g_pd_dbg_wgrad_wgs > 0 ? g_pd_dbg_wgrad_wgs : (tiles <= 4 ? 256 : 1024);
  int splits = (target + tiles - 1) / tiles;
  int m_chunk = ((M + splits - 1) / splits + WM - 1) / WM * WM;
  if (m_chunk < 4 * WM) m_chunk = 4 * WM;
  splits = (M + m_chunk - 1) / m_chunk;
  hipLaunchKernelGGL(gemm_wgrad_f32, dim3((unsigned)(tiles * splits)), dim3(256), 0, s, dY, X, dW, dB, M, N, K, ldy, ldx, ldw, tk,
                     tiles, m_chunk);
  return pd_check_launch(who);
}

extern "C" int pd_gemm_wgrad_f32(const float *dY, const float *X, float *dW, float *dB, int M, int N, int K, int ldy,
                                 int ldx, int ldw, void *stream_)
{
  return wgrad_launch(dY, X, dW, dB, M, N, K, ldy, ldx, ldw, true, (hipStream_t)stream_, "pd_gemm_wgrad_f32");
}

extern "C" int pd_gemm_wgrad_acc_f32(const float *dY, const float *X, float *dW, float *dB, int M, int N, int K, int ldy,
                                     int ldx, int ldw, void *stream_)
{
  return wgrad_launch(dY, X, dW, dB, M, N, K, ldy, ldx, ldw, false, (hipStream_t)stream_, "pd_gemm_wgrad_acc_f32");
}
