// bf16 matrix-core GEMMs for skinny activations (gfx950): 32 x 128 output tiles, 64-deep chunks double-buffered through
// LDS, one 32 x 32 sub-tile per wavefront on v_mfma_f32_32x32x8_bf16_1k.  C-ABI in include/pd_smallgemm.h.
//   tn     Y = X W^T (+bias)(ReLU)   both operands are read as "row, 4 consecutive k": 8-byte LDS reads
//   nn     dX = dY W                 the W operand is "4 consecutive n at fixed k": four 16-bit LDS reads
//   wgrad  dW = dY^T X               both operands are "4 consecutive m at fixed column": 16-bit LDS reads
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdint.h>

#include "mfma_bf16.h"
#include "pd_common.h"
#include "pd_msda.h"
#include "pd_smallgemm.h"

namespace {

using namespace pdmfma;

constexpr int P64 = 68;     // padded LDS row (bf16 elements) of a 64-column tile: 136-byte rows, 8-byte reads conflict-free
constexpr int P128 = 136;   // padded row of a 128-column tile
constexpr int P32 = 40;     // padded row of a 32-column tile

// 8 bf16 (16 bytes) of a [nrows x ncols] row-major matrix at (row, col), zeros outside; ncols % 4 == 0, col % 8 == 0
__device__ __forceinline__ uint4 load_piece(const bf16_t *base, int ld, int row, int col, int nrows, int ncols)
{
  uint4 v = make_uint4(0, 0, 0, 0);
  if (row >= nrows || col >= ncols) return v;
  const bf16_t *p = base + (int64_t)row * ld + col;
  if (col + 8 <= ncols) return *reinterpret_cast<const uint4 *>(p);
  const uint2 h = *reinterpret_cast<const uint2 *>(p);                     // 4 valid elements
  v.x = h.x; v.y = h.y;
  return v;
}
__device__ __forceinline__ void store_piece(bf16_t *dst, uint4 v)          // 8-byte aligned LDS destination
{
  *reinterpret_cast<uint2 *>(dst) = make_uint2(v.x, v.y);
  *reinterpret_cast<uint2 *>(dst + 4) = make_uint2(v.z, v.w);
}

// ------------------------------------------------------------------------------------------------ Y = X W^T
template <bool RELU>
__global__ __launch_bounds__(256) void sgemm_tn(const bf16_t *__restrict__ X, const bf16_t *__restrict__ W,
                                                const bf16_t *__restrict__ bias, bf16_t *__restrict__ Y, int M, int N, int K,
                                                int ldx, int ldw, int ldy)
{
  __shared__ __attribute__((aligned(16))) bf16_t Xs[2][32][P64];
  __shared__ __attribute__((aligned(16))) bf16_t Ws[2][128][P64];
  const int nb = blockIdx.x * 128, mb = blockIdx.y * 32;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int prow = tid >> 3, pcol = (tid & 7) * 8;                          // 32 rows x 8 pieces per 256 threads
  uint4 xr, wr[4];
  auto gload = [&](int k0) {
    xr = load_piece(X, ldx, mb + prow, k0 + pcol, M, K);
#pragma unroll
    for (int i = 0; i < 4; ++i) wr[i] = load_piece(W, ldw, nb + prow + 32 * i, k0 + pcol, N, K);
  };
  auto lstore = [&](int buf) {
    store_piece(&Xs[buf][prow][pcol], xr);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_piece(&Ws[buf][prow + 32 * i][pcol], wr[i]);
  };
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int nchunk = K / 64;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunk) gload((c + 1) * 64);
#pragma unroll
    for (int s = 0; s < 8; ++s)                                             // acc[n][m] += W[n][k..] . X[m][k..]
      mma(acc, lds4(&Ws[buf][32 * wave + r][8 * s + 4 * hh]), lds4(&Xs[buf][r][8 * s + 4 * hh]));
    if (c + 1 < nchunk) lstore(buf ^ 1);
    __syncthreads();
  }
  const int m = mb + r;
  if (m >= M) return;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n0 = nb + 32 * wave + 8 * g + 4 * hh;
    if (n0 >= N) continue;
    float v[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    if (bias) {
      const uint2 b = *reinterpret_cast<const uint2 *>(bias + n0);
      v[0] += bf_lo(b.x); v[1] += bf_hi(b.x); v[2] += bf_lo(b.y); v[3] += bf_hi(b.y);
    }
    if (RELU) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    *reinterpret_cast<bf16x4 *>(Y + (int64_t)m * ldy + n0) = pack4(v[0], v[1], v[2], v[3]);
  }
}

// The same product for K <= 256 with the WHOLE contraction staged at once.  These GEMMs are pure latency: 4 chunks of 64 walked one
// after the other each wait out a global-load latency (~1.7 us) for 4 MFMAs of work — 12.5 us per launch of which ~5 are launch cost.
// Here every load of the workgroup (X 32 x K, W 128 x K: 20 x 16 bytes per thread) is in flight together, one LDS stage, one barrier.
constexpr int PK256 = 256 + 8;                                               // bf16 elements per LDS row (528 B)
template <bool RELU>
__global__ __launch_bounds__(256) void sgemm_tn_k256(const bf16_t *__restrict__ X, const bf16_t *__restrict__ W,
                                                     const bf16_t *__restrict__ bias, bf16_t *__restrict__ Y, int M, int N, int K,
                                                     int ldx, int ldw, int ldy, int64_t bsx = 0, int64_t bsw = 0, int64_t bsy = 0)
{
  // blockIdx.z = problem of a batch of equally shaped products (element strides bsx / bsw / bsy; pd_sgemm_tn_batched_bf16)
  X += blockIdx.z * bsx; W += blockIdx.z * bsw; Y += blockIdx.z * bsy;
  extern __shared__ __attribute__((aligned(16))) unsigned char sg_smem[];
  bf16_t(*Xs)[PK256] = reinterpret_cast<bf16_t(*)[PK256]>(sg_smem);                       // [32][PK256]
  bf16_t(*Ws)[PK256] = reinterpret_cast<bf16_t(*)[PK256]>(sg_smem + 32 * PK256 * sizeof(bf16_t));   // [128][PK256]
  const int nb = blockIdx.x * 128, mb = blockIdx.y * 32;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int prow = tid >> 5, pcol = (tid & 31) * 8;                          // 8 rows x 32 pieces per pass of the 256 threads
  uint4 xr[4], wr[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) xr[i] = load_piece(X, ldx, mb + prow + 8 * i, pcol, M, K);
#pragma unroll
  for (int i = 0; i < 16; ++i) wr[i] = load_piece(W, ldw, nb + prow + 8 * i, pcol, N, K);
#pragma unroll
  for (int i = 0; i < 4; ++i) store_piece(&Xs[prow + 8 * i][pcol], xr[i]);
#pragma unroll
  for (int i = 0; i < 16; ++i) store_piece(&Ws[prow + 8 * i][pcol], wr[i]);
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int steps = K / 8;
  for (int s = 0; s < steps; ++s)                                           // acc[n][m] += W[n][k..] . X[m][k..]
    mma(acc, lds4(&Ws[32 * wave + r][8 * s + 4 * hh]), lds4(&Xs[r][8 * s + 4 * hh]));
  const int m = mb + r;
  if (m >= M) return;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n0 = nb + 32 * wave + 8 * g + 4 * hh;
    if (n0 >= N) continue;
    float v[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    if (bias) {
      const uint2 b = *reinterpret_cast<const uint2 *>(bias + n0);
      v[0] += bf_lo(b.x); v[1] += bf_hi(b.x); v[2] += bf_lo(b.y); v[3] += bf_hi(b.y);
    }
    if (RELU) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    *reinterpret_cast<bf16x4 *>(Y + (int64_t)m * ldy + n0) = pack4(v[0], v[1], v[2], v[3]);
  }
}

// Up to four independent K <= 256 products in ONE launch (blockIdx.z = problem): the q / k / v projections of an attention block read
// two different inputs and three slices of one packed weight — three 7 us launches (and, for the cross-attention's keys and values
// over the memory tokens, two library GEMMs) become one.  Same body as sgemm_tn_k256; the problem table travels as a kernel argument.
struct SgTnMulti {
  const bf16_t *X[4], *W[4], *bias[4];
  bf16_t *Y[4];
  int M[4], N[4], ldx[4], ldw[4], ldy[4];
};
__global__ __launch_bounds__(256) void sgemm_tn_k256_multi(const SgTnMulti p, int K)
{
  const int z = blockIdx.z;
  const int M = p.M[z], N = p.N[z];
  const int nb = blockIdx.x * 128, mb = blockIdx.y * 32;
  if (nb >= N || mb >= M) return;                                            // the grid covers the largest problem
  const bf16_t *X = p.X[z], *W = p.W[z], *bias = p.bias[z];
  bf16_t *Y = p.Y[z];
  const int ldx = p.ldx[z], ldw = p.ldw[z], ldy = p.ldy[z];
  extern __shared__ __attribute__((aligned(16))) unsigned char sg_smem[];
  bf16_t(*Xs)[PK256] = reinterpret_cast<bf16_t(*)[PK256]>(sg_smem);
  bf16_t(*Ws)[PK256] = reinterpret_cast<bf16_t(*)[PK256]>(sg_smem + 32 * PK256 * sizeof(bf16_t));
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int prow = tid >> 5, pcol = (tid & 31) * 8;
  uint4 xr[4], wr[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) xr[i] = load_piece(X, ldx, mb + prow + 8 * i, pcol, M, K);
#pragma unroll
  for (int i = 0; i < 16; ++i) wr[i] = load_piece(W, ldw, nb + prow + 8 * i, pcol, N, K);
#pragma unroll
  for (int i = 0; i < 4; ++i) store_piece(&Xs[prow + 8 * i][pcol], xr[i]);
#pragma unroll
  for (int i = 0; i < 16; ++i) store_piece(&Ws[prow + 8 * i][pcol], wr[i]);
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int steps = K / 8;
  for (int s = 0; s < steps; ++s) mma(acc, lds4(&Ws[32 * wave + r][8 * s + 4 * hh]), lds4(&Xs[r][8 * s + 4 * hh]));
  const int m = mb + r;
  if (m >= M) return;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n0 = nb + 32 * wave + 8 * g + 4 * hh;
    if (n0 >= N) continue;
    float v[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    if (bias) {
      const uint2 b = *reinterpret_cast<const uint2 *>(bias + n0);
      v[0] += bf_lo(b.x); v[1] += bf_hi(b.x); v[2] += bf_lo(b.y); v[3] += bf_hi(b.y);
    }
    *reinterpret_cast<bf16x4 *>(Y + (int64_t)m * ldy + n0) = pack4(v[0], v[1], v[2], v[3]);
  }
}

// Long contractions over few rows (the decoder FFN's second Linear: [200, 2048] -> 256, 14 output tiles): walked chunk by chunk in
// ONE workgroup per tile they are a chain of global-load latencies on 14 of the 256 CUs (39 us).  Here the contraction is cut into
// 256-wide slices (blockIdx.z), each exactly the all-loads-in-flight kernel above; a slice leaves its fp32 accumulators in a
// workspace (accumulator order: element e of thread t at e * 256 + t) and the LAST workgroup of a tile to arrive (agent-scope
// ticket, reset for the next launch) adds the slices up in slice order — deterministic — and runs the epilogue.
// EPI 0: Y = sum + bias (ReLU); EPI 1 (the nn form below shares the reduction).
__device__ __forceinline__ bool splitk_arrive(float *__restrict__ ws, int *__restrict__ tickets, int tile, int tiles, int slices, const f32x16 &acc)
{
  __shared__ int s_last;
  float *mine = ws + ((int64_t)blockIdx.z * tiles + tile) * (32 * 128);
#pragma unroll
  for (int e = 0; e < 16; ++e) mine[e * 256 + threadIdx.x] = acc[e];
  __threadfence();                                                          // this thread's partials visible device-wide ...
  __syncthreads();
  if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(tickets + tile, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == slices - 1;
  __syncthreads();
  if (!s_last) return false;
  __threadfence();                                                          // ... before the last arrival reads the others'
  if (threadIdx.x == 0) tickets[tile] = 0;
  return true;
}
__device__ __forceinline__ void splitk_sum(const float *__restrict__ ws, int tile, int tiles, int slices, f32x16 &acc)
{
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  // plain loads: the agent-scope fence of splitk_arrive() has invalidated this CU's L1 and nothing here has read these addresses
  // since; all of a slice's 16 loads are independent (agent-scope atomic loads were issued one by one: 23 us per launch)
#pragma unroll 2
  for (int sl = 0; sl < slices; ++sl) {
    const float *o = ws + ((int64_t)sl * tiles + tile) * (32 * 128) + threadIdx.x;
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = __builtin_nontemporal_load(o + e * 256);
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] += v[e];
  }
}

template <bool RELU>
__global__ __launch_bounds__(256) void sgemm_tn_splitk(const bf16_t *__restrict__ X, const bf16_t *__restrict__ W,
                                                       const bf16_t *__restrict__ bias, bf16_t *__restrict__ Y, float *__restrict__ ws,
                                                       int *__restrict__ tickets, int M, int N, int K, int ldx, int ldw, int ldy)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char sg_smem[];
  bf16_t(*Xs)[PK256] = reinterpret_cast<bf16_t(*)[PK256]>(sg_smem);                       // [32][PK256]
  bf16_t(*Ws)[PK256] = reinterpret_cast<bf16_t(*)[PK256]>(sg_smem + 32 * PK256 * sizeof(bf16_t));   // [128][PK256]
  const int nb = blockIdx.x * 128, mb = blockIdx.y * 32, k0 = blockIdx.z * 256;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int prow = tid >> 5, pcol = (tid & 31) * 8;
  uint4 xr[4], wr[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) xr[i] = load_piece(X, ldx, mb + prow + 8 * i, k0 + pcol, M, K);
#pragma unroll
  for (int i = 0; i < 16; ++i) wr[i] = load_piece(W, ldw, nb + prow + 8 * i, k0 + pcol, N, K);
#pragma unroll
  for (int i = 0; i < 4; ++i) store_piece(&Xs[prow + 8 * i][pcol], xr[i]);
#pragma unroll
  for (int i = 0; i < 16; ++i) store_piece(&Ws[prow + 8 * i][pcol], wr[i]);
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int s = 0; s < 32; ++s) mma(acc, lds4(&Ws[32 * wave + r][8 * s + 4 * hh]), lds4(&Xs[r][8 * s + 4 * hh]));
  const int tiles = gridDim.x * gridDim.y, tile = blockIdx.y * gridDim.x + blockIdx.x, slices = gridDim.z;
  if (!splitk_arrive(ws, tickets, tile, tiles, slices, acc)) return;
  splitk_sum(ws, tile, tiles, slices, acc);
  const int m = mb + r;
  if (m >= M) return;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n0 = nb + 32 * wave + 8 * g + 4 * hh;
    if (n0 >= N) continue;
    float v[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    if (bias) {
      const uint2 b = *reinterpret_cast<const uint2 *>(bias + n0);
      v[0] += bf_lo(b.x); v[1] += bf_hi(b.x); v[2] += bf_lo(b.y); v[3] += bf_hi(b.y);
    }
    if (RELU) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
    }
    *reinterpret_cast<bf16x4 *>(Y + (int64_t)m * ldy + n0) = pack4(v[0], v[1], v[2], v[3]);
  }
}

// ------------------------------------------------------------------------------------------------ decoder prediction head
// decoder_norm + the 3-layer mask-embedding MLP of ONE prediction head in one launch (reference mask2former_transformer_decoder.py:
// 449-459 + :198-204; the per-layer mask prediction only feeds the next layer's attention mask and carries no gradient, :457):
//   d = LayerNorm(tgt) (fp32: written to dec_out, with its statistics for the backward pass of the gradient-carrying heads)
//   e = W3 relu(W2 relu(W1 bf16(d) + b1) + b2) + b3      (bf16 operands, fp32 accumulation, bf16 between the layers)
//   ef[b][q][:] = fp32(e[q B + b][:])                     (batch-major: the operand of the mask-logit product)
// A workgroup owns 32 rows; each layer's 256 x 256 weight passes through LDS with all its loads in flight.  Was 4 launches + a
// transposing copy per head.
__global__ __launch_bounds__(256) void decoder_head_256(const float *__restrict__ tgt, const float *__restrict__ ln_w, const float *__restrict__ ln_b,
                                                        float eps, const bf16_t *__restrict__ w1, const bf16_t *__restrict__ b1,
                                                        const bf16_t *__restrict__ w2, const bf16_t *__restrict__ b2,
                                                        const bf16_t *__restrict__ w3, const bf16_t *__restrict__ b3, float *__restrict__ dec_out,
                                                        float *__restrict__ mean, float *__restrict__ rstd, void *__restrict__ ef_, int ef_bf16,
                                                        int R, int B)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char sg_smem[];
  bf16_t(*Xs)[PK256] = reinterpret_cast<bf16_t(*)[PK256]>(sg_smem);                       // [32][PK256]
  bf16_t(*Ws)[PK256] = reinterpret_cast<bf16_t(*)[PK256]>(sg_smem + 32 * PK256 * sizeof(bf16_t));   // [256][PK256]
  const int mb = blockIdx.x * 32;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  // ---- LayerNorm: a wavefront per row (64 lanes x 4 channels)
  {
    const float4 gm = *reinterpret_cast<const float4 *>(ln_w + lane * 4), bt = *reinterpret_cast<const float4 *>(ln_b + lane * 4);
    for (int i = 0; i < 8; ++i) {
      const int row = wave * 8 + i, m = mb + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < R) v = *reinterpret_cast<const float4 *>(tgt + (int64_t)m * 256 + lane * 4);
      float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
      for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
      const float mu = s * (1.f / 256);
      v.x -= mu; v.y -= mu; v.z -= mu; v.w -= mu;
      float q = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
#pragma unroll
      for (int o = 32; o; o >>= 1) q += __shfl_xor(q, o, 64);
      const float rs = rsqrtf(q * (1.f / 256) + eps);
      float4 o4 = make_float4(v.x * rs * gm.x + bt.x, v.y * rs * gm.y + bt.y, v.z * rs * gm.z + bt.z, v.w * rs * gm.w + bt.w);
      if (m < R) {
        *reinterpret_cast<float4 *>(dec_out + (int64_t)m * 256 + lane * 4) = o4;
        if (lane == 0) { mean[m] = mu; rstd[m] = rs; }
      } else {
        o4 = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      *reinterpret_cast<bf16x4 *>(&Xs[row][lane * 4]) = pack4(o4.x, o4.y, o4.z, o4.w);
    }
  }
  if (!ef_) return;                                              // the last head: no mask prediction follows
  const int prow = tid >> 5, pcol = (tid & 31) * 8;
#pragma unroll 1
  for (int layer = 0; layer < 3; ++layer) {
    const bf16_t *W = layer == 0 ? w1 : layer == 1 ? w2 : w3;
    const bf16_t *bias = layer == 0 ? b1 : layer == 1 ? b2 : b3;
    {                                                            // the whole 128 KB weight in flight at once: 32 pieces of 16 bytes per thread
      uint4 wr[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) wr[i] = *reinterpret_cast<const uint4 *>(W + (int64_t)(prow + 8 * i) * 256 + pcol);
#pragma unroll
      for (int i = 0; i < 32; ++i) store_piece(&Ws[prow + 8 * i][pcol], wr[i]);
    }
    __syncthreads();                                             // Ws (and the Xs rows of the previous phase) complete
    f32x16 acc[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
    for (int s = 0; s < 32; ++s) {                               // acc[n][m] += W[n][k..] . X[m][k..]
      const bf16x4 x = lds4(&Xs[r][8 * s + 4 * hh]);
      mma(acc[0], lds4(&Ws[64 * wave + r][8 * s + 4 * hh]), x);
      mma(acc[1], lds4(&Ws[64 * wave + 32 + r][8 * s + 4 * hh]), x);
    }
    __syncthreads();                                             // every wave is done with Xs / Ws
    const int m = mb + r;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n0 = 64 * wave + 32 * blk + 8 * g + 4 * hh;
        const uint2 bb = *reinterpret_cast<const uint2 *>(bias + n0);
        float v[4] = {acc[blk][4 * g] + bf_lo(bb.x), acc[blk][4 * g + 1] + bf_hi(bb.x), acc[blk][4 * g + 2] + bf_lo(bb.y), acc[blk][4 * g + 3] + bf_hi(bb.y)};
        if (layer < 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
          *reinterpret_cast<bf16x4 *>(&Xs[r][n0]) = pack4(v[0], v[1], v[2], v[3]);     // the next layer's input (rounded like a bf16 Linear's output)
        } else if (m < R) {
          const int q = m / B, b = m - q * B, Q = R / B;         // rounded to bf16 like the Linear's output, then widened
          if (ef_bf16)                                           // the product that follows runs in bf16 (autocast): no widening, no cast launch
            *reinterpret_cast<bf16x4 *>(reinterpret_cast<bf16_t *>(ef_) + ((int64_t)b * Q + q) * 256 + n0) = pack4(v[0], v[1], v[2], v[3]);
          else
            *reinterpret_cast<float4 *>(reinterpret_cast<float *>(ef_) + ((int64_t)b * Q + q) * 256 + n0) =
                make_float4(bf_lo(pk_bf16(v[0], 0.f)), bf_lo(pk_bf16(v[1], 0.f)), bf_lo(pk_bf16(v[2], 0.f)), bf_lo(pk_bf16(v[3], 0.f)));
        }
      }
    // (the next layer's weight stores are followed by a barrier before anyone reads Xs again)
  }
}

// ------------------------------------------------------------------------------------------------ dX = dY W
template <bool ACC, bool MASK>
__global__ __launch_bounds__(256) void sgemm_nn(const bf16_t *__restrict__ dY, const bf16_t *__restrict__ W,
                                                const bf16_t *__restrict__ ref, bf16_t *__restrict__ dX, int M, int N, int K,
                                                int ldy, int ldw, int ldx)
{
  __shared__ __attribute__((aligned(16))) bf16_t Ys[2][32][P64];
  __shared__ __attribute__((aligned(16))) bf16_t Ws[2][64][P128];
  const int kb = blockIdx.x * 128, mb = blockIdx.y * 32;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int yrow = tid >> 3, ycol = (tid & 7) * 8;                          // dY chunk: 32 rows x 8 pieces
  const int wrow = tid >> 4, wcol = (tid & 15) * 8;                         // W chunk: 64 rows x 16 pieces, 4 per thread
  uint4 yr, wr[4];
  auto gload = [&](int n0) {
    yr = load_piece(dY, ldy, mb + yrow, n0 + ycol, M, N);
#pragma unroll
    for (int i = 0; i < 4; ++i) wr[i] = load_piece(W, ldw, n0 + wrow + 16 * i, kb + wcol, N, K);
  };
  auto lstore = [&](int buf) {
    store_piece(&Ys[buf][yrow][ycol], yr);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_piece(&Ws[buf][wrow + 16 * i][wcol], wr[i]);
  };
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int nchunk = N / 64;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunk) gload((c + 1) * 64);
#pragma unroll
    for (int s = 0; s < 8; ++s)                                             // acc[k][m] += W^T[k][n..] . dY[m][n..]
      mma(acc, gather4(&Ws[buf][8 * s + 4 * hh][32 * wave + r], P128), lds4(&Ys[buf][r][8 * s + 4 * hh]));
    if (c + 1 < nchunk) lstore(buf ^ 1);
    __syncthreads();
  }
  const int m = mb + r;
  if (m >= M) return;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int k0 = kb + 32 * wave + 8 * g + 4 * hh;
    if (k0 >= K) continue;
    float v[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    bf16_t *dst = dX + (int64_t)m * ldx + k0;
    if (ACC) {
      const uint2 o = *reinterpret_cast<const uint2 *>(dst);
      v[0] += bf_lo(o.x); v[1] += bf_hi(o.x); v[2] += bf_lo(o.y); v[3] += bf_hi(o.y);
    }
    if (MASK) {
      const uint2 h = *reinterpret_cast<const uint2 *>(ref + (int64_t)m * ldx + k0);
      if (!(bf_lo(h.x) > 0.f)) v[0] = 0.f;
      if (!(bf_hi(h.x) > 0.f)) v[1] = 0.f;
      if (!(bf_lo(h.y) > 0.f)) v[2] = 0.f;
      if (!(bf_hi(h.y) > 0.f)) v[3] = 0.f;
    }
    *reinterpret_cast<bf16x4 *>(dst) = pack4(v[0], v[1], v[2], v[3]);
  }
}

// dX = dY W for N <= 256 with the whole contraction staged at once (see sgemm_tn_k256)
constexpr int PN256 = 256 + 8;
template <bool ACC, bool MASK>
__global__ __launch_bounds__(256) void sgemm_nn_n256(const bf16_t *__restrict__ dY, const bf16_t *__restrict__ W,
                                                     const bf16_t *__restrict__ ref, bf16_t *__restrict__ dX, int M, int N, int K,
                                                     int ldy, int ldw, int ldx)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char sg_smem[];
  bf16_t(*Ys)[PN256] = reinterpret_cast<bf16_t(*)[PN256]>(sg_smem);                       // [32][PN256]
  bf16_t(*Ws)[P128] = reinterpret_cast<bf16_t(*)[P128]>(sg_smem + 32 * PN256 * sizeof(bf16_t));     // [256][P128]
  const int kb = blockIdx.x * 128, mb = blockIdx.y * 32;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int yrow = tid >> 5, ycol = (tid & 31) * 8;                          // dY: 8 rows x 32 pieces per pass
  const int wrow = tid >> 4, wcol = (tid & 15) * 8;                          // W: 16 rows x 16 pieces per pass
  uint4 yr[4], wr[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) yr[i] = load_piece(dY, ldy, mb + yrow + 8 * i, ycol, M, N);
#pragma unroll
  for (int i = 0; i < 16; ++i) wr[i] = load_piece(W, ldw, wrow + 16 * i, kb + wcol, N, K);
#pragma unroll
  for (int i = 0; i < 4; ++i) store_piece(&Ys[yrow + 8 * i][ycol], yr[i]);
#pragma unroll
  for (int i = 0; i < 16; ++i) store_piece(&Ws[wrow + 16 * i][wcol], wr[i]);
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int steps = N / 8;
  for (int s = 0; s < steps; ++s)                                           // acc[k][m] += W^T[k][n..] . dY[m][n..]
    mma(acc, gather4(&Ws[8 * s + 4 * hh][32 * wave + r], P128), lds4(&Ys[r][8 * s + 4 * hh]));
  const int m = mb + r;
  if (m >= M) return;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int k0 = kb + 32 * wave + 8 * g + 4 * hh;
    if (k0 >= K) continue;
    float v[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    bf16_t *dst = dX + (int64_t)m * ldx + k0;
    if (ACC) {
      const uint2 o = *reinterpret_cast<const uint2 *>(dst);
      v[0] += bf_lo(o.x); v[1] += bf_hi(o.x); v[2] += bf_lo(o.y); v[3] += bf_hi(o.y);
    }
    if (MASK) {
      const uint2 h = *reinterpret_cast<const uint2 *>(ref + (int64_t)m * ldx + k0);
      if (!(bf_lo(h.x) > 0.f)) v[0] = 0.f;
      if (!(bf_hi(h.x) > 0.f)) v[1] = 0.f;
      if (!(bf_lo(h.y) > 0.f)) v[2] = 0.f;
      if (!(bf_hi(h.y) > 0.f)) v[3] = 0.f;
    }
    *reinterpret_cast<bf16x4 *>(dst) = pack4(v[0], v[1], v[2], v[3]);
  }
}

// dX = dY W with a long contraction (N = 2048: the decoder FFN's first Linear, input gradient) cut into 256-wide slices of N, one
// all-loads-in-flight workgroup each; partial tiles and the last-arrival reduction as in sgemm_tn_splitk
template <bool ACC, bool MASK>
__global__ __launch_bounds__(256) void sgemm_nn_splitn(const bf16_t *__restrict__ dY, const bf16_t *__restrict__ W,
                                                       const bf16_t *__restrict__ ref, bf16_t *__restrict__ dX, float *__restrict__ ws,
                                                       int *__restrict__ tickets, int M, int N, int K, int ldy, int ldw, int ldx)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char sg_smem[];
  bf16_t(*Ys)[PN256] = reinterpret_cast<bf16_t(*)[PN256]>(sg_smem);                       // [32][PN256]
  bf16_t(*Ws)[P128] = reinterpret_cast<bf16_t(*)[P128]>(sg_smem + 32 * PN256 * sizeof(bf16_t));     // [256][P128]
  const int kb = blockIdx.x * 128, mb = blockIdx.y * 32, n0s = blockIdx.z * 256;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int yrow = tid >> 5, ycol = (tid & 31) * 8;
  const int wrow = tid >> 4, wcol = (tid & 15) * 8;
  uint4 yr[4], wr[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) yr[i] = load_piece(dY, ldy, mb + yrow + 8 * i, n0s + ycol, M, N);
#pragma unroll
  for (int i = 0; i < 16; ++i) wr[i] = load_piece(W, ldw, n0s + wrow + 16 * i, kb + wcol, N, K);
#pragma unroll
  for (int i = 0; i < 4; ++i) store_piece(&Ys[yrow + 8 * i][ycol], yr[i]);
#pragma unroll
  for (int i = 0; i < 16; ++i) store_piece(&Ws[wrow + 16 * i][wcol], wr[i]);
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int s = 0; s < 32; ++s) mma(acc, gather4(&Ws[8 * s + 4 * hh][32 * wave + r], P128), lds4(&Ys[r][8 * s + 4 * hh]));
  const int tiles = gridDim.x * gridDim.y, tile = blockIdx.y * gridDim.x + blockIdx.x, slices = gridDim.z;
  if (!splitk_arrive(ws, tickets, tile, tiles, slices, acc)) return;
  splitk_sum(ws, tile, tiles, slices, acc);
  const int m = mb + r;
  if (m >= M) return;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int k0 = kb + 32 * wave + 8 * g + 4 * hh;
    if (k0 >= K) continue;
    float v[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    bf16_t *dst = dX + (int64_t)m * ldx + k0;
    if (ACC) {
      const uint2 o = *reinterpret_cast<const uint2 *>(dst);
      v[0] += bf_lo(o.x); v[1] += bf_hi(o.x); v[2] += bf_lo(o.y); v[3] += bf_hi(o.y);
    }
    if (MASK) {
      const uint2 h = *reinterpret_cast<const uint2 *>(ref + (int64_t)m * ldx + k0);
      if (!(bf_lo(h.x) > 0.f)) v[0] = 0.f;
      if (!(bf_hi(h.x) > 0.f)) v[1] = 0.f;
      if (!(bf_lo(h.y) > 0.f)) v[2] = 0.f;
      if (!(bf_hi(h.y) > 0.f)) v[3] = 0.f;
    }
    *reinterpret_cast<bf16x4 *>(dst) = pack4(v[0], v[1], v[2], v[3]);
  }
}

// ------------------------------------------------------------------------------------------------ dW = dY^T X
__device__ __forceinline__ void sgemm_wgrad_body(const bf16_t *__restrict__ dY, const bf16_t *__restrict__ X,
                                                 bf16_t *__restrict__ dW, float *__restrict__ dB, int M, int N, int K, int ldy,
                                                 int ldx, int ldw, int bx, int by)
{
  __shared__ __attribute__((aligned(16))) bf16_t Ys[2][64][P32];
  __shared__ __attribute__((aligned(16))) bf16_t Xs[2][64][P128];
  const int kb = bx * 128, nb = by * 32;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int yrow = tid >> 2, ycol = (tid & 3) * 8;                          // dY chunk: 64 rows x 4 pieces
  const int xrow = tid >> 4, xcol = (tid & 15) * 8;                         // X chunk: 64 rows x 16 pieces, 4 per thread
  uint4 yr, xr[4];
  auto gload = [&](int m0) {
    yr = load_piece(dY, ldy, m0 + yrow, nb + ycol, M, N);
#pragma unroll
    for (int i = 0; i < 4; ++i) xr[i] = load_piece(X, ldx, m0 + xrow + 16 * i, kb + xcol, M, K);
  };
  auto lstore = [&](int buf) {
    store_piece(&Ys[buf][yrow][ycol], yr);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_piece(&Xs[buf][xrow + 16 * i][xcol], xr[i]);
  };
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const bool do_bias = dB != nullptr && bx == 0 && wave == 0;              // the k-tile-0 workgroup also sums its dY columns
  float bsum = 0.f;
  const int nchunk = (M + 63) / 64;
  if (nchunk > 0) { gload(0); lstore(0); }
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunk) gload((c + 1) * 64);
#pragma unroll
    for (int s = 0; s < 8; ++s)                                             // acc[n][k] += dY^T[n][m..] . X^T[k][m..]
      mma(acc, gather4(&Ys[buf][8 * s + 4 * hh][r], P32), gather4(&Xs[buf][8 * s + 4 * hh][32 * wave + r], P128));
    if (do_bias) {
#pragma unroll 8
      for (int i = 0; i < 32; ++i) bsum += __uint_as_float((unsigned)Ys[buf][32 * hh + i][r] << 16);
    }
    if (c + 1 < nchunk) lstore(buf ^ 1);
    __syncthreads();
  }
  if (do_bias) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (hh == 0 && nb + r < N) dB[nb + r] = bsum;
  }
  const int k = kb + 32 * wave + r;
  if (k >= K) return;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int n = nb + (e & 3) + 8 * (e >> 2) + 4 * hh;
    if (n < N) dW[(int64_t)n * ldw + k] = (bf16_t)(pk_bf16(acc[e], 0.f) & 0xffffu);
  }
}

__global__ __launch_bounds__(256) void sgemm_wgrad(const bf16_t *__restrict__ dY, const bf16_t *__restrict__ X,
                                                   bf16_t *__restrict__ dW, float *__restrict__ dB, int M, int N, int K, int ldy,
                                                   int ldx, int ldw)
{
  sgemm_wgrad_body(dY, X, dW, dB, M, N, K, ldy, ldx, ldw, blockIdx.x, blockIdx.y);
}

// GROUPED: the ~70 weight gradients of the decoder's backward pass (each a 7 us launch over 200 rows) as ONE launch over a table
struct SgWgradProblem {
  const bf16_t *dY, *X;
  bf16_t *dW;
  float *dB;
  int M, N, K, ldy, ldx, ldw, gx, block_begin;
};
__global__ __launch_bounds__(256) void sgemm_wgrad_grouped(const SgWgradProblem *__restrict__ tab, int count)
{
  int p = 0;
  while (p + 1 < count && tab[p + 1].block_begin <= (int)blockIdx.x) ++p;
  const SgWgradProblem &q = tab[p];
  const int local = blockIdx.x - q.block_begin;
  sgemm_wgrad_body(pd_as_global(q.dY), pd_as_global(q.X), pd_as_global(q.dW), pd_as_global(q.dB), q.M, q.N, q.K, q.ldy, q.ldx, q.ldw, local % q.gx, local / q.gx);   // (pd_common.h: table pointers would be FLAT)
}

// ------------------------------------------------------------------------------------------------ dW = dY^T X, many rows
// The same 32 x 128 tile and 64-row chunks as sgemm_wgrad, but the M rows (10^4..10^5 tokens of a Swin stage) are cut into
// `gridDim.z` slices: an [N, K] weight gradient has only (N/32)(K/128) ~ 100-200 tiles, too few for 256 CUs when each tile
// must walk all the rows (the library picks such a kernel: 63 us for M = 8192, N = 1536, K = 512).  Each slice writes
// its fp32 partial tile (plain stores) and wgrad_reduce sums the slices -> bf16: deterministic, no atomics.
template <int NB>   // 32-row blocks of N per workgroup: 2 (a 64 x 128 tile, the X operand of an MFMA pair shared) or 1
__global__ __launch_bounds__(256) void sgemm_wgrad_split(const bf16_t *__restrict__ dY, const bf16_t *__restrict__ X,
                                                         float *__restrict__ part, float *__restrict__ bpart, int M, int N, int K,
                                                         int ldy, int ldx, int rows_per_split)
{
  constexpr int TN = 32 * NB, PY = NB == 2 ? P64 : P32;
  __shared__ __attribute__((aligned(16))) bf16_t Ys[2][64][PY];
  __shared__ __attribute__((aligned(16))) bf16_t Xs[2][64][P128];
  const int kb = blockIdx.x * 128, nb = blockIdx.y * TN, z = blockIdx.z;
  const int m_begin = z * rows_per_split, m_end = min(M, m_begin + rows_per_split);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int yrow = NB == 2 ? tid >> 3 : tid >> 2, ycol = NB == 2 ? (tid & 7) * 8 : (tid & 3) * 8;     // dY chunk: 64 rows x TN/8 pieces
  const int xrow = tid >> 4, xcol = (tid & 15) * 8;
  uint4 yr[NB], xr[4];
  auto gload = [&](int m0) {
#pragma unroll
    for (int i = 0; i < NB; ++i) yr[i] = load_piece(dY, ldy, m0 + yrow + 32 * i * (NB == 2), nb + ycol, m_end, N);
#pragma unroll
    for (int i = 0; i < 4; ++i) xr[i] = load_piece(X, ldx, m0 + xrow + 16 * i, kb + xcol, m_end, K);
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NB; ++i) store_piece(&Ys[buf][yrow + 32 * i * (NB == 2)][ycol], yr[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_piece(&Xs[buf][xrow + 16 * i][xcol], xr[i]);
  };
  f32x16 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
  const bool do_bias = bpart != nullptr && blockIdx.x == 0 && wave < NB;      // wave b sums the dY columns of n-block b
  float bsum = 0.f;
  const int nchunk = (m_end - m_begin + 63) / 64;
  if (nchunk > 0) { gload(m_begin); lstore(0); }
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunk) gload(m_begin + (c + 1) * 64);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const bf16x4 xo = gather4(&Xs[buf][8 * s + 4 * hh][32 * wave + r], P128);
#pragma unroll
      for (int b = 0; b < NB; ++b) mma(acc[b], gather4(&Ys[buf][8 * s + 4 * hh][32 * b + r], PY), xo);
    }
    if (do_bias) {
#pragma unroll 8
      for (int i = 0; i < 32; ++i) bsum += __uint_as_float((unsigned)Ys[buf][32 * hh + i][32 * wave + r] << 16);
    }
    if (c + 1 < nchunk) lstore(buf ^ 1);
    __syncthreads();
  }
  if (do_bias) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (hh == 0 && nb + 32 * wave + r < N) bpart[(int64_t)z * N + nb + 32 * wave + r] = bsum;
  }
  const int k = kb + 32 * wave + r;
  if (k >= K) return;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int n = nb + 32 * b + (e & 3) + 8 * (e >> 2) + 4 * hh;
      if (n < N) part[((int64_t)z * N + n) * K + k] = acc[b][e];
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce(const float *__restrict__ part, const float *__restrict__ bpart,
                                                    bf16_t *__restrict__ dW, float *__restrict__ dB, int N, int K, int ldw, int splits)
{
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, nk4 = (int64_t)N * K / 4;
  if (i < nk4) {
    float4 a = reinterpret_cast<const float4 *>(part)[i];
    for (int z = 1; z < splits; ++z) {
      const float4 b = reinterpret_cast<const float4 *>(part)[z * nk4 + i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const int n = (int)(i * 4 / K), k = (int)(i * 4 - (int64_t)n * K);
    uint2 o;
    o.x = pk_bf16(a.x, a.y); o.y = pk_bf16(a.z, a.w);
    *reinterpret_cast<uint2 *>(dW + (int64_t)n * ldw + k) = o;
  }
  if (dB && i < N) {
    float b = 0.f;
    for (int z = 0; z < splits; ++z) b += bpart[(int64_t)z * N + i];
    dB[i] = b;
  }
}

// slices of the M rows: ~1024 workgroups in flight, at least 256 rows per slice, slices a multiple of the 64-row chunk
int wgrad_tile_n(int N) { return N % 64 == 0 ? 64 : 32; }
int wgrad_splits(int M, int N, int K, int *rows_per_split)
{
  const int tn = wgrad_tile_n(N);
  const int tiles = ((N + tn - 1) / tn) * ((K + 127) / 128);
  int splits = (1024 + tiles - 1) / tiles;
  const int max_splits = (M + 255) / 256;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int rps = (M + splits - 1) / splits;
  rps = (rps + 63) / 64 * 64;
  *rows_per_split = rps;
  return (M + rps - 1) / rps;
}

bool ok_ptr(const void *p) { return ((uintptr_t)p & 15) == 0; }

int check_common(const char *who, const void *a, const void *b, const void *c, int M, int N, int K, int l0, int l1, int l2)
{
  if (M < 0 || N < 0 || K < 0) return pd_set_error(PD_ERR_INVALID_ARG, "%s: negative size", who);
  if (M > 0 && N > 0 && K > 0 && (!a || !b || !c)) return pd_set_error(PD_ERR_INVALID_ARG, "%s: null pointer", who);
  if ((l0 & 7) || (l1 & 7) || (l2 & 7) || !ok_ptr(a) || !ok_ptr(b) || !ok_ptr(c))
    return pd_set_error(PD_ERR_INVALID_ARG, "%s: leading dimensions must be multiples of 8 and bases 16-byte aligned", who);
  return PD_OK;
}

}  // namespace

int g_pd_dbg_sgemm_deep = 1;   // tools/ only (pd_debug_set "sgemm_deep"): 0 = the chunked kernels also for K / N <= 256

extern "C" int pd_sgemm_tn_bf16(const void *X, const void *W, const void *bias, void *Y, int M, int N, int K, int ldx, int ldw,
                                int ldy, int relu, void *stream_)
{
  int rc = check_common("pd_sgemm_tn_bf16", X, W, Y, M, N, K, ldx, ldw, ldy);
  if (rc) return rc;
  if ((K % 64) || (N & 3)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_sgemm_tn_bf16: K=%d must be a multiple of 64 and N=%d of 4", K, N);
  if (bias && ((uintptr_t)bias & 7)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_sgemm_tn_bf16: bias must be 8-byte aligned");
  if (M == 0 || N == 0) return PD_OK;
  const dim3 g((N + 127) / 128, (M + 31) / 32), b(256);
  hipStream_t s = (hipStream_t)stream_;
  if (K <= 256 && g_pd_dbg_sgemm_deep) {                 // whole contraction in flight at once (latency-bound products)
    constexpr size_t lds = (size_t)(32 + 128) * PK256 * sizeof(bf16_t);
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void *)sgemm_tn_k256<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void *)sgemm_tn_k256<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr = true;
    }
    if (relu) hipLaunchKernelGGL(sgemm_tn_k256<true>, g, b, lds, s, (const bf16_t *)X, (const bf16_t *)W, (const bf16_t *)bias, (bf16_t *)Y, M, N, K, ldx, ldw, ldy);
    else hipLaunchKernelGGL(sgemm_tn_k256<false>, g, b, lds, s, (const bf16_t *)X, (const bf16_t *)W, (const bf16_t *)bias, (bf16_t *)Y, M, N, K, ldx, ldw, ldy);
    return pd_check_launch("pd_sgemm_tn_bf16");
  }
  if (relu) hipLaunchKernelGGL(sgemm_tn<true>, g, b, 0, s, (const bf16_t *)X, (const bf16_t *)W, (const bf16_t *)bias, (bf16_t *)Y, M, N, K, ldx, ldw, ldy);
  else hipLaunchKernelGGL(sgemm_tn<false>, g, b, 0, s, (const bf16_t *)X, (const bf16_t *)W, (const bf16_t *)bias, (bf16_t *)Y, M, N, K, ldx, ldw, ldy);
  return pd_check_launch("pd_sgemm_tn_bf16");
}

// `batch` equally shaped products Y_b = X_b W_b^T in ONE launch (K <= 256): the Hungarian matcher's point logits of all (image, head)
// problems, [Q, C] x [points, C]^T each (reference matcher.py:108-125 through the linearity of point sampling) — was torch.bmm
extern "C" int pd_sgemm_tn_batched_bf16(const void *X, const void *W, void *Y, int M, int N, int K, int ldx, int ldw, int ldy, int batch,
                                        int64_t stride_x, int64_t stride_w, int64_t stride_y, void *stream_)
{
  int rc = check_common("pd_sgemm_tn_batched_bf16", X, W, Y, M, N, K, ldx, ldw, ldy);
  if (rc) return rc;
  if (K > 256 || (K % 64) || (N & 3) || batch < 0 || batch > 65535 || ((stride_x | stride_w) & 7) || (stride_y & 3))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_sgemm_tn_batched_bf16: K=%d (a multiple of 64, <= 256), N=%d (%% 4), batch=%d, strides %% 8 / 4", K, N, batch);
  if (M == 0 || N == 0 || batch == 0) return PD_OK;
  constexpr size_t lds = (size_t)(32 + 128) * PK256 * sizeof(bf16_t);
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void *)sgemm_tn_k256<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
  hipLaunchKernelGGL(sgemm_tn_k256<false>, dim3((N + 127) / 128, (M + 31) / 32, batch), dim3(256), lds, (hipStream_t)stream_, (const bf16_t *)X, (const bf16_t *)W,
                     (const bf16_t *)nullptr, (bf16_t *)Y, M, N, K, ldx, ldw, ldy, stride_x, stride_w, stride_y);
  return pd_check_launch("pd_sgemm_tn_batched_bf16");
}

extern "C" int pd_sgemm_tn_multi_bf16(const PdSgemmTnDesc *d, int count, int K, void *stream_)
{
  if (count <= 0 || count > 4 || !d) return pd_set_error(PD_ERR_INVALID_ARG, "pd_sgemm_tn_multi_bf16: count=%d (1..4)", count);
  if (K <= 0 || K > 256 || (K % 64)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_sgemm_tn_multi_bf16: K=%d must be a multiple of 64, <= 256", K);
  SgTnMulti p;
  memset(&p, 0, sizeof(p));
  int maxm = 0, maxn = 0;
  for (int i = 0; i < count; ++i) {
    int rc = check_common("pd_sgemm_tn_multi_bf16", d[i].X, d[i].W, d[i].Y, d[i].M, d[i].N, K, d[i].ldx, d[i].ldw, d[i].ldy);
    if (rc) return rc;
    if ((d[i].N & 3) || (d[i].bias && ((uintptr_t)d[i].bias & 7))) return pd_set_error(PD_ERR_INVALID_ARG, "pd_sgemm_tn_multi_bf16: N %% 4 / bias alignment");
    p.X[i] = (const bf16_t *)d[i].X; p.W[i] = (const bf16_t *)d[i].W; p.bias[i] = (const bf16_t *)d[i].bias; p.Y[i] = (bf16_t *)d[i].Y;
    p.M[i] = d[i].M; p.N[i] = d[i].N; p.ldx[i] = d[i].ldx; p.ldw[i] = d[i].ldw; p.ldy[i] = d[i].ldy;
    maxm = d[i].M > maxm ? d[i].M : maxm; maxn = d[i].N > maxn ? d[i].N : maxn;
  }
  if (maxm == 0 || maxn == 0) return PD_OK;
  constexpr size_t lds = (size_t)(32 + 128) * PK256 * sizeof(bf16_t);
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void *)sgemm_tn_k256_multi, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
  hipLaunchKernelGGL(sgemm_tn_k256_multi, dim3((maxn + 127) / 128, (maxm + 31) / 32, count), dim3(256), lds, (hipStream_t)stream_, p, K);
  return pd_check_launch("pd_sgemm_tn_multi_bf16");
}

extern "C" int pd_decoder_head_bf16(const float *tgt, const float *ln_w, const float *ln_b, float eps, const void *w1, const void *b1, const void *w2,
                                    const void *b2, const void *w3, const void *b3, float *dec_out, float *mean, float *rstd, void *ef, int ef_bf16,
                                    int R, int B, int C, void *stream_)
{
  if (R < 0 || B <= 0 || C != 256 || (R % B)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_decoder_head_bf16: R=%d B=%d C=%d (C must be 256, R a multiple of B)", R, B, C);
  if (R == 0) return PD_OK;
  if (!tgt || !ln_w || !ln_b || !dec_out || !mean || !rstd || (ef && (!w1 || !b1 || !w2 || !b2 || !w3 || !b3)))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_decoder_head_bf16: null pointer");
  constexpr size_t lds = (size_t)(32 + 256) * PK256 * sizeof(bf16_t);
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void *)decoder_head_256, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
  hipLaunchKernelGGL(decoder_head_256, dim3((R + 31) / 32), dim3(256), lds, (hipStream_t)stream_, tgt, ln_w, ln_b, eps, (const bf16_t *)w1,
                     (const bf16_t *)b1, (const bf16_t *)w2, (const bf16_t *)b2, (const bf16_t *)w3, (const bf16_t *)b3, dec_out, mean, rstd, ef, ef_bf16, R, B);
  return pd_check_launch("pd_decoder_head_bf16");
}

extern "C" int64_t pd_sgemm_split_workspace_floats(int M, int out_cols, int contraction)
{
  if (M <= 0 || out_cols <= 0 || contraction <= 0) return 0;
  return (int64_t)((contraction + 255) / 256) * ((M + 31) / 32) * ((out_cols + 127) / 128) * (32 * 128);
}
extern "C" int64_t pd_sgemm_split_tickets(int M, int out_cols) { return (M <= 0 || out_cols <= 0) ? 0 : (int64_t)((M + 31) / 32) * ((out_cols + 127) / 128); }

extern "C" int pd_sgemm_tn_splitk_bf16(const void *X, const void *W, const void *bias, void *Y, float *workspace, int64_t workspace_floats,
                                       int *tickets, int M, int N, int K, int ldx, int ldw, int ldy, int relu, void *stream_)
{
  int rc = check_common("pd_sgemm_tn_splitk_bf16", X, W, Y, M, N, K, ldx, ldw, ldy);
  if (rc) return rc;
  if ((K % 256) || K < 512 || (N & 3)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_sgemm_tn_splitk_bf16: K=%d must be a multiple of 256, >= 512, and N=%d of 4", K, N);
  if (bias && ((uintptr_t)bias & 7)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_sgemm_tn_splitk_bf16: bias must be 8-byte aligned");
  if (M == 0 || N == 0) return PD_OK;
  if (!workspace || !tickets || workspace_floats < pd_sgemm_split_workspace_floats(M, N, K))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_sgemm_tn_splitk_bf16: workspace / tickets missing or too small");
  const dim3 g((N + 127) / 128, (M + 31) / 32, K / 256), b(256);
  constexpr size_t lds = (size_t)(32 + 128) * PK256 * sizeof(bf16_t);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)sgemm_tn_splitk<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void *)sgemm_tn_splitk<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipStream_t s = (hipStream_t)stream_;
  if (relu) hipLaunchKernelGGL(sgemm_tn_splitk<true>, g, b, lds, s, (const bf16_t *)X, (const bf16_t *)W, (const bf16_t *)bias, (bf16_t *)Y, workspace, tickets, M, N, K, ldx, ldw, ldy);
  else hipLaunchKernelGGL(sgemm_tn_splitk<false>, g, b, lds, s, (const bf16_t *)X, (const bf16_t *)W, (const bf16_t *)bias, (bf16_t *)Y, workspace, tickets, M, N, K, ldx, ldw, ldy);
  return pd_check_launch("pd_sgemm_tn_splitk_bf16");
}

extern "C" int pd_sgemm_nn_splitn_bf16(const void *dY, const void *W, const void *relu_ref, void *dX, float *workspace, int64_t workspace_floats,
                                       int *tickets, int M, int N, int K, int ldy, int ldw, int ldx, int accumulate, void *stream_)
{
  int rc = check_common("pd_sgemm_nn_splitn_bf16", dY, W, dX, M, N, K, ldy, ldw, ldx);
  if (rc) return rc;
  if ((N % 256) || N < 512 || (K & 3)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_sgemm_nn_splitn_bf16: N=%d must be a multiple of 256, >= 512, and K=%d of 4", N, K);
  if (M == 0 || K == 0) return PD_OK;
  if (!workspace || !tickets || workspace_floats < pd_sgemm_split_workspace_floats(M, K, N))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_sgemm_nn_splitn_bf16: workspace / tickets missing or too small");
  const dim3 g((K + 127) / 128, (M + 31) / 32, N / 256), b(256);
  constexpr size_t lds = (size_t)32 * PN256 * sizeof(bf16_t) + (size_t)256 * P128 * sizeof(bf16_t);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)sgemm_nn_splitn<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void *)sgemm_nn_splitn<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void *)sgemm_nn_splitn<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void *)sgemm_nn_splitn<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipStream_t s = (hipStream_t)stream_;
#define LAUNCHS(A, R) hipLaunchKernelGGL((sgemm_nn_splitn<A, R>), g, b, lds, s, (const bf16_t *)dY, (const bf16_t *)W, (const bf16_t *)relu_ref, (bf16_t *)dX, workspace, tickets, M, N, K, ldy, ldw, ldx)
  if (accumulate) { if (relu_ref) LAUNCHS(true, true); else LAUNCHS(true, false); }
  else { if (relu_ref) LAUNCHS(false, true); else LAUNCHS(false, false); }
#undef LAUNCHS
  return pd_check_launch("pd_sgemm_nn_splitn_bf16");
}

extern "C" int pd_sgemm_nn_bf16(const void *dY, const void *W, const void *relu_ref, void *dX, int M, int N, int K, int ldy,
                                int ldw, int ldx, int accumulate, void *stream_)
{
  int rc = check_common("pd_sgemm_nn_bf16", dY, W, dX, M, N, K, ldy, ldw, ldx);
  if (rc) return rc;
  if ((N % 64) || (K & 3)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_sgemm_nn_bf16: N=%d must be a multiple of 64 and K=%d of 4", N, K);
  if (M == 0 || K == 0) return PD_OK;
  const dim3 g((K + 127) / 128, (M + 31) / 32), b(256);
  hipStream_t s = (hipStream_t)stream_;
  if (N <= 256 && g_pd_dbg_sgemm_deep) {
    constexpr size_t lds = (size_t)32 * PN256 * sizeof(bf16_t) + (size_t)256 * P128 * sizeof(bf16_t);
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void *)sgemm_nn_n256<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void *)sgemm_nn_n256<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void *)sgemm_nn_n256<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void *)sgemm_nn_n256<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr = true;
    }
#define LAUNCHD(A, R) hipLaunchKernelGGL((sgemm_nn_n256<A, R>), g, b, lds, s, (const bf16_t *)dY, (const bf16_t *)W, (const bf16_t *)relu_ref, (bf16_t *)dX, M, N, K, ldy, ldw, ldx)
    if (accumulate) { if (relu_ref) LAUNCHD(true, true); else LAUNCHD(true, false); }
    else { if (relu_ref) LAUNCHD(false, true); else LAUNCHD(false, false); }
#undef LAUNCHD
    return pd_check_launch("pd_sgemm_nn_bf16");
  }
#define LAUNCH(A, R) hipLaunchKernelGGL((sgemm_nn<A, R>), g, b, 0, s, (const bf16_t *)dY, (const bf16_t *)W, (const bf16_t *)relu_ref, (bf16_t *)dX, M, N, K, ldy, ldw, ldx)
  if (accumulate) { if (relu_ref) LAUNCH(true, true); else LAUNCH(true, false); }
  else { if (relu_ref) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
  return pd_check_launch("pd_sgemm_nn_bf16");
}

extern "C" int pd_sgemm_wgrad_bf16(const void *dY, const void *X, void *dW, float *dB, int M, int N, int K, int ldy, int ldx,
                                   int ldw, void *stream_)
{
  int rc = check_common("pd_sgemm_wgrad_bf16", M > 0 ? dY : dW, M > 0 ? X : dW, dW, M, N, K, ldy, ldx, 8);
  if (rc) return rc;
  if ((N & 3) || (K & 3)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_sgemm_wgrad_bf16: N=%d and K=%d must be multiples of 4", N, K);
  if (N == 0 || K == 0) return PD_OK;
  const dim3 g((K + 127) / 128, (N + 31) / 32), b(256);
  hipLaunchKernelGGL(sgemm_wgrad, g, b, 0, (hipStream_t)stream_, (const bf16_t *)dY, (const bf16_t *)X, (bf16_t *)dW, dB, M, N, K, ldy, ldx, ldw);
  return pd_check_launch("pd_sgemm_wgrad_bf16");
}

extern "C" int64_t pd_sgemm_wgrad_split_workspace(int M, int N, int K)
{
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int rps;
  const int splits = wgrad_splits(M, N, K, &rps);
  return (int64_t)splits * N * K + (int64_t)splits * N;
}

extern "C" int pd_sgemm_wgrad_split_bf16(const void *dY, const void *X, void *dW, float *dB, float *workspace, int M, int N, int K,
                                         int ldy, int ldx, int ldw, void *stream_)
{
  int rc = check_common("pd_sgemm_wgrad_split_bf16", M > 0 ? dY : dW, M > 0 ? X : dW, dW, M, N, K, ldy, ldx, 8);
  if (rc) return rc;
  if ((N & 3) || (K & 3) || (ldw & 3)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_sgemm_wgrad_split_bf16: N=%d, K=%d, ldw=%d must be multiples of 4", N, K, ldw);
  if (N == 0 || K == 0) return PD_OK;
  if (M <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_sgemm_wgrad_split_bf16: M must be positive (pd_sgemm_wgrad_bf16 handles M = 0)");
  if (!workspace || ((uintptr_t)workspace & 15)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_sgemm_wgrad_split_bf16: workspace null / not 16-byte aligned");
  int rps;
  const int splits = wgrad_splits(M, N, K, &rps);
  float *part = workspace, *bpart = workspace + (int64_t)splits * N * K;
  hipStream_t st = (hipStream_t)stream_;
  if (wgrad_tile_n(N) == 64)
    hipLaunchKernelGGL(sgemm_wgrad_split<2>, dim3((K + 127) / 128, N / 64, splits), dim3(256), 0, st, (const bf16_t *)dY, (const bf16_t *)X,
                       part, dB ? bpart : nullptr, M, N, K, ldy, ldx, rps);
  else
    hipLaunchKernelGGL(sgemm_wgrad_split<1>, dim3((K + 127) / 128, (N + 31) / 32, splits), dim3(256), 0, st, (const bf16_t *)dY, (const bf16_t *)X,
                       part, dB ? bpart : nullptr, M, N, K, ldy, ldx, rps);
  const int64_t work = (int64_t)N * K / 4 > N ? (int64_t)N * K / 4 : N;
  hipLaunchKernelGGL(wgrad_reduce, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, part, bpart, (bf16_t *)dW, dB, N, K, ldw, splits);
  return pd_check_launch("pd_sgemm_wgrad_split_bf16");
}


extern "C" int64_t pd_sgemm_wgrad_grouped_table_bytes(int count) { return (int64_t)count * (int64_t)sizeof(SgWgradProblem); }

extern "C" int pd_sgemm_wgrad_grouped_bf16(const PdSgemmWgradDesc *descs, int count, void *table_host_pinned, void *table_device,
                                           void *stream_)
{
  if (count < 0 || (count > 0 && (!descs || !table_host_pinned || !table_device)))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_sgemm_wgrad_grouped_bf16: null pointer / negative count");
  if (count == 0) return PD_OK;
  SgWgradProblem *tab = reinterpret_cast<SgWgradProblem *>(table_host_pinned);
  int blocks = 0, n = 0;
  for (int i = 0; i < count; ++i) {
    const PdSgemmWgradDesc &d = descs[i];
    int rc = check_common("pd_sgemm_wgrad_grouped_bf16", d.M > 0 ? d.dY : d.dW, d.M > 0 ? d.X : d.dW, d.dW, d.M, d.N, d.K, d.ldy, d.ldx, 8);
    if (rc) return rc;
    if ((d.N & 3) || (d.K & 3)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_sgemm_wgrad_grouped_bf16: N=%d and K=%d must be multiples of 4", d.N, d.K);
    if (d.N == 0 || d.K == 0) continue;
    SgWgradProblem &q = tab[n++];
    q.dY = (const bf16_t *)d.dY; q.X = (const bf16_t *)d.X; q.dW = (bf16_t *)d.dW; q.dB = d.dB;
    q.M = d.M; q.N = d.N; q.K = d.K; q.ldy = d.ldy; q.ldx = d.ldx; q.ldw = d.ldw;
    q.gx = (d.K + 127) / 128; q.block_begin = blocks;
    blocks += q.gx * ((d.N + 31) / 32);
  }
  if (n == 0) return PD_OK;
  hipStream_t st = (hipStream_t)stream_;
  if (hipMemcpyAsync(table_device, table_host_pinned, (size_t)n * sizeof(SgWgradProblem), hipMemcpyHostToDevice, st) != hipSuccess)
    return pd_set_error(PD_ERR_LAUNCH, "pd_sgemm_wgrad_grouped_bf16: table upload failed");
  hipLaunchKernelGGL(sgemm_wgrad_grouped, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const SgWgradProblem *>(table_device), n);
  return pd_check_launch("pd_sgemm_wgrad_grouped_bf16");
}
