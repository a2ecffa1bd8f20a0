// Device K-means (Lloyd) iteration, batched over images (C-ABI in include/pd_kmeans.h).
// assign: one wavefront per point, lanes across channels (16-byte pieces); the K <= 4 dot products are 64-lane DPP /
// shuffle reductions; every wave keeps the sums of the points it assigned in registers (K x C/64 floats per lane), the four
// waves of a workgroup are merged through LDS and leave as one set of atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pd_common.h"
#include "pd_kmeans.h"
#include "pd_msda.h"

namespace {

constexpr int KMAX = 4;       // centres per image
constexpr int PMAX = 8;       // 16-byte pieces per lane: C <= 2048

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void kmeans_assign(const float *__restrict__ X, const int32_t *__restrict__ blocks,
                                                     const float *__restrict__ centers, const float *__restrict__ cnorm,
                                                     const int32_t *__restrict__ done, int32_t *__restrict__ labels,
                                                     float *__restrict__ sums, float *__restrict__ counts,
                                                     int32_t *__restrict__ changed, int C, int K)
{
  extern __shared__ __attribute__((aligned(16))) float red[];                       // [K][C] partial sums of the workgroup + K counts + 1 changed
  const int b = blocks[blockIdx.x * 3], first = blocks[blockIdx.x * 3 + 1], npts = blocks[blockIdx.x * 3 + 2];
  if (done[b]) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int npiece = (C + 255) / 256;
  const float *cb = centers + (int64_t)b * K * C;
  float4 acc[KMAX][PMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
#pragma unroll
    for (int j = 0; j < PMAX; ++j) acc[k][j] = make_float4(0.f, 0.f, 0.f, 0.f);
  float cnt[KMAX] = {0.f, 0.f, 0.f, 0.f};
  int nchanged = 0;
  for (int p = wave; p < npts; p += 4) {
    const int n = first + p;
    const float *x = X + (int64_t)n * C;
    float4 xv[PMAX];
    float dot[KMAX] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < PMAX; ++j) {
      const int c = j * 256 + lane * 4;
      xv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < npiece && c < C) {
        xv[j] = *reinterpret_cast<const float4 *>(x + c);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
          if (k < K) {
            const float4 cv = *reinterpret_cast<const float4 *>(cb + (int64_t)k * C + c);
            dot[k] += xv[j].x * cv.x + xv[j].y * cv.y + xv[j].z * cv.z + xv[j].w * cv.w;
          }
        }
      }
    }
    float best = INFINITY;
    int arg = 0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k < K) {
        const float s = cnorm[b * K + k] - 2.f * wave_sum(dot[k]);
        if (s < best) { best = s; arg = k; }
      }
    }
    if (lane == 0) {
      nchanged += labels[n] != arg;
      labels[n] = arg;
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k == arg) {                                    // wave-uniform
        cnt[k] += 1.f;
#pragma unroll
        for (int j = 0; j < PMAX; ++j) { acc[k][j].x += xv[j].x; acc[k][j].y += xv[j].y; acc[k][j].z += xv[j].z; acc[k][j].w += xv[j].w; }
      }
    }
  }
  // merge the four waves (one after the other) in LDS, then one set of atomics per workgroup
  float *rcnt = red + K * C;
  int *rchg = reinterpret_cast<int *>(rcnt + KMAX);
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
#pragma unroll
          for (int j = 0; j < PMAX; ++j) {
            const int c = j * 256 + lane * 4;
            if (j < npiece && c < C) {
              float4 *dst = reinterpret_cast<float4 *>(red + k * C + c);
              if (w == 0) *dst = acc[k][j];
              else { float4 o = *dst; o.x += acc[k][j].x; o.y += acc[k][j].y; o.z += acc[k][j].z; o.w += acc[k][j].w; *dst = o; }
            }
          }
          if (lane == 0) rcnt[k] = (w == 0 ? 0.f : rcnt[k]) + cnt[k];
        }
      }
      if (lane == 0) *rchg = (w == 0 ? 0 : *rchg) + nchanged;
    }
  }
  __syncthreads();
  float *sb = sums + (int64_t)b * K * C;
  for (int i = threadIdx.x; i < K * C; i += 256) {
    const float v = red[i];
    if (v != 0.f) atomicAdd(sb + i, v);
  }
  if (threadIdx.x < K && rcnt[threadIdx.x] != 0.f) atomicAdd(counts + b * K + threadIdx.x, rcnt[threadIdx.x]);
  if (threadIdx.x == 0 && *rchg) atomicAdd(changed + b, *rchg);
}

// one workgroup per image
__global__ __launch_bounds__(256) void kmeans_update(float *__restrict__ centers, float *__restrict__ cnorm, float *__restrict__ sums,
                                                     float *__restrict__ counts, int32_t *__restrict__ changed,
                                                     const float *__restrict__ tol, int32_t *__restrict__ done,
                                                     int32_t *__restrict__ n_iter, int K, int C)
{
  __shared__ float part[256];
  __shared__ float norms[KMAX][4];
  const int b = blockIdx.x;
  if (done[b]) return;
  float shift = 0.f;
  float nk[KMAX] = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < K * C; i += 256) {
    const int k = i / C;
    const float cntk = counts[b * K + k];
    const int64_t o = (int64_t)b * K * C + i;
    const float old = centers[o];
    const float nw = cntk > 0.f ? sums[o] / cntk : old;
    centers[o] = nw;
    sums[o] = 0.f;
    shift += (nw - old) * (nw - old);
#pragma unroll
    for (int kk = 0; kk < KMAX; ++kk)
      if (kk == k) nk[kk] += nw * nw;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  shift = wave_sum(shift);
#pragma unroll
  for (int kk = 0; kk < KMAX; ++kk) nk[kk] = wave_sum(nk[kk]);
  if (lane == 0) {
    part[wave] = shift;
#pragma unroll
    for (int kk = 0; kk < KMAX; ++kk) norms[kk][wave] = nk[kk];
  }
  __syncthreads();
  if (threadIdx.x < K) {
    cnorm[b * K + threadIdx.x] = (norms[threadIdx.x][0] + norms[threadIdx.x][1]) + (norms[threadIdx.x][2] + norms[threadIdx.x][3]);
    counts[b * K + threadIdx.x] = 0.f;
  }
  if (threadIdx.x == 0) {
    const float total = (part[0] + part[1]) + (part[2] + part[3]);
    n_iter[b] += 1;
    if (changed[b] == 0 || total <= tol[b]) done[b] = 1;
    changed[b] = 0;
  }
}

}  // namespace

extern "C" int pd_kmeans_assign(const float *X, const int32_t *blocks, int n_blocks, const float *centers, const float *cnorm,
                                const int32_t *done, int32_t *labels, float *sums, float *counts, int32_t *changed, int C, int K,
                                void *stream_)
{
  if (n_blocks < 0 || C <= 0 || (C & 3) || C > 256 * PMAX || K <= 0 || K > KMAX)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_kmeans_assign: n_blocks=%d C=%d (<= 2048, %% 4) K=%d (<= 4)", n_blocks, C, K);
  if (n_blocks == 0) return PD_OK;
  if (!X || !blocks || !centers || !cnorm || !done || !labels || !sums || !counts || !changed)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_kmeans_assign: null pointer");
  const size_t lds = ((size_t)K * C + KMAX + 1) * sizeof(float);
  hipLaunchKernelGGL(kmeans_assign, dim3(n_blocks), dim3(256), lds, (hipStream_t)stream_, X, blocks, centers, cnorm, done, labels, sums,
                     counts, changed, C, K);
  return pd_check_launch("pd_kmeans_assign");
}

extern "C" int pd_kmeans_update(float *centers, float *cnorm, float *sums, float *counts, int32_t *changed, const float *tol,
                                int32_t *done, int32_t *n_iter, int B, int K, int C, void *stream_)
{
  if (B < 0 || C <= 0 || K <= 0 || K > KMAX) return pd_set_error(PD_ERR_INVALID_ARG, "pd_kmeans_update: B=%d C=%d K=%d", B, C, K);
  if (B == 0) return PD_OK;
  if (!centers || !cnorm || !sums || !counts || !changed || !tol || !done || !n_iter)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_kmeans_update: null pointer");
  hipLaunchKernelGGL(kmeans_update, dim3(B), dim3(256), 0, (hipStream_t)stream_, centers, cnorm, sums, counts, changed, tol, done, n_iter, K, C);
  return pd_check_launch("pd_kmeans_update");
}
