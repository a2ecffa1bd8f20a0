// Device K-means (Lloyd) iteration, batched over images (C-ABI in include/pd_kmeans.h).
// assign: a workgroup per slab of 64 points, two passes (labels, then per-centre sums): see kmeans_assign.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pd_common.h"
#include "pd_kmeans.h"
#include "pd_msda.h"

int g_pd_dbg_kmeans = 0;   // tools/ only: 1 skip the label pass, 2 skip the sums pass, 4 skip its atomics

namespace {

constexpr int KCAP = 8;       // centres per image: kernels are instantiated for KMAX = 4 (pixel grouping, K = 4) and 8 (part ranking, K = 8)
constexpr int PMAX = 8;       // 16-byte pieces per lane: C <= 2048

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// One workgroup owns a slab of <= 64 points of one image and makes TWO passes over it:
//   A  (a wavefront per point, lanes across channels, two points in flight): label = argmin_k |c_k|^2 - 2 x.c_k with the
//      centres read from LDS (staged once per workgroup) — the only pass that comes from HBM;
//   B  (a thread per 4 channels, all points of the slab, which are now L2-resident): the per-centre sums of the slab in
//      K float4 registers per thread (the label is workgroup-uniform per point: a scalar branch picks the accumulator),
//      flushed as one set of atomics per workgroup.
// The first version kept the K x C running sums in the assigning wave's registers (250 VGPRs, one point at a time per wave):
// two waves per SIMD and every point's HBM latency exposed — 187 us per Lloyd iteration at 4 x 5431 x 1536, ~5x the
// bandwidth bound.
template <int KMAX, bool PARTIAL>   // PARTIAL: the slab's sums / counts are STORED to sums[blockIdx] / counts[blockIdx] (no atomics)
__global__ __launch_bounds__(256) void kmeans_assign(const float *__restrict__ X, const int32_t *__restrict__ blocks,
                                                     const float *__restrict__ centers, const float *__restrict__ cnorm,
                                                     const int32_t *__restrict__ done, int32_t *__restrict__ labels,
                                                     float *__restrict__ sums, float *__restrict__ counts,
                                                     int32_t *__restrict__ changed, int C, int K, int ablate)
{
  extern __shared__ __attribute__((aligned(16))) float cs[];                        // [K][C] centres, then 64 slab labels + 1 changed
  const int b = blocks[blockIdx.x * 3], first = blocks[blockIdx.x * 3 + 1], npts = blocks[blockIdx.x * 3 + 2];
  if (done[b]) return;
  int *slab = reinterpret_cast<int *>(cs + (int64_t)K * C);
  int *rchg = slab + 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int npiece = (C + 255) / 256;
  const float *cb = centers + (int64_t)b * K * C;
  for (int i = threadIdx.x * 4; i < K * C; i += 1024) *reinterpret_cast<float4 *>(cs + i) = *reinterpret_cast<const float4 *>(cb + i);
  if (threadIdx.x == 0) *rchg = 0;
  float cn[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) cn[k] = k < K ? cnorm[b * K + k] : 0.f;
  __syncthreads();
  // ---- pass A: labels
  int nchanged = 0;
  for (int p0 = wave; p0 < ((ablate & 1) ? 0 : npts); p0 += 8) {
    float dot[2][KMAX];
    float4 xv[2][PMAX];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int p = p0 + 4 * u;
#pragma unroll
      for (int j = 0; j < PMAX; ++j) {
        const int c = j * 256 + lane * 4;
        xv[u][j] = (p < npts && j < npiece && c < C) ? *reinterpret_cast<const float4 *>(X + (int64_t)(first + p) * C + c)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int k = 0; k < KMAX; ++k) dot[u][k] = 0.f;
#pragma unroll
      for (int j = 0; j < PMAX; ++j) {
        const int c = j * 256 + lane * 4;
        if (j < npiece && c < C) {
#pragma unroll
          for (int k = 0; k < KMAX; ++k)
            if (k < K) {
              const float4 cv = *reinterpret_cast<const float4 *>(cs + k * C + c);
              dot[u][k] += xv[u][j].x * cv.x + xv[u][j].y * cv.y + xv[u][j].z * cv.z + xv[u][j].w * cv.w;
            }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int p = p0 + 4 * u;
      float best = INFINITY;
      int arg = 0;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) {
          const float sc = cn[k] - 2.f * wave_sum(dot[u][k]);
          if (sc < best) { best = sc; arg = k; }
        }
      if (lane == 0 && p < npts) {
        nchanged += labels[first + p] != arg;
        labels[first + p] = arg;
        slab[p] = arg;
      }
    }
  }
  if (lane == 0 && nchanged) atomicAdd(rchg, nchanged);
  __syncthreads();
  // ---- pass B: per-centre sums of the slab
  float *sb = sums + (PARTIAL ? (int64_t)blockIdx.x : (int64_t)b) * K * C;
  for (int c4 = threadIdx.x; c4 * 4 < ((ablate & 2) ? 0 : C); c4 += 256) {
    float4 acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *xc = X + (int64_t)first * C + c4 * 4;
#pragma unroll 4
    for (int p = 0; p < npts; ++p) {
      const float4 v = *reinterpret_cast<const float4 *>(xc + (int64_t)p * C);
      const int l = slab[p];                                           // uniform over the workgroup
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (l == k) { acc[k].x += v.x; acc[k].y += v.y; acc[k].z += v.z; acc[k].w += v.w; }
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K && PARTIAL) *reinterpret_cast<float4 *>(sb + (int64_t)k * C + c4 * 4) = acc[k];
      else if (k < K && !(ablate & 4)) {
        float *d = sb + (int64_t)k * C + c4 * 4;
        if (acc[k].x != 0.f) atomicAdd(d, acc[k].x);
        if (acc[k].y != 0.f) atomicAdd(d + 1, acc[k].y);
        if (acc[k].z != 0.f) atomicAdd(d + 2, acc[k].z);
        if (acc[k].w != 0.f) atomicAdd(d + 3, acc[k].w);
      }
  }
  if (threadIdx.x < K) {
    float n = 0.f;
    for (int p = 0; p < npts; ++p) n += slab[p] == (int)threadIdx.x ? 1.f : 0.f;
    if (PARTIAL) counts[(int64_t)blockIdx.x * K + threadIdx.x] = n;
    else if (n != 0.f) atomicAdd(counts + b * K + threadIdx.x, n);
  }
  if (threadIdx.x == 0 && *rchg) atomicAdd(changed + b, *rchg);
}

// sums[b] / counts[b] = sum over the image's slabs of the partial sums pd_kmeans_assign_partial stored (a thread per (k, c)
// element, slabs of an image are consecutive workgroups): the same-address atomics of ~170 workgroups per image cost 60 us
// per iteration, these 17 MB of plain reads a few
__global__ __launch_bounds__(64) void kmeans_reduce(const float *__restrict__ psums, const float *__restrict__ pcounts,
                                                     const int32_t *__restrict__ range, const int32_t *__restrict__ done,
                                                     float *__restrict__ sums, float *__restrict__ counts, int K, int C)
{
  const int b = blockIdx.y;
  if (done[b]) return;
  const int first = range[2 * b], n = range[2 * b + 1];
  const int e = blockIdx.x * blockDim.x + threadIdx.x, KC = K * C;
  if (e < KC) {
    const float *p = psums + (int64_t)first * KC + e;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};               // 8 independent chains: the loads are latency-bound
    int i = 0;
    for (; i + 8 <= n; i += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += p[(int64_t)(i + u) * KC];
    }
    for (; i < n; ++i) a[0] += p[(int64_t)i * KC];
    sums[(int64_t)b * KC + e] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  if (blockIdx.x == 0 && threadIdx.x < K) {
    float c = 0.f;
    for (int i = 0; i < n; ++i) c += pcounts[(int64_t)(first + i) * K + threadIdx.x];
    counts[b * K + threadIdx.x] = c;
  }
}

// one workgroup per image
template <int KMAX>
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
  float nk[KMAX];
#pragma unroll
  for (int kk = 0; kk < KMAX; ++kk) nk[kk] = 0.f;
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

// reduce + M-step + convergence in one launch (the product path): a thread per (k, c) element adds up the image's slab partials
// in a fixed order, writes the new centre and its contributions to the centre shift and to |c_k|^2; the workgroups of an image
// leave those as per-workgroup partials and the LAST one to finish (agent-scope ticket) sums them — again in a fixed order, so
// centres, norms and the stopping decision are bit-reproducible — and decides `done`.
template <int KMAX>
__global__ __launch_bounds__(256) void kmeans_reduce_update(const float *__restrict__ psums, const float *__restrict__ pcounts,
                                                            const int32_t *__restrict__ range, float *__restrict__ centers,
                                                            float *__restrict__ cnorm, int32_t *__restrict__ changed,
                                                            const float *__restrict__ tol, int32_t *__restrict__ done,
                                                            int32_t *__restrict__ n_iter, float *__restrict__ scratch,
                                                            int32_t *__restrict__ ticket, int K, int C)
{
  __shared__ float cnt_s[KMAX];
  __shared__ float red[4][KMAX + 1];
  __shared__ int last;
  const int b = blockIdx.y, G = gridDim.x;
  if (done[b]) return;
  const int first = range[2 * b], n = range[2 * b + 1], KC = K * C;
  if (threadIdx.x < K) {
    float c = 0.f;
    for (int i = 0; i < n; ++i) c += pcounts[(int64_t)(first + i) * K + threadIdx.x];
    cnt_s[threadIdx.x] = c;
  }
  __syncthreads();
  const int e = blockIdx.x * 256 + threadIdx.x;
  float shift = 0.f, nk[KMAX];
#pragma unroll
  for (int kk = 0; kk < KMAX; ++kk) nk[kk] = 0.f;
  if (e < KC) {
    const float *p = psums + (int64_t)first * KC + e;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int i = 0;
    for (; i + 8 <= n; i += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += p[(int64_t)(i + u) * KC];
    }
    for (; i < n; ++i) a[0] += p[(int64_t)i * KC];
    const float sm = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    const int k = e / C;
    const float old = centers[(int64_t)b * KC + e];
    const float nw = cnt_s[k] > 0.f ? sm / cnt_s[k] : old;
    centers[(int64_t)b * KC + e] = nw;
    shift = (nw - old) * (nw - old);
#pragma unroll
    for (int kk = 0; kk < KMAX; ++kk)
      if (kk == k) nk[kk] = nw * nw;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  shift = wave_sum(shift);
#pragma unroll
  for (int kk = 0; kk < KMAX; ++kk) nk[kk] = wave_sum(nk[kk]);
  if (lane == 0) {
    red[wave][KMAX] = shift;
#pragma unroll
    for (int kk = 0; kk < KMAX; ++kk) red[wave][kk] = nk[kk];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float *mine = scratch + ((int64_t)b * G + blockIdx.x) * (KMAX + 1);
#pragma unroll
    for (int j = 0; j <= KMAX; ++j) mine[j] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
    const int t = __hip_atomic_fetch_add(ticket + b, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);   // releases `mine`, acquires the others'
    last = t == G - 1;
    if (last) {
      float tot[KMAX + 1];
#pragma unroll
      for (int j = 0; j <= KMAX; ++j) tot[j] = 0.f;
      for (int g = 0; g < G; ++g) {
        const float *o = scratch + ((int64_t)b * G + g) * (KMAX + 1);
#pragma unroll
        for (int j = 0; j <= KMAX; ++j) tot[j] += __hip_atomic_load(o + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      for (int k = 0; k < K; ++k) cnorm[b * K + k] = tot[k];
      n_iter[b] += 1;
      if (changed[b] == 0 || tot[KMAX] <= tol[b]) done[b] = 1;
      changed[b] = 0;
      ticket[b] = 0;
    }
  }
}

}  // namespace

extern "C" int pd_kmeans_assign(const float *X, const int32_t *blocks, int n_blocks, const float *centers, const float *cnorm,
                                const int32_t *done, int32_t *labels, float *sums, float *counts, int32_t *changed, int C, int K,
                                void *stream_)
{
  if (n_blocks < 0 || C <= 0 || (C & 3) || C > 256 * PMAX || K <= 0 || K > KCAP)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_kmeans_assign: n_blocks=%d C=%d (<= 2048, %% 4) K=%d (<= 8)", n_blocks, C, K);
  if (n_blocks == 0) return PD_OK;
  if (!X || !blocks || !centers || !cnorm || !done || !labels || !sums || !counts || !changed)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_kmeans_assign: null pointer");
  const size_t lds = ((size_t)K * C + 64 + 1) * sizeof(float);
  if (K <= 4)
    hipLaunchKernelGGL((kmeans_assign<4, false>), dim3(n_blocks), dim3(256), lds, (hipStream_t)stream_, X, blocks, centers, cnorm, done, labels, sums,
                       counts, changed, C, K, g_pd_dbg_kmeans);
  else
    hipLaunchKernelGGL((kmeans_assign<8, false>), dim3(n_blocks), dim3(256), lds, (hipStream_t)stream_, X, blocks, centers, cnorm, done, labels, sums,
                       counts, changed, C, K, g_pd_dbg_kmeans);
  return pd_check_launch("pd_kmeans_assign");
}

extern "C" int pd_kmeans_assign_partial(const float *X, const int32_t *blocks, int n_blocks, const float *centers, const float *cnorm,
                                        const int32_t *done, int32_t *labels, float *partial_sums, float *partial_counts,
                                        int32_t *changed, int C, int K, void *stream_)
{
  if (n_blocks < 0 || C <= 0 || (C & 3) || C > 256 * PMAX || K <= 0 || K > KCAP)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_kmeans_assign_partial: n_blocks=%d C=%d (<= 2048, %% 4) K=%d (<= 8)", n_blocks, C, K);
  if (n_blocks == 0) return PD_OK;
  if (!X || !blocks || !centers || !cnorm || !done || !labels || !partial_sums || !partial_counts || !changed)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_kmeans_assign_partial: null pointer");
  const size_t lds = ((size_t)K * C + 64 + 1) * sizeof(float);
  if (K <= 4)
    hipLaunchKernelGGL((kmeans_assign<4, true>), dim3(n_blocks), dim3(256), lds, (hipStream_t)stream_, X, blocks, centers, cnorm, done, labels,
                       partial_sums, partial_counts, changed, C, K, g_pd_dbg_kmeans);
  else
    hipLaunchKernelGGL((kmeans_assign<8, true>), dim3(n_blocks), dim3(256), lds, (hipStream_t)stream_, X, blocks, centers, cnorm, done, labels,
                       partial_sums, partial_counts, changed, C, K, g_pd_dbg_kmeans);
  return pd_check_launch("pd_kmeans_assign_partial");
}

extern "C" int pd_kmeans_reduce(const float *partial_sums, const float *partial_counts, const int32_t *block_range, const int32_t *done,
                                float *sums, float *counts, int B, int K, int C, void *stream_)
{
  if (B < 0 || C <= 0 || K <= 0 || K > KCAP) return pd_set_error(PD_ERR_INVALID_ARG, "pd_kmeans_reduce: B=%d C=%d K=%d", B, C, K);
  if (B == 0) return PD_OK;
  if (!partial_sums || !partial_counts || !block_range || !done || !sums || !counts)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_kmeans_reduce: null pointer");
  hipLaunchKernelGGL(kmeans_reduce, dim3((K * C + 63) / 64, B), dim3(64), 0, (hipStream_t)stream_, partial_sums, partial_counts, block_range,
                     done, sums, counts, K, C);
  return pd_check_launch("pd_kmeans_reduce");
}

extern "C" int pd_kmeans_update(float *centers, float *cnorm, float *sums, float *counts, int32_t *changed, const float *tol,
                                int32_t *done, int32_t *n_iter, int B, int K, int C, void *stream_)
{
  if (B < 0 || C <= 0 || K <= 0 || K > KCAP) return pd_set_error(PD_ERR_INVALID_ARG, "pd_kmeans_update: B=%d C=%d K=%d", B, C, K);
  if (B == 0) return PD_OK;
  if (!centers || !cnorm || !sums || !counts || !changed || !tol || !done || !n_iter)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_kmeans_update: null pointer");
  if (K <= 4) hipLaunchKernelGGL(kmeans_update<4>, dim3(B), dim3(256), 0, (hipStream_t)stream_, centers, cnorm, sums, counts, changed, tol, done, n_iter, K, C);
  else hipLaunchKernelGGL(kmeans_update<8>, dim3(B), dim3(256), 0, (hipStream_t)stream_, centers, cnorm, sums, counts, changed, tol, done, n_iter, K, C);
  return pd_check_launch("pd_kmeans_update");
}

extern "C" int64_t pd_kmeans_reduce_update_scratch_floats(int B, int K, int C)
{
  if (B <= 0 || K <= 0 || C <= 0) return 0;
  return (int64_t)B * ((K * C + 255) / 256) * ((K <= 4 ? 4 : 8) + 1);
}

extern "C" int pd_kmeans_reduce_update(const float *partial_sums, const float *partial_counts, const int32_t *block_range, float *centers,
                                       float *cnorm, int32_t *changed, const float *tol, int32_t *done, int32_t *n_iter, float *scratch,
                                       int32_t *ticket, int B, int K, int C, void *stream_)
{
  if (B < 0 || C <= 0 || K <= 0 || K > KCAP) return pd_set_error(PD_ERR_INVALID_ARG, "pd_kmeans_reduce_update: B=%d C=%d K=%d", B, C, K);
  if (B == 0) return PD_OK;
  if (!partial_sums || !partial_counts || !block_range || !centers || !cnorm || !changed || !tol || !done || !n_iter || !scratch || !ticket)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_kmeans_reduce_update: null pointer");
  if (K <= 4)
    hipLaunchKernelGGL(kmeans_reduce_update<4>, dim3((K * C + 255) / 256, B), dim3(256), 0, (hipStream_t)stream_, partial_sums, partial_counts,
                       block_range, centers, cnorm, changed, tol, done, n_iter, scratch, ticket, K, C);
  else
    hipLaunchKernelGGL(kmeans_reduce_update<8>, dim3((K * C + 255) / 256, B), dim3(256), 0, (hipStream_t)stream_, partial_sums, partial_counts,
                       block_range, centers, cnorm, changed, tol, done, n_iter, scratch, ticket, K, C);
  return pd_check_launch("pd_kmeans_reduce_update");
}

