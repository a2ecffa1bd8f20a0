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

// The same E-step with DISTANCE BOUNDS (Hamerly 2010), exact: per point an upper bound ub on the distance to its centre and a lower bound lb on the
// distance to every other centre, both moved by how far the centres moved (cshift, from kmeans_reduce_update).  While lb^2 - ub^2 exceeds what
// fp32 rounding of the scores |c_k|^2 - 2 x.c_k could hide (margin below: >= 5 x the worst-case bound C * 2^-24 * 2 (|x|^2 + |c|^2) at C = 2048),
// the fp32 argmin IS the current label and the point's features are not read at all; every other point is recomputed with the arithmetic of
// kmeans_assign, so labels, partial sums, centres and iteration counts are bit-identical to the unbounded kernel.  A slab whose labels did not
// change keeps the partial sums it stored in an earlier iteration (pass B skipped).  After the first few iterations of a run a few per cent of
// the points are recomputed; an iteration is then the launch, 12 bytes per point and the reduce / update kernel.
template <int KMAX>
__global__ __launch_bounds__(256) void kmeans_assign_bounded(const float *__restrict__ X, const int32_t *__restrict__ blocks,
                                                             const float *__restrict__ centers, const float *__restrict__ cnorm,
                                                             const int32_t *__restrict__ done, int32_t *__restrict__ labels,
                                                             float *__restrict__ sums, float *__restrict__ counts, int32_t *__restrict__ changed,
                                                             float *__restrict__ ub, float *__restrict__ lb, float *__restrict__ xnorm,
                                                             const float *__restrict__ cshift, int C, int K)
{
  extern __shared__ __attribute__((aligned(16))) float cs[];                        // [K][C] centres, 64 slab labels, 64 work items, 2 counters
  const int b = blocks[blockIdx.x * 3], first = blocks[blockIdx.x * 3 + 1], npts = blocks[blockIdx.x * 3 + 2];
  if (done[b]) return;
  int *slab = reinterpret_cast<int *>(cs + (int64_t)K * C);
  int *work = slab + 64, *rchg = work + 64, *nwork = rchg + 1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int npiece = (C + 255) / 256;
  float cn[KMAX], cmax2 = 0.f;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    cn[k] = k < K ? cnorm[b * K + k] : 0.f;
    cmax2 = fmaxf(cmax2, cn[k]);
  }
  if (threadIdx.x == 0) *rchg = 0, *nwork = 0;
  __syncthreads();
  // ---- bounds: which points of the slab have to be looked at
  if (threadIdx.x < npts) {
    const int p = threadIdx.x, a = labels[first + p];
    bool skip = false;
    if (a >= 0) {
      const float u = (ub[first + p] + cshift[(b * 2) * KCAP + a]) * 1.000001f;
      const float l = (lb[first + p] - cshift[(b * 2 + 1) * KCAP + a]) * 0.999999f;
      skip = l > u && l * l - u * u > 2e-3f * (xnorm[first + p] + cmax2);
      if (skip) ub[first + p] = u, lb[first + p] = l;
    }
    slab[p] = a;
    if (!skip) work[atomicAdd(nwork, 1)] = p;
  }
  __syncthreads();
  const int nw = *nwork;
  if (nw == 0) return;                                                             // nothing to recompute: labels, partial sums and counts stand
  const float *cb = centers + (int64_t)b * K * C;
  for (int i = threadIdx.x * 4; i < K * C; i += 1024) *reinterpret_cast<float4 *>(cs + i) = *reinterpret_cast<const float4 *>(cb + i);
  __syncthreads();
  // ---- pass A over the work list (kmeans_assign's arithmetic; the order of the list does not enter any result)
  int nchanged = 0;
  for (int i0 = wave; i0 < nw; i0 += 8) {
    float dot[2][KMAX], xn[2];
    float4 xv[2][PMAX];
    int pp[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = i0 + 4 * u;
      pp[u] = i < nw ? work[i] : -1;
#pragma unroll
      for (int j = 0; j < PMAX; ++j) {
        const int c = j * 256 + lane * 4;
        xv[u][j] = (pp[u] >= 0 && j < npiece && c < C) ? *reinterpret_cast<const float4 *>(X + (int64_t)(first + pp[u]) * C + c)
                                                       : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      xn[u] = 0.f;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) dot[u][k] = 0.f;
#pragma unroll
      for (int j = 0; j < PMAX; ++j) {
        const int c = j * 256 + lane * 4;
        if (j < npiece && c < C) {
          xn[u] += xv[u][j].x * xv[u][j].x + xv[u][j].y * xv[u][j].y + xv[u][j].z * xv[u][j].z + xv[u][j].w * xv[u][j].w;
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
      const int p = pp[u];
      float best = INFINITY, sc[KMAX];
      int arg = 0;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) {
          sc[k] = cn[k] - 2.f * wave_sum(dot[u][k]);
          if (sc[k] < best) { best = sc[k]; arg = k; }
        }
      const float x2 = wave_sum(xn[u]);
      if (lane == 0 && p >= 0) {
        float second = INFINITY;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
          if (k < K && k != arg) second = fminf(second, sc[k]);
        nchanged += slab[p] != arg;
        labels[first + p] = arg;
        slab[p] = arg;
        // distances from the scores: d_k^2 = |x|^2 + score_k; the bounds are rounded outwards (their fp32 error is inside the margin above)
        ub[first + p] = sqrtf(fmaxf(x2 + best, 0.f)) * 1.00001f;
        lb[first + p] = second == INFINITY ? INFINITY : sqrtf(fmaxf(x2 + second, 0.f)) * 0.99999f;
        xnorm[first + p] = x2;
      }
    }
  }
  if (lane == 0 && nchanged) atomicAdd(rchg, nchanged);
  __syncthreads();
  if (*rchg == 0) return;                                                          // every recomputed point kept its label: the stored sums stand
  // ---- pass B: per-centre sums of the slab (as kmeans_assign<.., true>)
  float *sb = sums + (int64_t)blockIdx.x * K * C;
  for (int c4 = threadIdx.x; c4 * 4 < C; c4 += 256) {
    float4 acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *xc = X + (int64_t)first * C + c4 * 4;
#pragma unroll 4
    for (int p = 0; p < npts; ++p) {
      const float4 v = *reinterpret_cast<const float4 *>(xc + (int64_t)p * C);
      const int l = slab[p];
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (l == k) { acc[k].x += v.x; acc[k].y += v.y; acc[k].z += v.z; acc[k].w += v.w; }
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) *reinterpret_cast<float4 *>(sb + (int64_t)k * C + c4 * 4) = acc[k];
  }
  if (threadIdx.x < K) {
    float n = 0.f;
    for (int p = 0; p < npts; ++p) n += slab[p] == (int)threadIdx.x ? 1.f : 0.f;
    counts[(int64_t)blockIdx.x * K + threadIdx.x] = n;
  }
  if (threadIdx.x == 0) atomicAdd(changed + b, *rchg);
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

// reduce + M-step + convergence in one launch (the product path): 64 (k, c) elements per workgroup; wavefront q adds up the image's slab
// partials i = q (mod 4), eight independent chains each (the loads are latency-bound: one thread per element walked ~170 dependent rounds of
// 8 loads in 29 us per iteration — four wavefronts per element and 4 x the workgroups: 6 rounds), the four sums are added in wavefront order,
// wavefront 0 writes the new centre and the element's contributions to the centre shift and to |c_k|^2; the workgroups of an image leave those
// as per-workgroup partials and the LAST one to finish (agent-scope ticket) sums them — every sum in a fixed order, so centres, norms and
// the stopping decision are bit-reproducible — and decides `done`.
constexpr int RU_ELEMS = 64;
template <int KMAX>
__global__ __launch_bounds__(256) void kmeans_reduce_update(const float *__restrict__ psums, const float *__restrict__ pcounts,
                                                            const int32_t *__restrict__ range, float *__restrict__ centers,
                                                            float *__restrict__ cnorm, int32_t *__restrict__ changed,
                                                            const float *__restrict__ tol, int32_t *__restrict__ done,
                                                            int32_t *__restrict__ n_iter, float *__restrict__ scratch,
                                                            int32_t *__restrict__ ticket, float *__restrict__ cshift, int K, int C)
{
  constexpr int NP = 2 * KMAX + 1;                  // per-workgroup partials: |c_k|^2 (KMAX), total shift^2, shift^2 of centre k (KMAX)
  __shared__ float cnt_w[4][KMAX];
  __shared__ float part[4][RU_ELEMS];
  const int b = blockIdx.y, G = gridDim.x;
  if (done[b]) return;
  const int first = range[2 * b], n = range[2 * b + 1], KC = K * C;
  const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
  {
    // cluster sizes: whole numbers in fp32 (exact in any order), a slab per thread
    float cp[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) cp[k] = 0.f;
    for (int i = threadIdx.x; i < n; i += 256)
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) cp[k] += pcounts[(int64_t)(first + i) * K + k];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      cp[k] = wave_sum(cp[k]);
      if (lane == 0) cnt_w[q][k] = cp[k];
    }
  }
  const int e = blockIdx.x * RU_ELEMS + lane;
  {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (e < KC) {
      const float *p = psums + (int64_t)first * KC + e;
      int i = q;
      for (; i + 28 < n; i += 32) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += p[(int64_t)(i + 4 * u) * KC];
      }
      for (; i < n; i += 4) a[0] += p[(int64_t)i * KC];
    }
    part[q][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  __syncthreads();
  if (q != 0) return;                                // (no barrier below)
  float shift = 0.f, nk[KMAX], sk[KMAX];
#pragma unroll
  for (int kk = 0; kk < KMAX; ++kk) nk[kk] = 0.f, sk[kk] = 0.f;
  if (e < KC) {
    const float sm = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    const int k = e / C;
    const float old = centers[(int64_t)b * KC + e];
    const float cnt = (cnt_w[0][k] + cnt_w[1][k]) + (cnt_w[2][k] + cnt_w[3][k]);
    const float nw = cnt > 0.f ? sm / cnt : old;
    centers[(int64_t)b * KC + e] = nw;
    shift = (nw - old) * (nw - old);
#pragma unroll
    for (int kk = 0; kk < KMAX; ++kk)
      if (kk == k) nk[kk] = nw * nw, sk[kk] = shift;
  }
  shift = wave_sum(shift);
#pragma unroll
  for (int kk = 0; kk < KMAX; ++kk) nk[kk] = wave_sum(nk[kk]), sk[kk] = wave_sum(sk[kk]);
  int is_last = 0;
  if (lane == 0) {
    float *mine = scratch + ((int64_t)b * G + blockIdx.x) * NP;
    mine[KMAX] = shift;
#pragma unroll
    for (int kk = 0; kk < KMAX; ++kk) mine[kk] = nk[kk], mine[KMAX + 1 + kk] = sk[kk];
    const int t = __hip_atomic_fetch_add(ticket + b, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);   // releases `mine`, acquires the others'
    is_last = t == G - 1;
  }
  is_last = __shfl(is_last, 0, 64);
  if (!is_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // all lanes read the other workgroups' partials below
  // the last workgroup's first wavefront: lane l adds the partials of workgroups l, l + 64, ..; then the lanes in butterfly order
  float tot[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) tot[j] = 0.f;
  for (int g = lane; g < G; g += 64) {
    const float *o = scratch + ((int64_t)b * G + g) * NP;
#pragma unroll
    for (int j = 0; j < NP; ++j) tot[j] += __hip_atomic_load(o + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) tot[j] = wave_sum(tot[j]);
  if (lane == 0) {
    for (int k = 0; k < K; ++k) cnorm[b * K + k] = tot[k];
    if (cshift) {
      // how far every centre moved (rounded UP: these feed distance bounds), and for centre k the largest move of any OTHER centre
      float sh[KMAX];
#pragma unroll
      for (int k = 0; k < KMAX; ++k) sh[k] = k < K ? sqrtf(tot[KMAX + 1 + k]) * 1.0001f : 0.f;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        float mo = 0.f;
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
          if (j != k) mo = fmaxf(mo, sh[j]);
        if (k < K) cshift[(b * 2) * KCAP + k] = sh[k], cshift[(b * 2 + 1) * KCAP + k] = mo;
      }
    }
    n_iter[b] += 1;
    if (changed[b] == 0 || tot[KMAX] <= tol[b]) done[b] = 1;
    changed[b] = 0;
    ticket[b] = 0;
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
  return (int64_t)B * ((K * C + RU_ELEMS - 1) / RU_ELEMS) * (2 * (K <= 4 ? 4 : 8) + 1);
}

extern "C" int pd_kmeans_reduce_update(const float *partial_sums, const float *partial_counts, const int32_t *block_range, float *centers,
                                       float *cnorm, int32_t *changed, const float *tol, int32_t *done, int32_t *n_iter, float *scratch,
                                       int32_t *ticket, int B, int K, int C, void *stream_)
{
  return pd_kmeans_reduce_update_shift(partial_sums, partial_counts, block_range, centers, cnorm, changed, tol, done, n_iter, scratch, ticket, nullptr, B, K,
                                       C, stream_);
}

extern "C" int pd_kmeans_assign_bounded(const float *X, const int32_t *blocks, int n_blocks, const float *centers, const float *cnorm,
                                        const int32_t *done, int32_t *labels, float *partial_sums, float *partial_counts, int32_t *changed,
                                        float *ub, float *lb, float *xnorm, const float *cshift, int C, int K, void *stream_)
{
  if (n_blocks < 0 || C <= 0 || (C & 3) || C > 256 * PMAX || K <= 0 || K > KCAP)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_kmeans_assign_bounded: n_blocks=%d C=%d (<= 2048, %% 4) K=%d (<= 8)", n_blocks, C, K);
  if (n_blocks == 0) return PD_OK;
  if (!X || !blocks || !centers || !cnorm || !done || !labels || !partial_sums || !partial_counts || !changed || !ub || !lb || !xnorm || !cshift)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_kmeans_assign_bounded: null pointer");
  const size_t lds = ((size_t)K * C + 64 + 64 + 2) * sizeof(float);
  if (K <= 4)
    hipLaunchKernelGGL((kmeans_assign_bounded<4>), dim3(n_blocks), dim3(256), lds, (hipStream_t)stream_, X, blocks, centers, cnorm, done, labels,
                       partial_sums, partial_counts, changed, ub, lb, xnorm, cshift, C, K);
  else
    hipLaunchKernelGGL((kmeans_assign_bounded<8>), dim3(n_blocks), dim3(256), lds, (hipStream_t)stream_, X, blocks, centers, cnorm, done, labels,
                       partial_sums, partial_counts, changed, ub, lb, xnorm, cshift, C, K);
  return pd_check_launch("pd_kmeans_assign_bounded");
}

extern "C" int pd_kmeans_reduce_update_shift(const float *partial_sums, const float *partial_counts, const int32_t *block_range, float *centers,
                                             float *cnorm, int32_t *changed, const float *tol, int32_t *done, int32_t *n_iter, float *scratch,
                                             int32_t *ticket, float *cshift, int B, int K, int C, void *stream_)
{
  if (B < 0 || C <= 0 || K <= 0 || K > KCAP) return pd_set_error(PD_ERR_INVALID_ARG, "pd_kmeans_reduce_update: B=%d C=%d K=%d", B, C, K);
  if (B == 0) return PD_OK;
  if (!partial_sums || !partial_counts || !block_range || !centers || !cnorm || !changed || !tol || !done || !n_iter || !scratch || !ticket)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_kmeans_reduce_update: null pointer");
  if (K <= 4)
    hipLaunchKernelGGL(kmeans_reduce_update<4>, dim3((K * C + RU_ELEMS - 1) / RU_ELEMS, B), dim3(256), 0, (hipStream_t)stream_, partial_sums, partial_counts,
                       block_range, centers, cnorm, changed, tol, done, n_iter, scratch, ticket, cshift, K, C);
  else
    hipLaunchKernelGGL(kmeans_reduce_update<8>, dim3((K * C + RU_ELEMS - 1) / RU_ELEMS, B), dim3(256), 0, (hipStream_t)stream_, partial_sums, partial_counts,
                       block_range, centers, cnorm, changed, tol, done, n_iter, scratch, ticket, cshift, K, C);
  return pd_check_launch("pd_kmeans_reduce_update");
}

