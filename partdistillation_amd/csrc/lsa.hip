// Batched rectangular linear-sum-assignment on the device (gfx950).
//
// Replaces the host round trip of reference modeling/matcher.py:159-163
// (C.cpu() -> scipy.optimize.linear_sum_assignment -> topk by cost).  One
// wavefront per problem: the problem is tiny (<= 64 targets x a few hundred
// queries) but there are B x 10 of them per step and each one costs the
// reference a device->host sync.  Algorithm = SciPy's (Crouse 2016,
// shortest augmenting paths with duals, float64), the column scan of each
// Dijkstra step spread over the 64 lanes and its arg-min taken with a wave
// reduction whose ordering key reproduces the serial scan's tie-breaking
// (prefer an unassigned column among equal distances; last such in scan order,
// else the first minimal one).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "pd_common.h"
#include "pd_criterion.h"
#include "pd_msda.h"

namespace {

constexpr int kMaxSmall = 64;     // smaller dimension (rows of the solved problem)
constexpr int kMaxLarge = 4096;   // larger dimension (columns of the solved problem)

struct Key {       // ordering of candidates inside one Dijkstra step
  double d;
  int unassigned;  // 1 if the column has no row yet
  int k;           // position in the todo list
};

__device__ __forceinline__ bool better(const Key &a, const Key &b)
{
  // true if a should replace b as the pick
  if (a.d != b.d) return a.d < b.d;
  if (a.unassigned != b.unassigned) return a.unassigned > b.unassigned;
  return a.unassigned ? (a.k > b.k) : (a.k < b.k);
}

__device__ __forceinline__ Key wave_best(Key x)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Key y;
    y.d = __shfl_xor(x.d, o, 64);
    y.unassigned = __shfl_xor(x.unassigned, o, 64);
    y.k = __shfl_xor(x.k, o, 64);
    if (better(y, x)) x = y;
  }
  return x;
}

// cost(r, c) of the SOLVED problem (R <= C); `tr` = the input was transposed
__device__ __forceinline__ double cost_at(const float *cost, int ld, bool tr, int r, int c)
{
  return tr ? (double)cost[(int64_t)c * ld + r] : (double)cost[(int64_t)r * ld + c];
}

__global__ __launch_bounds__(64) void lsa_kernel(const float *__restrict__ cost_all, const int32_t *__restrict__ ncols_all,
                                                  int64_t *__restrict__ out_rows, int64_t *__restrict__ out_cols,
                                                  int nrows, int ncols_max)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int nc_in = ncols_all[b];
  const float *cost = cost_all + (int64_t)b * nrows * ncols_max;
  int64_t *orow = out_rows + (int64_t)b * ncols_max, *ocol = out_cols + (int64_t)b * ncols_max;
  for (int k = lane; k < ncols_max; k += 64) { orow[k] = -1; ocol[k] = -1; }
  if (nc_in <= 0 || nrows <= 0) return;
  const bool tr = nc_in < nrows;            // SciPy transposes when there are more rows than columns
  const int R = tr ? nc_in : nrows, C = tr ? nrows : nc_in;
  // LDS carve (C-sized arrays first, 8-byte ones first)
  double *v = reinterpret_cast<double *>(smem);
  double *dist = v + C;
  double *u = dist + C;                      // R
  int *pred = reinterpret_cast<int *>(u + R);
  int *row4col = pred + C;
  int *todo = row4col + C;
  int *col4row = todo + C;                   // R
  unsigned char *row_seen = reinterpret_cast<unsigned char *>(col4row + R);   // R
  unsigned char *col_seen = row_seen + R;    // C
  __shared__ int s_fail;
  for (int j = lane; j < C; j += 64) { v[j] = 0.0; row4col[j] = -1; }
  for (int i = lane; i < R; i += 64) { u[i] = 0.0; col4row[i] = -1; }
  if (lane == 0) s_fail = 0;
  __syncthreads();
  // infeasible inputs (NaN / -inf) -> leave -1 (the reference would raise in SciPy)
  for (int idx = lane; idx < R * C; idx += 64) {
    const double c = cost_at(cost, ncols_max, tr, idx / C, idx % C);
    if (isnan(c) || c == -INFINITY) s_fail = 1;
  }
  __syncthreads();
  if (s_fail) return;

  for (int cur = 0; cur < R; ++cur) {
    for (int k = lane; k < C; k += 64) { todo[k] = C - k - 1; dist[k] = INFINITY; col_seen[k] = 0; }
    for (int k = lane; k < R; k += 64) row_seen[k] = 0;
    __syncthreads();
    double low = 0.0;
    int n_todo = C, i = cur, sink = -1;
    while (sink < 0) {
      if (lane == 0) row_seen[i] = 1;
      const double ui = u[i];
      Key best{INFINITY, 0, 0x7fffffff};
      for (int k = lane; k < n_todo; k += 64) {
        const int j = todo[k];
        const double r = low + cost_at(cost, ncols_max, tr, i, j) - ui - v[j];
        if (r < dist[j]) { dist[j] = r; pred[j] = i; }
        Key cand{dist[j], row4col[j] < 0 ? 1 : 0, k};
        // serial rule: replace if strictly smaller, or equal and unassigned
        if (cand.d < best.d || (cand.d == best.d && cand.unassigned)) best = cand;
      }
      best = wave_best(best);
      low = best.d;
      if (!(low < INFINITY)) { if (lane == 0) s_fail = 1; break; }
      const int pick = best.k;
      const int j = todo[pick];
      __syncthreads();
      if (row4col[j] < 0) sink = j; else i = row4col[j];
      if (lane == 0) { col_seen[j] = 1; todo[pick] = todo[n_todo - 1]; }
      --n_todo;
      __syncthreads();
    }
    __syncthreads();
    if (s_fail) return;
    // dual update
    if (lane == 0) u[cur] += low;
    for (int r = lane; r < R; r += 64)
      if (row_seen[r] && r != cur) u[r] += low - dist[col4row[r]];
    for (int j = lane; j < C; j += 64)
      if (col_seen[j]) v[j] -= low - dist[j];
    __syncthreads();
    // augment along the path (serial, short)
    if (lane == 0) {
      for (int j = sink;;) {
        const int r = pred[j];
        row4col[j] = r;
        const int prev = col4row[r];
        col4row[r] = j;
        j = prev;
        if (r == cur) break;
      }
    }
    __syncthreads();
  }
  // pairs in the ORIGINAL orientation: (row = query, col = target), then order by fp32 cost ascending
  if (lane == 0) {
    const int np = R;
    for (int k = 0; k < np; ++k) {
      const int64_t qi = tr ? col4row[k] : k, tj = tr ? k : col4row[k];
      const float c = cost[qi * ncols_max + tj];
      int pos = k;                                   // insertion sort by cost (stable)
      while (pos > 0 && cost[orow[pos - 1] * ncols_max + ocol[pos - 1]] > c) {
        orow[pos] = orow[pos - 1]; ocol[pos] = ocol[pos - 1]; --pos;
      }
      orow[pos] = qi; ocol[pos] = tj;
    }
  }
}

}  // namespace

extern "C" int pd_lsa_batched(const float *cost, const int32_t *ncols, int64_t *out_rows, int64_t *out_cols, int nbatch,
                              int nrows, int ncols_max, void *stream_)
{
  if (nbatch < 0 || nrows < 0 || ncols_max < 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_lsa_batched: negative size");
  if (nbatch == 0 || ncols_max == 0) return PD_OK;
  if (!cost || !ncols || !out_rows || !out_cols) return pd_set_error(PD_ERR_INVALID_ARG, "pd_lsa_batched: null pointer");
  const int small = nrows < ncols_max ? nrows : ncols_max, large = nrows < ncols_max ? ncols_max : nrows;
  if (small > kMaxSmall || large > kMaxLarge)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_lsa_batched: problem %dx%d exceeds %dx%d", nrows, ncols_max, kMaxLarge, kMaxSmall);
  const size_t C = (size_t)large, R = (size_t)(nrows > ncols_max ? nrows : ncols_max);   // upper bounds for both orientations
  (void)R;
  const size_t lds = (2 * C + C) * sizeof(double) + (3 * C + C) * sizeof(int) + 2 * C + 64;   // R <= C always
  hipLaunchKernelGGL(lsa_kernel, dim3(nbatch), dim3(64), lds, (hipStream_t)stream_, cost, ncols, out_rows, out_cols, nrows,
                     ncols_max);
  return pd_check_launch("pd_lsa_batched");
}
