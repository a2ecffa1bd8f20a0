// Pixel grouping: label map from per-centroid score maps (C-ABI in include/pd_grouping.h).  One thread per output pixel:
// K <= 32 bilinear taps from a [K, h, w] score block that stays in L2 (256 KB at K = 4, 128 x 128), one byte written.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pd_common.h"
#include "pd_grouping.h"
#include "pd_msda.h"

namespace {

// torch upsample_bilinear2d (align_corners = false) source index
__device__ __forceinline__ void src_index(int dst, float scale, int in_size, int &i0, int &ip, float &l0, float &l1)
{
  float src = scale * (dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  ip = (i0 < in_size - 1) ? 1 : 0;
  l1 = src - i0;
  l0 = 1.f - l1;
}

__global__ __launch_bounds__(256) void scores_argmax_u8(const float *__restrict__ scores, const uint8_t *__restrict__ mask,
                                                        uint8_t *__restrict__ labels, int K, int h, int w, float sh, float sw,
                                                        int H, int W)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const int64_t o = (int64_t)y * W + x;
  if (!mask[o]) { labels[o] = 0; return; }
  int y0, yp, x0, xp; float hy0, hy1, wx0, wx1;
  src_index(y, sh, h, y0, yp, hy0, hy1);
  src_index(x, sw, w, x0, xp, wx0, wx1);
  const int a = y0 * w + x0, b = a + xp, c = (y0 + yp) * w + x0, d = c + xp;
  float best = -INFINITY;
  int arg = 0;
  for (int k = 0; k < K; ++k) {
    const float *s = scores + (int64_t)k * h * w;
    const float v = hy0 * (wx0 * s[a] + wx1 * s[b]) + hy1 * (wx0 * s[c] + wx1 * s[d]);
    if (v > best) { best = v; arg = k; }
  }
  labels[o] = (uint8_t)(arg + 1);
}

// Per-pixel assignment of K mask-logit maps given at low resolution (inference of the proposal / part models):
// v_k = bilinear(logits_k)(y, x) * object(y, x);  arg = argmax_k score_k * sigmoid(v_k) (first maximum),
// obj = max_k v_k > 0;  positive[k] += [v_k > 0].  One thread per pixel, K taps-of-4 from an L2 / MALL resident block.
__global__ __launch_bounds__(256) void mask_assign(const float *__restrict__ logits, const float *__restrict__ scores,
                                                   const uint8_t *__restrict__ object, int16_t *__restrict__ arg,
                                                   uint8_t *__restrict__ obj, int32_t *__restrict__ positive, int K, int h, int w,
                                                   float sh, float sw, int H, int W)
{
  extern __shared__ int cnt[];                               // K positive-pixel counters of the workgroup
  for (int k = threadIdx.x; k < K; k += 256) cnt[k] = 0;
  __syncthreads();
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const bool inside = x < W && y < H;
  const int64_t o = (int64_t)y * W + x;
  const float om = (inside && (!object || object[o])) ? 1.f : 0.f;
  int y0 = 0, yp = 0, x0 = 0, xp = 0; float hy0 = 0.f, hy1 = 0.f, wx0 = 0.f, wx1 = 0.f;
  if (inside) {
    src_index(y, sh, h, y0, yp, hy0, hy1);
    src_index(x, sw, w, x0, xp, wx0, wx1);
  }
  const int a = y0 * w + x0, b = a + xp, c = (y0 + yp) * w + x0, d = c + xp;
  float best = -INFINITY, vmax = -INFINITY;
  int besti = 0;
  for (int k = 0; k < K; ++k) {
    float v = 0.f;
    if (inside) {
      const float *s = logits + (int64_t)k * h * w;
      v = (hy0 * (wx0 * s[a] + wx1 * s[b]) + hy1 * (wx0 * s[c] + wx1 * s[d])) * om;
    }
    const bool pos = inside && v > 0.f;
    const unsigned long long bal = __ballot(pos);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&cnt[k], __popcll(bal));
    const float p = scores[k] * (1.f / (1.f + __expf(-v)));
    if (p > best) { best = p; besti = k; }
    vmax = fmaxf(vmax, v);
  }
  if (inside) {
    arg[o] = (int16_t)besti;
    obj[o] = vmax > 0.f ? 1 : 0;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 256)
    if (cnt[k]) atomicAdd(positive + k, cnt[k]);
}

}  // namespace

extern "C" int pd_mask_assign(const float *logits, const float *scores, const uint8_t *object, int16_t *arg, uint8_t *obj,
                              int32_t *positive, int K, int h, int w, int Hp, int Wp, int H, int W, void *stream_)
{
  if (K <= 0 || K > 8192 || h <= 0 || w <= 0 || Hp <= 0 || Wp <= 0 || H < 0 || W < 0 || H > Hp || W > Wp)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_mask_assign: K=%d h=%d w=%d Hp=%d Wp=%d H=%d W=%d", K, h, w, Hp, Wp, H, W);
  if (H == 0 || W == 0) return PD_OK;
  if (!logits || !scores || !arg || !obj || !positive) return pd_set_error(PD_ERR_INVALID_ARG, "pd_mask_assign: null pointer");
  hipLaunchKernelGGL(mask_assign, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), (size_t)K * sizeof(int), (hipStream_t)stream_, logits, scores,
                     object, arg, obj, positive, K, h, w, (float)h / (float)Hp, (float)w / (float)Wp, H, W);
  return pd_check_launch("pd_mask_assign");
}

extern "C" int pd_scores_argmax_u8(const float *scores, const uint8_t *mask, uint8_t *labels, int K, int h, int w, int Hp, int Wp,
                                   int H, int W, void *stream_)
{
  if (K <= 0 || K > 32 || h <= 0 || w <= 0 || Hp <= 0 || Wp <= 0 || H < 0 || W < 0 || H > Hp || W > Wp)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_scores_argmax_u8: K=%d h=%d w=%d Hp=%d Wp=%d H=%d W=%d", K, h, w, Hp, Wp, H, W);
  if (H == 0 || W == 0) return PD_OK;
  if (!scores || !mask || !labels) return pd_set_error(PD_ERR_INVALID_ARG, "pd_scores_argmax_u8: null pointer");
  hipLaunchKernelGGL(scores_argmax_u8, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, (hipStream_t)stream_, scores, mask, labels, K,
                     h, w, (float)h / (float)Hp, (float)w / (float)Wp, H, W);
  return pd_check_launch("pd_scores_argmax_u8");
}
