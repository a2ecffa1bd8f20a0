// Device input pipeline (include/pd_input.h): Pillow-exact 8-bit bilinear resample in two passes with flip / crops / pad
// folded into the addressing, and mask sampling straight from COCO run lengths.  All three kernels are byte streams
// (HBM-bound at a few MB per image); the point is to take ~10 ms of per-image CPU work off the dataloader.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pd_common.h"
#include "pd_input.h"
#include "pd_msda.h"

namespace {
constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ uint8_t clip8(int v)
{
  v >>= PRECISION_BITS;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// one thread per (row, output column): 3 channels, taps contiguous in the source row
__global__ __launch_bounds__(256) void resample_rows(const uint8_t *__restrict__ src, int W, int row0, int rows, int x0, int flip,
                                                     const int32_t *__restrict__ xmin, const int32_t *__restrict__ cnt,
                                                     const int32_t *__restrict__ kk, int ksize, int out_w, uint8_t *__restrict__ tmp)
{
  const int x = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (x >= out_w) return;
  const uint8_t *row = src + (int64_t)(row0 + r) * W * 3;
  const int32_t *k = kk + (int64_t)x * ksize;
  const int first = x0 + xmin[x], n = cnt[x];
  int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
  for (int j = 0; j < n; ++j) {
    const int col = flip ? W - 1 - (first + j) : first + j;
    const uint8_t *p = row + col * 3;
    const int w = k[j];
    a0 += p[0] * w; a1 += p[1] * w; a2 += p[2] * w;
  }
  uint8_t *o = tmp + ((int64_t)r * out_w + x) * 3;
  o[0] = clip8(a0); o[1] = clip8(a1); o[2] = clip8(a2);
}

// one thread per output pixel of the S x S canvas, all 3 channel planes
__global__ __launch_bounds__(256) void resample_cols(const uint8_t *__restrict__ tmp, int tmp_w, int r0, const int32_t *__restrict__ ymin,
                                                     const int32_t *__restrict__ cnt, const int32_t *__restrict__ kk, int ksize, int vh,
                                                     int vw, int S, int pad_value, uint8_t *__restrict__ out)
{
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= S) return;
  int a0, a1, a2;
  if (y < vh && x < vw) {
    const int32_t *k = kk + (int64_t)y * ksize;
    const int first = ymin[y] - r0, n = cnt[y];
    a0 = a1 = a2 = 1 << (PRECISION_BITS - 1);
    for (int j = 0; j < n; ++j) {
      const uint8_t *p = tmp + ((int64_t)(first + j) * tmp_w + x) * 3;
      const int w = k[j];
      a0 += p[0] * w; a1 += p[1] * w; a2 += p[2] * w;
    }
    a0 = clip8(a0); a1 = clip8(a1); a2 = clip8(a2);
  } else {
    a0 = a1 = a2 = pad_value;
  }
  const int64_t plane = (int64_t)S * S, o = (int64_t)y * S + x;
  out[o] = (uint8_t)a0; out[plane + o] = (uint8_t)a1; out[2 * plane + o] = (uint8_t)a2;
}

// one thread per (mask, output pixel): binary search of the pixel's column-major position in the mask's run starts
__global__ __launch_bounds__(256) void rle_sample(const int32_t *__restrict__ starts, const int32_t *__restrict__ offsets, int H, int W,
                                                  int flip, const int32_t *__restrict__ src_x, const int32_t *__restrict__ src_y, int vh,
                                                  int vw, int S, uint8_t *__restrict__ out, int32_t *__restrict__ area)
{
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, i = blockIdx.z;
  int v = 0;
  if (x < S && y < vh && x < vw) {
    const int sx = flip ? W - 1 - src_x[x] : src_x[x];
    const int pos = sx * H + src_y[y];
    const int32_t *st = starts + offsets[i];
    int lo = 0, hi = offsets[i + 1] - offsets[i];          // last run whose start <= pos
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (st[mid] <= pos) lo = mid; else hi = mid;
    }
    v = lo & 1;                                             // runs alternate 0, 1, 0, ... starting with zeros
  }
  if (x < S) out[((int64_t)i * S + y) * S + x] = (uint8_t)v;
  const unsigned long long ball = __ballot(v != 0);
  if ((threadIdx.x & 63) == 0 && ball) atomicAdd(area + i, __popcll(ball));
}
}  // namespace

extern "C" int pd_resample_rows_u8(const uint8_t *src, int H, int W, int row0, int rows, int x0, int flip, const int32_t *xmin,
                                   const int32_t *cnt, const int32_t *kk, int ksize, int out_w, uint8_t *tmp, void *stream_)
{
  if (H <= 0 || W <= 0 || rows < 0 || out_w < 0 || row0 < 0 || row0 + rows > H || ksize <= 0)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_resample_rows_u8: bad sizes H=%d W=%d row0=%d rows=%d out_w=%d", H, W, row0, rows, out_w);
  if (rows == 0 || out_w == 0) return PD_OK;
  if (!src || !xmin || !cnt || !kk || !tmp) return pd_set_error(PD_ERR_INVALID_ARG, "pd_resample_rows_u8: null pointer");
  hipLaunchKernelGGL(resample_rows, dim3((out_w + 255) / 256, rows), dim3(256), 0, (hipStream_t)stream_, src, W, row0, rows, x0, flip,
                     xmin, cnt, kk, ksize, out_w, tmp);
  return pd_check_launch("pd_resample_rows_u8");
}

extern "C" int pd_resample_cols_u8(const uint8_t *tmp, int tmp_rows, int tmp_w, int r0, const int32_t *ymin, const int32_t *cnt,
                                   const int32_t *kk, int ksize, int vh, int vw, int S, int pad_value, uint8_t *out, void *stream_)
{
  if (S <= 0 || vh < 0 || vw < 0 || vh > S || vw > S || vw > tmp_w || ksize <= 0)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_resample_cols_u8: bad sizes S=%d vh=%d vw=%d tmp_w=%d", S, vh, vw, tmp_w);
  if (!out || (vh > 0 && vw > 0 && (!tmp || !ymin || !cnt || !kk))) return pd_set_error(PD_ERR_INVALID_ARG, "pd_resample_cols_u8: null pointer");
  hipLaunchKernelGGL(resample_cols, dim3((S + 255) / 256, S), dim3(256), 0, (hipStream_t)stream_, tmp, tmp_w, r0, ymin, cnt, kk, ksize,
                     vh, vw, S, pad_value, out);
  return pd_check_launch("pd_resample_cols_u8");
}

extern "C" int pd_rle_sample_u8(const int32_t *starts, const int32_t *offsets, int n_masks, int H, int W, int flip, const int32_t *src_x,
                                const int32_t *src_y, int vh, int vw, int S, uint8_t *out, int32_t *area, void *stream_)
{
  if (n_masks < 0 || S <= 0 || H <= 0 || W <= 0 || vh < 0 || vw < 0 || vh > S || vw > S || (int64_t)H * W > 0x7fffffffLL)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_rle_sample_u8: bad sizes n=%d S=%d H=%d W=%d", n_masks, S, H, W);
  if (n_masks == 0) return PD_OK;
  if (!starts || !offsets || !out || !area || (vh > 0 && vw > 0 && (!src_x || !src_y)))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_rle_sample_u8: null pointer");
  hipLaunchKernelGGL(rle_sample, dim3((S + 255) / 256, S, n_masks), dim3(256), 0, (hipStream_t)stream_, starts, offsets, H, W, flip, src_x,
                     src_y, vh, vw, S, out, area);
  return pd_check_launch("pd_rle_sample_u8");
}

// (x - mean) / std of B same-size planar uint8 images [3, H, W] written straight into the channels-last fp32 batch the backbone reads
// (reference proposal_model.py / part_distillation_model.py: `(x - self.pixel_mean) / self.pixel_std` per image, then ImageList.from_tensors)
// — one launch for the batch (was a subtraction per image plus a division over the batch).  The same two fp32 operations per value.
namespace {
struct NormImages { const uint8_t *img[PD_NORMALIZE_MAX_IMAGES]; float mean[3], std[3]; };

__global__ __launch_bounds__(256) void normalize_u8_nhwc(NormImages in, float *__restrict__ out, int B, int64_t hw)
{
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)B * hw; i += (int64_t)gridDim.x * 256) {
    const int b = (int)(i / hw);
    const int64_t px = i - (int64_t)b * hw;
    const uint8_t *p = in.img[b];
    const float r = ((float)p[px] - in.mean[0]) / in.std[0];
    const float g = ((float)p[hw + px] - in.mean[1]) / in.std[1];
    const float bl = ((float)p[2 * hw + px] - in.mean[2]) / in.std[2];
    float *o = out + i * 3;
    o[0] = r; o[1] = g; o[2] = bl;
  }
}
}  // namespace

extern "C" int pd_normalize_u8_nhwc(const uint8_t *const *images, int B, int H, int W, const float *mean3, const float *std3, float *out, void *stream_)
{
  if (B < 0 || B > PD_NORMALIZE_MAX_IMAGES || H <= 0 || W <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_normalize_u8_nhwc: B=%d (<= %d) H=%d W=%d", B, PD_NORMALIZE_MAX_IMAGES, H, W);
  if (B == 0) return PD_OK;
  if (!images || !mean3 || !std3 || !out) return pd_set_error(PD_ERR_INVALID_ARG, "pd_normalize_u8_nhwc: null pointer");
  NormImages in;
  for (int b = 0; b < B; ++b) {
    if (!images[b]) return pd_set_error(PD_ERR_INVALID_ARG, "pd_normalize_u8_nhwc: null image %d", b);
    in.img[b] = images[b];
  }
  for (int c = 0; c < 3; ++c) { in.mean[c] = mean3[c]; in.std[c] = std3[c]; }
  const int64_t hw = (int64_t)H * W, total = (int64_t)B * hw;
  const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  hipLaunchKernelGGL(normalize_u8_nhwc, dim3(grid), dim3(256), 0, (hipStream_t)stream_, in, out, B, hw);
  return pd_check_launch("pd_normalize_u8_nhwc");
}
