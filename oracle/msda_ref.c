/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not shipped, never on the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 *
 * Plain-C restatement of multi-scale deformable attention, forward and
 * backward, following the arithmetic of the reference's intended native op:
 *   reference part_distillation/modeling/pixel_decoder/ops/src/cuda/ms_deform_im2col_cuda.cuh
 *     :38-89    4-corner zero-padded bilinear read        -> bilinear_read()
 *     :92-164   backward of the same                        -> bilinear_bwd()
 *     :242-304  forward kernel (index decomposition, h_im = y*H-0.5, in-range test)
 *     :306-408  backward kernel (per-(b,q,m) reduction over channels of
 *               grad_sampling_loc / grad_attn_weight; scatter-add grad_value)
 * which is the same function as the PyTorch fallback the reference actually
 * runs (functions/ms_deform_attn_func.py:55-75, F.grid_sample bilinear /
 * zeros / align_corners=False) — pinned against it in tests/test_oracle.py
 * with the reference's own fixture (ops/test.py:27-34, torch.manual_seed(3)).
 *
 * Layouts (all contiguous, row-major):
 *   value  [N,S,M,D]   shapes i64 [L,2]=(H,W)   level_start i64 [L]
 *   loc    [N,Lq,M,L,P,2] (x,y) in [0,1]        attn [N,Lq,M,L,P]
 *   out / grad_out [N,Lq,M*D]
 * Accumulation order is deterministic (serial loops), unlike the atomics of
 * the CUDA backward; sums are carried in the element type like the reference.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define DEFINE_MSDA(T, SUF)                                                                     \
  static T bilinear_read_##SUF(const T *v, int H, int W, int M, int D, T h, T w, int m, int c)  \
  {                                                                                             \
    int h_low = (int)floor((double)h), w_low = (int)floor((double)w);                           \
    int h_high = h_low + 1, w_high = w_low + 1;                                                 \
    T lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;                                 \
    int ws = M * D, hs = W * ws, base = m * D + c;                                              \
    T v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                                           \
    if (h_low >= 0 && w_low >= 0) v1 = v[h_low * hs + w_low * ws + base];                       \
    if (h_low >= 0 && w_high <= W - 1) v2 = v[h_low * hs + w_high * ws + base];                 \
    if (h_high <= H - 1 && w_low >= 0) v3 = v[h_high * hs + w_low * ws + base];                 \
    if (h_high <= H - 1 && w_high <= W - 1) v4 = v[h_high * hs + w_high * ws + base];           \
    T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;                                   \
    return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);                                             \
  }                                                                                             \
                                                                                                \
  void pd_oracle_msda_forward_##SUF(const T *value, const int64_t *shapes,                      \
                                    const int64_t *lvl_start, const T *loc, const T *attn,      \
                                    T *out, int N, int S, int M, int D, int L, int Lq, int P)   \
  {                                                                                             \
    for (int b = 0; b < N; ++b)                                                                 \
      for (int q = 0; q < Lq; ++q)                                                              \
        for (int m = 0; m < M; ++m) {                                                           \
          int64_t sidx = ((int64_t)b * Lq + q) * M + m;                                         \
          for (int c = 0; c < D; ++c) {                                                         \
            int64_t wp = sidx * L * P, lp = wp * 2;                                             \
            T col = 0;                                                                          \
            for (int l = 0; l < L; ++l) {                                                       \
              int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                           \
              const T *v = value + ((int64_t)b * S + lvl_start[l]) * M * D;                     \
              for (int p = 0; p < P; ++p, ++wp, lp += 2) {                                      \
                T x = loc[lp], y = loc[lp + 1], a = attn[wp];                                   \
                T h_im = y * H - (T)0.5, w_im = x * W - (T)0.5;                                 \
                if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)                             \
                  col += bilinear_read_##SUF(v, H, W, M, D, h_im, w_im, m, c) * a;              \
              }                                                                                 \
            }                                                                                   \
            out[sidx * D + c] = col;                                                            \
          }                                                                                     \
        }                                                                                       \
  }                                                                                             \
                                                                                                \
  void pd_oracle_msda_backward_##SUF(const T *value, const int64_t *shapes,                     \
                                     const int64_t *lvl_start, const T *loc, const T *attn,     \
                                     const T *grad_out, T *grad_value, T *grad_loc,             \
                                     T *grad_attn, int N, int S, int M, int D, int L, int Lq,   \
                                     int P)                                                     \
  {                                                                                             \
    memset(grad_value, 0, sizeof(T) * (size_t)N * S * M * D);                                   \
    memset(grad_loc, 0, sizeof(T) * (size_t)N * Lq * M * L * P * 2);                            \
    memset(grad_attn, 0, sizeof(T) * (size_t)N * Lq * M * L * P);                               \
    for (int b = 0; b < N; ++b)                                                                 \
      for (int q = 0; q < Lq; ++q)                                                              \
        for (int m = 0; m < M; ++m) {                                                           \
          int64_t sidx = ((int64_t)b * Lq + q) * M + m;                                         \
          for (int l = 0; l < L; ++l) {                                                         \
            int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                             \
            int64_t voff = ((int64_t)b * S + lvl_start[l]) * M * D;                             \
            const T *v = value + voff;                                                          \
            T *gv = grad_value + voff;                                                          \
            for (int p = 0; p < P; ++p) {                                                       \
              int64_t wp = (sidx * L + l) * P + p, lp = wp * 2;                                 \
              T x = loc[lp], y = loc[lp + 1], a = attn[wp];                                     \
              T h = y * H - (T)0.5, w = x * W - (T)0.5;                                         \
              if (!(h > -1 && w > -1 && h < H && w < W)) continue;                              \
              int h_low = (int)floor((double)h), w_low = (int)floor((double)w);                 \
              int h_high = h_low + 1, w_high = w_low + 1;                                       \
              T lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;                       \
              int ws = M * D, hs = W * ws;                                                      \
              T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;                         \
              T g_loc_x = 0, g_loc_y = 0, g_attn = 0;                                           \
              for (int c = 0; c < D; ++c) {                                                     \
                int base = m * D + c;                                                           \
                T top = grad_out[sidx * D + c];                                                 \
                T tgv = top * a;                                                                \
                T gh = 0, gw = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;                               \
                if (h_low >= 0 && w_low >= 0) {                                                 \
                  int i = h_low * hs + w_low * ws + base;                                       \
                  v1 = v[i]; gh -= hw * v1; gw -= hh * v1; gv[i] += w1 * tgv;                   \
                }                                                                               \
                if (h_low >= 0 && w_high <= W - 1) {                                            \
                  int i = h_low * hs + w_high * ws + base;                                      \
                  v2 = v[i]; gh -= lw * v2; gw += hh * v2; gv[i] += w2 * tgv;                   \
                }                                                                               \
                if (h_high <= H - 1 && w_low >= 0) {                                            \
                  int i = h_high * hs + w_low * ws + base;                                      \
                  v3 = v[i]; gh += hw * v3; gw -= lh * v3; gv[i] += w3 * tgv;                   \
                }                                                                               \
                if (h_high <= H - 1 && w_high <= W - 1) {                                       \
                  int i = h_high * hs + w_high * ws + base;                                     \
                  v4 = v[i]; gh += lw * v4; gw += lh * v4; gv[i] += w4 * tgv;                   \
                }                                                                               \
                T val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);                                \
                g_attn += top * val;                                                            \
                g_loc_x += W * gw * tgv;                                                        \
                g_loc_y += H * gh * tgv;                                                        \
              }                                                                                 \
              grad_attn[wp] = g_attn;                                                           \
              grad_loc[lp] = g_loc_x;                                                           \
              grad_loc[lp + 1] = g_loc_y;                                                       \
            }                                                                                   \
          }                                                                                     \
        }                                                                                       \
  }

DEFINE_MSDA(float, f32)
DEFINE_MSDA(double, f64)
