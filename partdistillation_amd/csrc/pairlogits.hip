// Mask logits of the MATCHED queries and their two gradients (C-ABI: include/pd_criterion.h, pd_pair_logits_*).
//
// The reference forms every query's mask, outputs_mask = einsum("bqc,bchw->bqhw", mask_embed, mask_features)
// (mask2former_transformer_decoder.py:441-459), and the criterion then keeps the matched ones, src_masks = pred_masks[src_idx]
// (criterion.py:147-160).  Here the decoder hands over mask_embed and mask_features, and only the N matched (query, target) pairs of
// all heads are multiplied out:  out[row(p), t] = sum_c e[p, c] * tok[b(p), t, c],  tok = the channels-last mask features of image b
// as [T = h w, C] tokens, e = the pairs' embeddings grouped by image.  Rounds 2-4 ran this as one library GEMM per image plus transposes,
// a concatenation and a row gather (forward), and two library GEMMs, two split-K launches and two copies (backward).
//
// All three products are fp32 x fp32 -> fp32 on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation: the criterion runs
// outside autocast).  Operand reads go straight from global memory into the MFMA layout — the contraction index of an MFMA step is
// only a summation index, so a lane takes FOUR consecutive elements of its row (one 16-byte load) as the k = lane / 16 slices of four
// steps; no operand is transposed through LDS.  Only the pairs' embeddings (<= 48 x 256 floats per pass) are staged in LDS.
//   pair_logits_fwd       M = pairs (3 tiles of 16), N = 64 tokens per wavefront, K = channels.      reads tok once (134 MB at config 2)
//   pair_logits_bwd_tok   d_tok[b, t, :] = sum_p g[row(p), t] e[p, :]:  M = channels, N = 16 tokens per wavefront, K = pairs; the channel
//                         order inside an M tile is chosen so that a lane ends up with 8 consecutive channels (full 128-byte lines)
//   pair_logits_bwd_rows  d_e[p, :] = sum_t g[row(p), t] tok[b, t, :]:  M = pairs, N = 64 channels per wavefront, K = a slab of tokens per
//                         workgroup; slab partials are stored and summed in slab order by pair_logits_rows_reduce (deterministic)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pd_common.h"
#include "pd_criterion.h"
#include "pd_msda.h"

namespace {
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int PL_PC = 48;                          // pairs per pass: three 16-row MFMA tiles
constexpr int PL_C = PD_PAIR_LOGITS_CHANNELS;      // 256
constexpr int PL_LD = PL_C + 4;                    // LDS row pitch of the staged embeddings (floats)
struct PairImgs {
  int start[PD_PAIR_LOGITS_MAX_IMAGES + 1];
};

__device__ __forceinline__ v4f mfma4(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// rows [pc, pc + cnt) of image-major e -> LDS [PL_PC][PL_LD], rows >= cnt zero; rows_lds[r] = output / gradient row of pair r
__device__ __forceinline__ void stage_pairs(float *lds, int64_t *rows_lds, const float *e, const int64_t *out_row, int first, int cnt)
{
  for (int i = threadIdx.x; i < PL_PC * (PL_C / 4); i += blockDim.x) {
    const int r = i / (PL_C / 4), c4 = i - r * (PL_C / 4);
    const float4 v = r < cnt ? *reinterpret_cast<const float4 *>(e + (int64_t)(first + r) * PL_C + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4 *>(lds + r * PL_LD + 4 * c4) = v;
  }
  if (threadIdx.x < PL_PC) {
    const int r = threadIdx.x;
    rows_lds[r] = r < cnt ? (out_row ? out_row[first + r] : (int64_t)(first + r)) : -1;
  }
}

__global__ __launch_bounds__(256) void pair_logits_fwd(const float *__restrict__ tok, const float *__restrict__ e, PairImgs imgs,
                                                       const int64_t *__restrict__ out_row, float *__restrict__ out, int T)
{
  __shared__ __attribute__((aligned(16))) float lds[PL_PC * PL_LD];
  __shared__ int64_t rows_lds[PL_PC];
  const int b = blockIdx.y, p0 = imgs.start[b], np = imgs.start[b + 1] - p0;
  if (np <= 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, kq = lane >> 4;
  const int t0 = blockIdx.x * 256 + wave * 64 + 4 * n;              // this lane's four tokens: t0 .. t0 + 3 (N tiles tt = 0 .. 3)
  const float *rowp[4];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) rowp[tt] = tok + ((int64_t)b * T + min(t0 + tt, T - 1)) * PL_C + 4 * kq;
  for (int pc = 0; pc < np; pc += PL_PC) {
    const int cnt = min(PL_PC, np - pc);
    __syncthreads();
    stage_pairs(lds, rows_lds, e, out_row, p0 + pc, cnt);
    __syncthreads();
    const int mts = (cnt + 15) >> 4;                                // M tiles that hold pairs
    v4f acc[3][4];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) acc[mt][tt] = v4f{0.f, 0.f, 0.f, 0.f};
    float4 bq[4], bn[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) bq[tt] = *reinterpret_cast<const float4 *>(rowp[tt]);
#pragma unroll 2
    for (int j = 0; j < PL_C / 16; ++j) {
      if (j + 1 < PL_C / 16) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) bn[tt] = *reinterpret_cast<const float4 *>(rowp[tt] + 16 * (j + 1));
      }
      float4 aq[3];
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) aq[mt] = *reinterpret_cast<const float4 *>(lds + (16 * mt + n) * PL_LD + 16 * j + 4 * kq);
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) {
        if (mt < mts) {
          const float a[4] = {aq[mt].x, aq[mt].y, aq[mt].z, aq[mt].w};
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) {
            const float bv[4] = {bq[tt].x, bq[tt].y, bq[tt].z, bq[tt].w};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[mt][tt] = mfma4(a[s], bv[s], acc[mt][tt]);
          }
        }
      }
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) bq[tt] = bn[tt];
    }
    // D[i = 4 kq + r][j = n]: pair 16 mt + 4 kq + r, tokens t0 + tt
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int pr = 16 * mt + 4 * kq + r;
        if (pr >= cnt) continue;
        float *o = out + rows_lds[pr] * (int64_t)T + t0;
        const float4 v = make_float4(acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]);
        if (t0 + 3 < T && !(T & 3)) {
          *reinterpret_cast<float4 *>(o) = v;
        } else {
          if (t0 < T) o[0] = v.x;
          if (t0 + 1 < T) o[1] = v.y;
          if (t0 + 2 < T) o[2] = v.z;
          if (t0 + 3 < T) o[3] = v.w;
        }
      }
  }
}

constexpr int PL_TOK_TILES = 4;                    // 64-token tiles per workgroup of pair_logits_bwd_tok

__global__ __launch_bounds__(256) void pair_logits_bwd_tok(const float *__restrict__ g, const float *__restrict__ e, PairImgs imgs,
                                                           const int64_t *__restrict__ out_row, float *__restrict__ d_tok, int T)
{
  __shared__ __attribute__((aligned(16))) float lds[PL_PC * PL_LD];
  __shared__ int64_t rows_lds[PL_PC];
  const int b = blockIdx.y, p0 = imgs.start[b], np = imgs.start[b + 1] - p0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, kq = lane >> 4;
  // A[i = n][k = kq] = e[pair 4 s + kq][channel 32 blk + 8 (n / 4) + 4 mt + n % 4]   (M tile u = 2 blk + mt)
  const int chn = 8 * (n >> 2) + (n & 3);
  const bool single = np <= PL_PC;                  // the usual case: the image's pairs are staged once for all token tiles
  float bv[PL_PC / 4], bvn[PL_PC / 4];
  const int tbase = blockIdx.x * (64 * PL_TOK_TILES) + wave * 16 + n;
  // g[row(pair 4 s + kq), t] for the k steps of one pass (rows_lds / cnt of the staged pairs)
  auto load_g = [&](float *dst, int t, int cnt) {
#pragma unroll
    for (int s = 0; s < PL_PC / 4; ++s) {
      const int pr = 4 * s + kq;
      dst[s] = (pr < cnt && t < T) ? g[rows_lds[pr] * (int64_t)T + t] : 0.f;
    }
  };
  if (single && np > 0) {
    stage_pairs(lds, rows_lds, e, out_row, p0, np);
    __syncthreads();
    load_g(bv, tbase, np);
  }
  for (int tile = 0; tile < PL_TOK_TILES; ++tile) {
    const int t = tbase + 64 * tile;
    if (blockIdx.x * (64 * PL_TOK_TILES) + 64 * tile >= T) break;
    if (single && np > 0 && tile + 1 < PL_TOK_TILES) load_g(bvn, t + 64, np);
    v4f acc[PL_C / 16];
#pragma unroll
    for (int u = 0; u < PL_C / 16; ++u) acc[u] = v4f{0.f, 0.f, 0.f, 0.f};
    for (int pc = 0; pc < np; pc += PL_PC) {
      const int cnt = min(PL_PC, np - pc);
      if (!single) {
        __syncthreads();
        stage_pairs(lds, rows_lds, e, out_row, p0 + pc, cnt);
        __syncthreads();
        load_g(bv, t, cnt);
      }
      const int ks = (cnt + 3) >> 2;
#pragma unroll
      for (int s = 0; s < PL_PC / 4; ++s) {
        if (s < ks) {
          const float *ar = lds + (4 * s + kq) * PL_LD + chn;
#pragma unroll
          for (int u = 0; u < PL_C / 16; ++u) acc[u] = mfma4(ar[32 * (u >> 1) + 4 * (u & 1)], bv[s], acc[u]);
        }
      }
    }
    // D[i = 4 kq + r][j = n]: channel 32 blk + 8 kq + 4 mt + r of token t
    if (t < T) {
      float *o = d_tok + ((int64_t)b * T + t) * PL_C + 8 * kq;
#pragma unroll
      for (int u = 0; u < PL_C / 16; ++u)
        *reinterpret_cast<float4 *>(o + 32 * (u >> 1) + 4 * (u & 1)) = make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
    }
#pragma unroll
    for (int s = 0; s < PL_PC / 4; ++s) bv[s] = bvn[s];
  }
}

// ws[(slab x, pair p) , c]: partial d_e of token slab x (slab tokens per workgroup, a multiple of 16)
__global__ __launch_bounds__(256) void pair_logits_bwd_rows(const float *__restrict__ g, const float *__restrict__ tok, PairImgs imgs,
                                                            const int64_t *__restrict__ out_row, float *__restrict__ ws, int T, int slab, int N)
{
  const int b = blockIdx.y, p0 = imgs.start[b], np = imgs.start[b + 1] - p0;
  if (np <= 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, kq = lane >> 4;
  const int tbeg = blockIdx.x * slab, tend = min(T, tbeg + slab);
  const bool vec = !(T & 3);
  const float *tb = tok + (int64_t)b * T * PL_C + 64 * wave + 4 * n;            // B[k][j = n]: channels 64 wave + 4 n + nt
  for (int pc = 0; pc < np; pc += PL_PC) {
    const int cnt = min(PL_PC, np - pc);
    const int mts = (cnt + 15) >> 4;
    const float *grow[3];
    bool valid[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
      const int pr = 16 * mt + n;
      valid[mt] = pr < cnt;
      const int64_t row = valid[mt] ? (out_row ? out_row[p0 + pc + pr] : (int64_t)(p0 + pc + pr)) : 0;
      grow[mt] = g + row * (int64_t)T;
    }
    v4f acc[3][4];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = v4f{0.f, 0.f, 0.f, 0.f};
    // this lane's four tokens ta .. ta + 3 are the k slices of four MFMA steps
    auto load_ab = [&](float4 *aq, float4 *bq, int t0) {
      const int ta = t0 + 4 * kq;
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) {
        aq[mt] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid[mt] && mt < mts) {
          if (vec && ta + 3 < tend) {
            aq[mt] = *reinterpret_cast<const float4 *>(grow[mt] + ta);
          } else {
            if (ta < tend) aq[mt].x = grow[mt][ta];
            if (ta + 1 < tend) aq[mt].y = grow[mt][ta + 1];
            if (ta + 2 < tend) aq[mt].z = grow[mt][ta + 2];
            if (ta + 3 < tend) aq[mt].w = grow[mt][ta + 3];
          }
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) bq[s] = *reinterpret_cast<const float4 *>(tb + (int64_t)min(ta + s, T - 1) * PL_C);
    };
    auto products = [&](const float4 *aq, const float4 *bq) {
#pragma unroll
      for (int mt = 0; mt < 3; ++mt) {
        if (mt < mts) {
          const float a[4] = {aq[mt].x, aq[mt].y, aq[mt].z, aq[mt].w};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            acc[mt][0] = mfma4(a[s], bq[s].x, acc[mt][0]);
            acc[mt][1] = mfma4(a[s], bq[s].y, acc[mt][1]);
            acc[mt][2] = mfma4(a[s], bq[s].z, acc[mt][2]);
            acc[mt][3] = mfma4(a[s], bq[s].w, acc[mt][3]);
          }
        }
      }
    };
    // every load is a new 256-byte piece of four token rows (nothing is re-read), so two 16-token groups are kept in flight behind the
    // one being multiplied; groups past the slab's end load zeros for g (guards above) and add nothing
    float4 a0[3], b0[4], a1[3], b1[4], a2[3], b2[4];
    load_ab(a0, b0, tbeg);
    load_ab(a1, b1, tbeg + 16);
    for (int t0 = tbeg; t0 < tend; t0 += 48) {
      load_ab(a2, b2, t0 + 32);
      products(a0, b0);
      load_ab(a0, b0, t0 + 48);
      products(a1, b1);
      load_ab(a1, b1, t0 + 64);
      products(a2, b2);
    }
    // D[i = 4 kq + r][j = n]: pair 16 mt + 4 kq + r, channels 64 wave + 4 n + nt
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int pr = 16 * mt + 4 * kq + r;
        if (pr < cnt)
          *reinterpret_cast<float4 *>(ws + ((int64_t)blockIdx.x * N + p0 + pc + pr) * PL_C + 64 * wave + 4 * n) =
              make_float4(acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]);
      }
  }
}

// d_e = sum over the slabs, in a fixed order: 16 interleaved partial sums per output (228 dependent 16-byte loads per thread otherwise), then
// the 16 partials in order
__global__ __launch_bounds__(256) void pair_logits_rows_reduce(const float *__restrict__ ws, float *__restrict__ d_e, int64_t n4, int nslab)
{
  __shared__ float4 red[16][16];
  const int part = threadIdx.x >> 4, li = threadIdx.x & 15;
  const int64_t i = (int64_t)blockIdx.x * 16 + li;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const float4 *w = reinterpret_cast<const float4 *>(ws) + i;
    for (int x = part; x < nslab; x += 16) {
      const float4 v = w[(int64_t)x * n4];
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
  }
  red[part][li] = s;
  __syncthreads();
  if (part == 0 && i < n4) {
#pragma unroll
    for (int p = 1; p < 16; ++p) {
      const float4 v = red[p][li];
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    reinterpret_cast<float4 *>(d_e)[i] = s;
  }
}

constexpr int PL_SLAB = 288;                       // tokens per workgroup of pair_logits_bwd_rows: a multiple of its 48-token ring

int pl_check(const char *what, const void *a, const void *b2, const void *c, const int32_t *img_start, int B, int T, int C, int N, PairImgs *imgs)
{
  if (B < 0 || B > PD_PAIR_LOGITS_MAX_IMAGES || T < 0 || C != PL_C || N < 0)
    return pd_set_error(PD_ERR_INVALID_ARG, "%s: B=%d (<= %d) T=%d C=%d (== %d) N=%d", what, B, PD_PAIR_LOGITS_MAX_IMAGES, T, C, PL_C, N);
  if (!img_start) return pd_set_error(PD_ERR_INVALID_ARG, "%s: null img_start", what);
  for (int i = 0; i <= B; ++i) {
    imgs->start[i] = img_start[i];
    if (img_start[i] < 0 || img_start[i] > N || (i && img_start[i] < img_start[i - 1]) || (!i && img_start[0] != 0))
      return pd_set_error(PD_ERR_INVALID_ARG, "%s: img_start[%d] = %d is not a non-decreasing offset into %d pairs", what, i, img_start[i], N);
  }
  if (img_start[B] != N) return pd_set_error(PD_ERR_INVALID_ARG, "%s: img_start[B] = %d != N = %d", what, img_start[B], N);
  if (B && T && (!a || !b2 || !c) && N) return pd_set_error(PD_ERR_INVALID_ARG, "%s: null pointer", what);
  if ((((uintptr_t)a | (uintptr_t)b2 | (uintptr_t)c) & 15)) return pd_set_error(PD_ERR_INVALID_ARG, "%s: pointers must be 16-byte aligned", what);
  return PD_OK;
}
}  // namespace

extern "C" int pd_pair_logits_fwd(const float *tok, const float *e, const int32_t *img_start, const int64_t *out_row, float *out, int B, int T,
                                  int C, int N, void *stream)
{
  PairImgs imgs;
  const int rc = pl_check("pd_pair_logits_fwd", tok, e, out, img_start, B, T, C, N, &imgs);
  if (rc != PD_OK) return rc;
  if (!B || !T || !N) return PD_OK;
  hipLaunchKernelGGL(pair_logits_fwd, dim3((unsigned)((T + 255) / 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream, tok, e, imgs, out_row, out, T);
  return pd_check_launch("pd_pair_logits_fwd");
}

extern "C" int pd_pair_logits_bwd_tok(const float *g, const float *e, const int32_t *img_start, const int64_t *out_row, float *d_tok, int B, int T,
                                      int C, int N, void *stream)
{
  PairImgs imgs;
  const int rc = pl_check("pd_pair_logits_bwd_tok", N ? g : (const float *)d_tok, N ? e : (const float *)d_tok, d_tok, img_start, B, T, C, N, &imgs);
  if (rc != PD_OK) return rc;
  if (!B || !T) return PD_OK;
  if (!d_tok) return pd_set_error(PD_ERR_INVALID_ARG, "pd_pair_logits_bwd_tok: null d_tok");
  hipLaunchKernelGGL(pair_logits_bwd_tok, dim3((unsigned)((T + 64 * PL_TOK_TILES - 1) / (64 * PL_TOK_TILES)), (unsigned)B), dim3(256), 0, (hipStream_t)stream, g, e, imgs, out_row,
                     d_tok, T);
  return pd_check_launch("pd_pair_logits_bwd_tok");
}

extern "C" int64_t pd_pair_logits_workspace_floats(int T, int C, int N)
{
  return (int64_t)((T + PL_SLAB - 1) / PL_SLAB) * N * C;
}

extern "C" int pd_pair_logits_bwd_rows(const float *g, const float *tok, const int32_t *img_start, const int64_t *out_row, float *d_e,
                                       float *workspace, int B, int T, int C, int N, void *stream)
{
  PairImgs imgs;
  const int rc = pl_check("pd_pair_logits_bwd_rows", g, tok, d_e, img_start, B, T, C, N, &imgs);
  if (rc != PD_OK) return rc;
  if (!N) return PD_OK;
  if (!workspace || ((uintptr_t)workspace & 15)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_pair_logits_bwd_rows: workspace null or not 16-byte aligned");
  const int nslab = (T + PL_SLAB - 1) / PL_SLAB;
  if (nslab && B)
    hipLaunchKernelGGL(pair_logits_bwd_rows, dim3((unsigned)nslab, (unsigned)B), dim3(256), 0, (hipStream_t)stream, g, tok, imgs, out_row, workspace, T, PL_SLAB, N);
  const int64_t n4 = (int64_t)N * C / 4;
  hipLaunchKernelGGL(pair_logits_rows_reduce, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, (hipStream_t)stream, workspace, d_e, n4, B ? nslab : 0);
  return pd_check_launch("pd_pair_logits_bwd_rows");
}
