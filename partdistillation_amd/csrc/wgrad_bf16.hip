// dW[N, K] = dY[M, N]^T X[M, K] in bf16 with fp32 accumulation (C-ABI: pd_wgrad_bf16 in include/pd_igemm.h): the weight gradient of an
// nn.Linear over the M tokens of a Swin stage (modeling/backbone/swin.py:34-36, 127-129, 312), of the decoder's key / value projections
// over the memory tokens, of a 1 x 1 convolution over M pixels.
//
// The contraction runs over the ROWS of both operands — the dimension in which neither is contiguous — while an MFMA lane wants 8
// consecutive contraction elements of one column.  Like csrc/conv_bf16.hip's filter-gradient kernel the tiles are staged AS THEY LIE in
// memory ([row][column]) and transposed on the way out of LDS by ds_read_b64_tr_b16 (lane map: tools/probes/tr_read_probe.hip, profiles/
// r02_tr_read_probe.txt: within 16 lanes, lane s addresses row s / 4, columns 4 (s % 4) .. +3 of a [4 rows][16 columns] block and receives
// column s of the 4 rows; two reads = one 32x32x16 operand).  What is new (the igemm_bf16.hip recipe):
//   * tiles go global -> LDS directly (global_load_lds_dwordx4, 1 KB = 4 rows of a 128-column tile per wavefront instruction).  The LDS
//     image of such a load is lane-linear, but WHICH 16 bytes a lane fetches is free: the 64 lanes of an instruction fetch their 4 x 128
//     piece as eight dense [4 rows][16 columns] blocks, the natural source of the transpose read (32 lanes then read 256 contiguous bytes).
//     (First attempt: row-major 256-byte rows with an XOR swizzle of the 16-byte chunks — correct, and 3.3 us per 64-row stage: the
//     transpose read has conflict classes beyond the plain bank rule, as cdna_hip_programming.md T10 warns.)
//   * rows beyond the slice and columns beyond N / K load from a zero line;
//   * 128 x 128 output tile, 64 rows per stage, one stage and four workgroups per CU (or two stages and two);
//   * the M rows are cut into slices; a slice's fp32 partial tile goes to a slab and the LAST workgroup of a tile to arrive (agent-scope
//     release / ticket / acquire) sums the slabs in slice order (deterministic), applies an optional per-row scale and writes bf16;
//   * the bias gradient (column sums of dY): a lane's dY fragment is 8 rows of one column — summed on the vector pipe in the workgroups of
//     the first column tile.
// Measured before this kernel (tools/bench_swin_wgrad.py, Swin-B stage 3 fc1, 10 368 x 512 -> 2 048): split-rows skinny kernel 104 us,
// transpose-read convolution kernel 152, library 74 (21.7 GFLOP: 290 TFLOP/s at best).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mfma_bf16.h"
#include "pd_common.h"
#include "xcd.h"
#include "pd_igemm.h"
#include "pd_msda.h"

int g_wg_nst = 0, g_wg_splits = 0, g_wg_mode = 0;                       // pd_debug_set "wg_nst" / "wg_splits" (tools/ only; 0 = automatic)

namespace {
using namespace pdmfma;

typedef __bf16 hwbf16x8 __attribute__((ext_vector_type(8)));
typedef short v4s16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr;
typedef __attribute__((address_space(1))) const void *glb_ptr;
typedef __attribute__((address_space(3))) v4s16 *lds_v4;

__device__ __attribute__((aligned(256))) unsigned char g_wg_zero_line[256];

constexpr int TILE = 128, ROWS = 64, ROWB = 256;         // output tile edge, rows per stage, bytes per LDS row

struct WgArgs {
  const bf16_t *dY, *X;
  bf16_t *dW;
  float *dB;
  const float *rscale;
  float *slabs;
  int M, N, K, ldy, ldx, ldw, dw_f32;
  int tiles_k, tiles, splits, rows_per_split;
  int mode;                                              // tools/ only, bits: 1 no arithmetic, 2 no global loads, 4 no epilogue
};

__device__ __forceinline__ int xcd_chunk(int bid, int nb) { return pd_xcd_chunk(bid, nb); }   // xcd.h: any workgroup count

typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

// The transpose reads are issued as inline assembly with hand-placed s_waitcnt lgkmcnt: through the builtin the compiler cannot tell a
// read of the CURRENT stage from the global->LDS loads in flight for the NEXT one and puts s_waitcnt vmcnt(0) in front of the first read
// (seen in the ISA: the prefetch of the two-stage loop was fully serialised), and it waits for every read before the first MFMA of a step.
struct Frags { v2i y[2][2], x[2][2]; };                  // [32-column tile][rows 0-3 | 4-7 of the lane's 8]

template <int OFF>
__device__ __forceinline__ void tr_issue(Frags &f, const unsigned (&ya)[2], const unsigned (&xa)[2])
{
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.y[i][0]) : "v"(ya[i]), "n"(OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.y[i][1]) : "v"(ya[i]), "n"(OFF + 1024));
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.x[i][0]) : "v"(xa[i]), "n"(OFF + 16384));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.x[i][1]) : "v"(xa[i]), "n"(OFF + 16384 + 1024));
  }
}
// LDS operations return in order: "at most PENDING outstanding" = everything issued before the last PENDING reads has arrived.  The fragments
// are in/out operands so that nothing that uses them can be scheduled above the wait.
template <int PENDING>
__device__ __forceinline__ void tr_wait(Frags &f)
{
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(f.y[0][0]), "+v"(f.y[0][1]), "+v"(f.y[1][0]), "+v"(f.y[1][1]), "+v"(f.x[0][0]), "+v"(f.x[0][1]), "+v"(f.x[1][0]), "+v"(f.x[1][1])
               : "n"(PENDING));
}
__device__ __forceinline__ hwbf16x8 frag_of(const v2i &lo, const v2i &hi)
{
  union { v4i i; hwbf16x8 v; } u;
  u.i = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
  return u.v;
}

constexpr int SLAB_FLOATS = TILE * TILE + TILE;          // partial tile + partial column sums

template <int NST>
__global__ __launch_bounds__(256, NST == 1 ? 3 : 2) void wgrad_bf16(WgArgs a)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = 2 * ROWS * ROWB;                 // dY rows then X rows: 32 KB
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lb = xcd_chunk(blockIdx.x, gridDim.x);
  const int tile = lb / a.splits, split = lb - tile * a.splits;
  const int tn = tile / a.tiles_k, tk = tile - tn * a.tiles_k;
  const int n0 = tn * TILE, k0 = tk * TILE;
  const int m_begin = split * a.rows_per_split, m_end = min(a.M, m_begin + a.rows_per_split);
  const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;
  // ---- loads: instruction j of wavefront w fills row group rg = w * 4 + j (4 rows x 128 columns = 1 KB) as EIGHT dense [4 rows][16 columns]
  // blocks of 128 bytes — the natural source of the transpose read (lane s of a 16-lane group addresses byte 8 s of its block, so 32 lanes read
  // 256 contiguous bytes = every bank once).  Lane l of the load: block l / 8, row (l % 8) / 2 of the block, 8-column half l % 2.
  const int lr = (lane & 7) >> 1;
  const int csrc = (lane >> 3) * 16 + (lane & 1) * 8;    // source column (elements) of this lane's 16-byte chunk
  const bool yok = n0 + csrc < a.N, xok = k0 + csrc < a.K;
  const bf16_t *zl = reinterpret_cast<const bf16_t *>(g_wg_zero_line) + (lane & 15) * 8;
  const bf16_t *yb = a.dY + n0 + csrc, *xb = a.X + k0 + csrc;
  auto issue = [&](int m0, int buf) {
    unsigned char *Ys = smem + buf * STAGE, *Xs = Ys + ROWS * ROWB;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (wave * 4 + j) * 4 + lr, m = m0 + row;
      const bool in = m < m_end;
      __builtin_amdgcn_global_load_lds((glb_ptr)(in && yok ? yb + (int64_t)m * a.ldy : zl), (lds_ptr)(Ys + (wave * 4 + j) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr)(in && xok ? xb + (int64_t)m * a.ldx : zl), (lds_ptr)(Xs + (wave * 4 + j) * 1024), 16, 0, 0);
    }
  };
  // ---- fragments: lane l -> column l % 32 of its 32-column tile, rows 8 (l / 32) .. +7 of the 16-row step
  //      = block (row group 2 (l / 32) of the step [+ 1 for the second read], column block (tile column / 16) + (l / 16) % 2), byte 8 (l % 16)
  const int grp = lane >> 4, sl = lane & 15;
  int yoff[2], xoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    yoff[i] = 2 * (grp >> 1) * 1024 + (((wn + i * 32) >> 4) + (grp & 1)) * 128 + sl * 8;
    xoff[i] = 2 * (grp >> 1) * 1024 + (((wk + i * 32) >> 4) + (grp & 1)) * 128 + sl * 8;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[i][0][e] = 0.f; acc[i][1][e] = 0.f; }
  }
  // bias gradient: the dY fragment of a lane IS 8 rows of column n = l % 32 — summed on the vector pipe (two registers) in the wavefronts
  // of the first column tile
  const bool do_bias = a.dB != nullptr && tk == 0 && wk == 0;
  float bsum[2] = {0.f, 0.f};

  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
  auto step = [&](const Frags &f) {
    hwbf16x8 ya[2], xv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { ya[i] = frag_of(f.y[i][0], f.y[i][1]); xv[i] = frag_of(f.x[i][0], f.x[i][1]); }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya[i], xv[j], acc[i][j], 0, 0, 0);
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[i] += (float)ya[i][e];
    }
  };
  // one stage = four 16-row steps; the reads of step s + 1 are in flight while the matrix cores work on step s
  auto compute = [&](int buf) {
    unsigned ya[2], xa[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { ya[i] = lds0 + buf * STAGE + yoff[i]; xa[i] = lds0 + buf * STAGE + xoff[i]; }
    Frags f0, f1;
    tr_issue<0>(f0, ya, xa);
    tr_issue<4096>(f1, ya, xa);
    tr_wait<8>(f0);
    step(f0);
    tr_issue<8192>(f0, ya, xa);
    tr_wait<8>(f1);
    step(f1);
    tr_issue<12288>(f1, ya, xa);
    tr_wait<8>(f0);
    step(f0);
    tr_wait<0>(f1);
    step(f1);
  };

  if (NST >= 2) {
    // ring of NST stages, NST - 1 of them in flight: a CU needs ~100 KB of loads in flight to keep its share of the L2 -> LDS path busy
    // (0.8 us round trip measured with one 32 KB stage in flight per CU: a quarter of the path's rate)
    const int nstage = (m_end - m_begin + ROWS - 1) / ROWS;
#pragma unroll
    for (int p = 0; p < NST - 1; ++p)
      if (p < nstage && !(a.mode & 2)) issue(m_begin + p * ROWS, p);
    int buf = 0, nxt = NST - 1;
    for (int i = 0; i < nstage; ++i) {
      // stage i has landed when at most the 8 loads per stage of the stages issued after it are outstanding
      const int after = min(NST - 2, nstage - 1 - i);
      if (after >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (after == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (i + NST - 1 < nstage && !(a.mode & 2)) issue(m_begin + (i + NST - 1) * ROWS, nxt);
      if (!(a.mode & 1)) compute(buf);
      buf = buf + 1 == NST ? 0 : buf + 1;
      nxt = nxt + 1 == NST ? 0 : nxt + 1;
    }
  } else {
    for (int m0 = m_begin; m0 < m_end; m0 += ROWS) {
      if (!(a.mode & 2)) issue(m0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (!(a.mode & 1)) compute(0);
      __syncthreads();
    }
  }

  if (a.mode & 4) {                                      // tools/ only: the main loop without the epilogue (keeps the accumulators alive)
    if (acc[0][0][0] + acc[0][1][1] + acc[1][0][2] + acc[1][1][3] == 123.456f) a.dW[t] = 0;
    return;
  }
  // result layout: lane -> column k = l % 32 of tile j, register e -> row n = (e & 3) + 8 (e >> 2) + 4 (l / 32) of tile i
  // (bias sums: lane -> column n = l % 32 of dY tile i, halves l / 32 = rows 0-7 / 8-15)
  const int fr = lane & 31, kh = lane >> 5;
  if (a.splits > 1) {
    float *slab = a.slabs + ((int64_t)tile * a.splits + split) * SLAB_FLOATS;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          reinterpret_cast<float4 *>(slab)[((i * 2 + j) * 4 + q) * 256 + t] =
              make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float v = bsum[i] + __shfl_xor(bsum[i], 32, 64);      // rows 0-7 + rows 8-15 of every step
        if (kh == 0) slab[TILE * TILE + wn + i * 32 + fr] = v;
      }
    }
    return;                                              // wgrad_bf16_reduce (the next launch) sums the slabs
  } else if (do_bias) {
#pragma unroll
    for (int i = 0; i < 2; ++i) bsum[i] += __shfl_xor(bsum[i], 32, 64);
  }
  if (do_bias && kh == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n = n0 + wn + i * 32 + fr;
      if (n < a.N) a.dB[n] += bsum[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int n = n0 + wn + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
      if (n >= a.N) continue;
      const float rs = a.rscale ? a.rscale[n] : 1.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int k = k0 + wk + j * 32 + fr;
        if (k < a.K) {
          if (a.dw_f32) reinterpret_cast<float *>(a.dW)[(int64_t)n * a.ldw + k] = acc[i][j][e] * rs;
          else a.dW[(int64_t)n * a.ldw + k] = (bf16_t)(pk_bf16(acc[i][j][e] * rs, 0.f) & 0xffffu);
        }
      }
    }
  }
}

// The slabs of a tile summed in slice order (deterministic) -> bf16 dW and fp32 dB.  A slab keeps the main kernel's register order: float4 number
// ((i * 2 + j) * 4 + q) * 256 + t = rows 8 q + 4 (lane / 32) + 0..3, column lane % 32 of the 32 x 32 piece (i, j) of wavefront t / 64's 64 x 64 quarter.
// (A first version let the LAST workgroup of a tile to
// arrive — agent-scope release / ticket / acquire, as csrc/igemm_bf16.hip does for its few-tile problems — sum the slabs inside the main
// kernel: 17 us of a 58 us launch at 4 slices of 64 tiles, 200 us at 16 slices of 256 tiles; one workgroup walking `splits` x 64 KB with 16
// loads in flight, plus an L2 write-back per release and an invalidate per acquire.)
__device__ __forceinline__ void wgrad_reduce_body(const WgArgs &a, const int block)
{
  // 64 float4 positions of a tile per workgroup; the slices are cut into four runs, one per wavefront (8 loads in flight per thread), whose
  // sums are added in run order through LDS — few-tile problems have 32..128 slices and only tiles x 64 workgroups to hide the latency with
  __shared__ float4 part[4][64];
  const int t = threadIdx.x, pos = t & 63, g = t >> 6;
  const int tile = block >> 6, sub = block & 63;
  const int tn = tile / a.tiles_k, tk = tile - tn * a.tiles_k;
  const float *s0 = a.slabs + (int64_t)tile * a.splits * SLAB_FLOATS;
  const int run = (a.splits + 3) >> 2, s_end = min(a.splits, (g + 1) * run);
  const int f = sub * 64 + pos;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  int s = g * run;
  for (; s + 8 <= s_end; s += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const float4 *>(s0 + (int64_t)(s + u) * SLAB_FLOATS)[f];
#pragma unroll
    for (int u = 0; u < 8; ++u) { sum.x += v[u].x; sum.y += v[u].y; sum.z += v[u].z; sum.w += v[u].w; }
  }
  for (; s < s_end; ++s) {
    const float4 v = reinterpret_cast<const float4 *>(s0 + (int64_t)s * SLAB_FLOATS)[f];
    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
  }
  part[g][pos] = sum;
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int u = 1; u < 4; ++u) { const float4 v = part[u][pos]; sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w; }
    const int piece = sub >> 2, wave = sub & 3, lane = pos;           // the main kernel's thread wave * 64 + lane, float4 number `piece`
    const int i = piece >> 3, j = (piece >> 2) & 1, q = piece & 3;
    const int n = tn * TILE + (wave >> 1) * 64 + i * 32 + 8 * q + 4 * (lane >> 5);
    const int k = tk * TILE + (wave & 1) * 64 + j * 32 + (lane & 31);
    if (k < a.K) {
      const float r[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e >= a.N) break;
        const float v = r[e] * (a.rscale ? a.rscale[n + e] : 1.f);
        if (a.dw_f32) reinterpret_cast<float *>(a.dW)[(int64_t)(n + e) * a.ldw + k] = v;
        else a.dW[(int64_t)(n + e) * a.ldw + k] = (bf16_t)(pk_bf16(v, 0.f) & 0xffffu);
      }
    }
  } else if (a.dB != nullptr && sub == 0 && tk == 0) {
    const int c = t - 64;                                             // wavefronts 1 and 2: the tile's 128 column sums of dY
    if (c < TILE && tn * TILE + c < a.N) {
      float b = 0.f;
      for (int u = 0; u < a.splits; ++u) b += s0[(int64_t)u * SLAB_FLOATS + TILE * TILE + c];
      a.dB[tn * TILE + c] += b;
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_bf16_reduce(WgArgs a) { wgrad_reduce_body(a, (int)blockIdx.x); }

// the reduce passes of up to PD_WGRAD_SEQ_MAX problems as ONE launch (pd_wgrad_bf16_seq): the four weight gradients of a Swin block each
// paid a ~10 us second launch of a few hundred small workgroups
struct WgGroup { WgArgs a[PD_WGRAD_SEQ_MAX]; int first_block[PD_WGRAD_SEQ_MAX + 1]; int count; };
__global__ __launch_bounds__(256) void wgrad_bf16_reduce_grouped(WgGroup g)
{
  int i = 0;
  while (i + 1 < g.count && (int)blockIdx.x >= g.first_block[i + 1]) ++i;        // (block-uniform)
  wgrad_reduce_body(g.a[i], (int)blockIdx.x - g.first_block[i]);
}

struct WgPlan { WgArgs a; int nst; int64_t slab_bytes; };
constexpr int64_t WG_HEADER_BYTES = 16384;             // left untouched: the workspace may be the one pd_igemm_bf16 keeps its (zero) tickets in

int wg_plan(const PdWgrad *p, WgPlan &pl)
{
  if (!p || !p->dy || !p->x || !p->dw) return pd_set_error(PD_ERR_INVALID_ARG, "pd_wgrad_bf16: null pointer");
  if (p->m <= 0 || p->n <= 0 || p->k <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_wgrad_bf16: empty problem");
  if ((p->n & 7) || (p->k & 7) || (p->ldy & 7) || (p->ldx & 7) || p->ldy < p->n || p->ldx < p->k || p->ldw < p->k)
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_wgrad_bf16: n, k and the operands' row strides must be multiples of 8 (n=%d k=%d ldy=%d ldx=%d ldw=%d)", p->n,
                        p->k, p->ldy, p->ldx, p->ldw);
  if (((uintptr_t)p->dy | (uintptr_t)p->x) & 15) return pd_set_error(PD_ERR_INVALID_ARG, "pd_wgrad_bf16: operands must be 16-byte aligned");
  WgArgs &a = pl.a;
  a.dY = (const bf16_t *)p->dy; a.X = (const bf16_t *)p->x; a.dW = (bf16_t *)p->dw; a.dB = p->db; a.rscale = p->row_scale;
  a.slabs = nullptr; a.mode = g_wg_mode;
  a.M = p->m; a.N = p->n; a.K = p->k; a.ldy = p->ldy; a.ldx = p->ldx; a.ldw = p->ldw; a.dw_f32 = p->dw_f32 != 0;
  a.tiles_k = (a.K + TILE - 1) / TILE;
  a.tiles = a.tiles_k * ((a.N + TILE - 1) / TILE);
  // slices of the rows.  Measured (tools/debug/wgrad_modes.py, tools/bench_swin_wgrad.py): a CU works off a 64-row stage of a tile in ~0.7 us
  // with two resident workgroups (two stages of 32 KB each) and ~0.9 us with one; a slice costs its 64 KB slab written and read back,
  // ~0.27 us per MB on top of ~5 us for the second launch.  So: fill the 512 resident slots once, but keep runs of >= 512 rows (>= 1024
  // when there are so few tiles that the slabs of ONE tile are the reduce kernel's serial chain).
  int rmin = a.tiles <= 16 ? 1024 : 512;
  if (a.tiles * (a.M / rmin > 1 ? a.M / rmin : 1) < 32) rmin = 256;     // a few tiles over a few thousand rows (the decoder's 2 000-row MLP: 4 tiles
                                                                         // ran as 4 workgroups x 32 stages, 42 us): shorter runs, 7 slabs per tile
  int splits = a.tiles >= 512 ? 1 : 512 / a.tiles;
  if (splits > a.M / rmin) splits = a.M / rmin;
  if (splits < 1) splits = 1;
  if (g_wg_splits > 0) splits = g_wg_splits;
  if (a.tiles > 4096) splits = 1;
  a.rows_per_split = ((a.M + splits - 1) / splits + ROWS - 1) / ROWS * ROWS;
  a.splits = (a.M + a.rows_per_split - 1) / a.rows_per_split;
  pl.nst = 2;
  if (g_wg_nst == 1 || g_wg_nst == 2) pl.nst = g_wg_nst;
  pl.slab_bytes = a.splits > 1 ? (int64_t)a.tiles * a.splits * SLAB_FLOATS * 4 : 0;
  return PD_OK;
}

template <int NST>
int wg_launch(const WgPlan &pl, hipStream_t st)
{
  constexpr size_t lds = (size_t)NST * 2 * ROWS * ROWB;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)wgrad_bf16<NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL((wgrad_bf16<NST>), dim3((unsigned)((int64_t)pl.a.tiles * pl.a.splits)), dim3(256), lds, st, pl.a);
  return pd_check_launch("pd_wgrad_bf16");
}
}  // namespace

extern "C" int64_t pd_wgrad_bf16_workspace_bytes(const PdWgrad *p)
{
  WgPlan pl;
  if (wg_plan(p, pl) != PD_OK) return -1;
  return pl.slab_bytes ? pl.slab_bytes + WG_HEADER_BYTES : 0;
}

extern "C" int pd_wgrad_bf16(const PdWgrad *p, void *workspace, int64_t workspace_bytes, void *stream)
{
  WgPlan pl;
  const int rc = wg_plan(p, pl);
  if (rc != PD_OK) return rc;
  if (pl.slab_bytes) {
    if (!workspace || workspace_bytes < pl.slab_bytes + WG_HEADER_BYTES)
      return pd_set_error(PD_ERR_INVALID_ARG, "pd_wgrad_bf16: workspace of %lld bytes needed (%lld given)", (long long)(pl.slab_bytes + WG_HEADER_BYTES),
                          (long long)workspace_bytes);
    pl.a.slabs = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(workspace) + WG_HEADER_BYTES);
  }
  hipStream_t st = (hipStream_t)stream;
  const int rc2 = pl.nst == 1 ? wg_launch<1>(pl, st) : wg_launch<2>(pl, st);
  if (rc2 != PD_OK || pl.a.splits == 1 || (pl.a.mode & 4)) return rc2;
  hipLaunchKernelGGL(wgrad_bf16_reduce, dim3((unsigned)(pl.a.tiles * 64)), dim3(256), 0, st, pl.a);
  return pd_check_launch("pd_wgrad_bf16 (reduce)");
}

extern "C" int64_t pd_wgrad_bf16_seq_workspace_bytes(const PdWgrad *list, int count)
{
  if (!list || count <= 0 || count > PD_WGRAD_SEQ_MAX) return -1;
  int64_t slabs = 0;
  for (int i = 0; i < count; ++i) {
    WgPlan pl;
    if (wg_plan(list + i, pl) != PD_OK) return -1;
    slabs += pl.slab_bytes;
  }
  return slabs ? slabs + WG_HEADER_BYTES : 0;
}

extern "C" int pd_wgrad_bf16_seq(const PdWgrad *list, int count, void *workspace, int64_t workspace_bytes, void *stream)
{
  if (!list || count <= 0 || count > PD_WGRAD_SEQ_MAX) return pd_set_error(PD_ERR_INVALID_ARG, "pd_wgrad_bf16_seq: 1..%d problems", PD_WGRAD_SEQ_MAX);
  WgPlan pl[PD_WGRAD_SEQ_MAX];
  int64_t need = 0;
  for (int i = 0; i < count; ++i) {
    const int rc = wg_plan(list + i, pl[i]);
    if (rc != PD_OK) return rc;
    need += pl[i].slab_bytes;
  }
  if (need && (!workspace || workspace_bytes < need + WG_HEADER_BYTES))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_wgrad_bf16_seq: workspace of %lld bytes needed (%lld given)", (long long)(need + WG_HEADER_BYTES), (long long)workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  unsigned char *slab = reinterpret_cast<unsigned char *>(workspace) + WG_HEADER_BYTES;
  WgGroup g;
  g.count = 0;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {                      // every problem its own slab range: the reduce passes run after ALL the main kernels
    if (pl[i].slab_bytes) { pl[i].a.slabs = reinterpret_cast<float *>(slab); slab += pl[i].slab_bytes; }
    const int rc = pl[i].nst == 1 ? wg_launch<1>(pl[i], st) : wg_launch<2>(pl[i], st);
    if (rc != PD_OK) return rc;
    if (pl[i].a.splits > 1 && !(pl[i].a.mode & 4)) {
      g.a[g.count] = pl[i].a; g.first_block[g.count] = blocks; blocks += pl[i].a.tiles * 64; ++g.count;
    }
  }
  if (!g.count) return PD_OK;
  g.first_block[g.count] = blocks;
  hipLaunchKernelGGL(wgrad_bf16_reduce_grouped, dim3((unsigned)blocks), dim3(256), 0, st, g);
  return pd_check_launch("pd_wgrad_bf16_seq (reduce)");
}

extern "C" int pd_wgrad_bf16_time(const PdWgrad *p, void *workspace, int64_t workspace_bytes, int iters, float *us, void *stream)
{
  if (!us || iters <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_wgrad_bf16_time: bad arguments");
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return pd_set_error(PD_ERR_LAUNCH, "pd_wgrad_bf16_time: events");
  int rc = PD_OK;
  for (int i = 0; i < 3 && rc == PD_OK; ++i) rc = pd_wgrad_bf16(p, workspace, workspace_bytes, stream);
  (void)hipEventRecord(e0, (hipStream_t)stream);
  for (int i = 0; i < iters && rc == PD_OK; ++i) rc = pd_wgrad_bf16(p, workspace, workspace_bytes, stream);
  (void)hipEventRecord(e1, (hipStream_t)stream);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *us = ms * 1e3f / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return rc;
}
