// The ResNet stem (7 x 7 / 2 convolution, 3 -> 64 channels, + frozen-BN affine + ReLU) and its filter gradient on the bf16 matrix cores
// (include/pd_stem.h).  Round 5: the last library convolution of BASELINE config 2 (MIOpen igemm_fwd_gtc / igemm_wrw_gtc + SubTensorOp fills,
// 0.2 ms per step and a shipped find-db) becomes two kernels of this library.
//
// Shape of the problem: 524 288 output pixels x 64 channels x K = 147 at 2 x 1024^2 — 9.9 GFLOP, nothing for the matrix cores; the layer is
// its 67 MB bf16 output (forward) and two 67 MB reads (backward).  Both kernels tile the OUTPUT by row segments of 128 pixels: such a tile
// reads a 7-row x 261-column x 3-channel patch of the image (11 KB), staged in LDS once.  With the contraction ordered (ky, kx, c) — the
// channels-last filter's own order — the 21 (kx, c) values of a filter row at output pixel p are the 21 CONSECUTIVE patch elements from
// 6 p on, so a filter row is padded to 24 = three 8-element MFMA operand groups (the 3 extra elements meet zero weights) and an operand
// of `v_mfma_f32_32x32x16_bf16` is four aligned 4-byte LDS reads.  K' = 7 x 24 = 168 -> 11 steps of 16 (the 22nd group is all-zero weights).
//
//   forward : C^T[channel][pixel] = W . X^T — the accumulator then holds 4 consecutive CHANNELS of one pixel per register quad (lane =
//             pixel), the layout the NHWC store wants; the 64 x 176 filter sits in registers (88 VGPRs) for the life of a persistent
//             workgroup; epilogue scale / bias / ReLU / bf16.
//   gradient: dW[channel][k'] = sum over pixels — the contraction runs over pixels, along which neither operand is contiguous: both tiles are laid
//             out in LDS pixel-major as they are produced (the masked gradient [pixel][64], the im2col rows [pixel][7 x 24] copied 48 bytes at a
//             time from the patch) and transposed on the way out by ds_read_b64_tr_b16.  The tile's (y > 0 ? gy : 0) is formed while staging (no
//             frozen-BN / ReLU backward pass: the stem has no input gradient), partial sums stay in registers over a workgroup's tiles and
//             leave as ONE fp32 block per workgroup; a second kernel adds the blocks in a fixed order, applies the frozen-BN scale and rounds.
//   Measured at 2 x 1024^2 (rocprofv3, tools/bench_stem.py): forward 55 us (MIOpen 66 + frozen-BN / ReLU pass 21 + input cast 8), gradient 60 + 8 us
//   (MIOpen 113 + its fill 13 + frozen-BN / ReLU backward 30).  First versions: 85 us forward with per-lane 2-byte global loads of the filter
//   (700 uncoalesced wave loads per workgroup) and 8-byte stores at a 128-byte lane stride; 220 us gradient with 2-byte LDS gathers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pd_common.h"
#include "pd_msda.h"
#include "pd_stem.h"

namespace {
typedef unsigned short bf16_t;
typedef __bf16 hwbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hwbf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TP = 128;                    // output pixels of a tile (one row segment)
constexpr int PROW = (2 * TP + 6) * 3;     // 786 patch elements per input row (262 columns: one left of the first window, see stage_patch)
constexpr int PITCH = 800;                 // LDS pitch of a patch row (elements); reads reach 6 * 127 + 4 + 23 = 789
constexpr int KROW = 24;                   // a filter row's 21 (kx, c) values padded to three groups of 8

constexpr int WG_WGRAD = 512;              // workgroups of the filter-gradient kernel (64 KB of LDS each: two per CU, their phases overlap)

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi)
{
  const f32x2 x = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(x, hwbf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pk_bf16(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float((unsigned)u << 16); }

// the 7-row x 262-column x 3-channel patch of tile (b, oy, ox0) -> LDS, zeros outside the image.  LDS element 1 + 3 (ix - (2 ox0 - 4)) + c of
// a row: the patch starts one column left of the first window (2 ox0 - 4: a multiple of 4 ELEMENTS into the image row when W % 4 == 0, so
// the fast path moves 16-byte / 8-byte pieces that lie wholly inside or wholly outside the row) and is shifted by one element, which puts
// output pixel p's window at the EVEN element 6 p + 4 (4-byte LDS reads).
constexpr int PBASE = 4;                   // patch element of pixel 0's window start
template <bool XF32>
__device__ __forceinline__ void stage_patch(bf16_t *patch, const void *x, int b, int oy, int ox0, int H, int W, int t, int nth, bool fast)
{
  const int ix0 = 2 * ox0 - 4;
  if (fast) {
    constexpr int NCH = (PROW + 3 + 3) / 4;                        // 197 pieces of 4 elements cover the row's 786
    for (int idx = t; idx < 7 * NCH; idx += nth) {
      const int ky = idx / NCH, m = idx - ky * NCH;
      const int iy = 2 * oy - 3 + ky, el = ix0 * 3 + 4 * m;        // element offset inside the image row
      unsigned lo = 0u, hi = 0u;                                   // 4 bf16
      if (iy >= 0 && iy < H && el >= 0 && el < 3 * W) {
        const int64_t g = ((int64_t)b * H + iy) * W * 3 + el;
        if (XF32) {
          const float4 v = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(x) + g);
          lo = pk_bf16(v.x, v.y); hi = pk_bf16(v.z, v.w);
        } else {
          const uint2 v = *reinterpret_cast<const uint2 *>(reinterpret_cast<const bf16_t *>(x) + g);
          lo = v.x; hi = v.y;
        }
      }
      bf16_t *d = patch + ky * PITCH + 1 + 4 * m;                  // odd element: 2 + 4 + 2 bytes
      d[0] = (bf16_t)(lo & 0xffffu);
      *reinterpret_cast<unsigned *>(d + 1) = (lo >> 16) | (hi << 16);
      d[3] = (bf16_t)(hi >> 16);
    }
    return;
  }
  for (int idx = t; idx < 7 * PROW; idx += nth) {
    const int ky = idx / PROW, e = idx - ky * PROW;
    const int iy = 2 * oy - 3 + ky, col = e / 3, ix = ix0 + col;
    bf16_t v = 0;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
      const int64_t g = (((int64_t)b * H + iy) * W + ix0) * 3 + e;
      v = XF32 ? f2bf(reinterpret_cast<const float *>(x)[g]) : reinterpret_cast<const bf16_t *>(x)[g];
    }
    patch[ky * PITCH + 1 + e] = v;
  }
}

// the fast path split into its two halves, so that a tile's pieces can be in flight while the previous tile is multiplied: every thread
// owns PF_N pieces (7 rows x 198 pieces of 4 elements / 256 threads)
constexpr int PF_NCH = (PROW + 6) / 4, PF_N = (7 * PF_NCH + 255) / 256;
// (Round 6: the pieces are kept RAW and loaded unconditionally — from a clamped address, dropped by value when stored.  Converted to bf16 inside
// `if (in range) { load; convert }` every piece was waited for on the spot, one after the other: the "flight" was six serial round trips in front
// of the multiplication it was meant to hide under, tools/debug/serial_loads.py.)
template <bool XF32> struct PatchRegs { float4 raw[PF_N]; unsigned ok; };
template <> struct PatchRegs<false> { uint2 raw[PF_N]; unsigned ok; };
__device__ __forceinline__ void load_piece(float4 &v, const void *x, int64_t g) { v = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(x) + g); }
__device__ __forceinline__ void load_piece(uint2 &v, const void *x, int64_t g) { v = *reinterpret_cast<const uint2 *>(reinterpret_cast<const bf16_t *>(x) + g); }
template <bool XF32>
__device__ __forceinline__ void patch_load(PatchRegs<XF32> &r, const void *x, int b, int oy, int ox0, int H, int W, int t)
{
  const int ix0 = 2 * ox0 - 4;
  r.ok = 0u;
#pragma unroll
  for (int i = 0; i < PF_N; ++i) {
    const int idx = t + 256 * i, ky = idx / PF_NCH, m = idx - ky * PF_NCH;
    const int iy = 2 * oy - 3 + ky, el = ix0 * 3 + 4 * m;
    const bool in = idx < 7 * PF_NCH && iy >= 0 && iy < H && el >= 0 && el < 3 * W;
    const int64_t g = in ? ((int64_t)b * H + iy) * W * 3 + el : 0;
    r.ok |= (in ? 1u : 0u) << i;
    load_piece(r.raw[i], x, g);
  }
}
__device__ __forceinline__ void piece_bits(const float4 &v, unsigned &lo, unsigned &hi) { lo = pk_bf16(v.x, v.y); hi = pk_bf16(v.z, v.w); }
__device__ __forceinline__ void piece_bits(const uint2 &v, unsigned &lo, unsigned &hi) { lo = v.x; hi = v.y; }
template <bool XF32>
__device__ __forceinline__ void patch_store(bf16_t *patch, const PatchRegs<XF32> &r, int t)
{
#pragma unroll
  for (int i = 0; i < PF_N; ++i) {
    const int idx = t + 256 * i, ky = idx / PF_NCH, m = idx - ky * PF_NCH;
    if (idx < 7 * PF_NCH) {
      unsigned lo, hi;
      piece_bits(r.raw[i], lo, hi);
      if (!((r.ok >> i) & 1u)) { lo = 0u; hi = 0u; }
      bf16_t *d = patch + ky * PITCH + 1 + 4 * m;
      d[0] = (bf16_t)(lo & 0xffffu);
      *reinterpret_cast<unsigned *>(d + 1) = (lo >> 16) | (hi << 16);
      d[3] = (bf16_t)(hi >> 16);
    }
  }
}

template <bool XF32>
__global__ __launch_bounds__(256) void stem_fwd(const void *__restrict__ x, const bf16_t *__restrict__ w, const float *__restrict__ scale,
                                                const float *__restrict__ bias, bf16_t *__restrict__ y, int B, int H, int W, int Ho, int Wo,
                                                int tiles_x, int ntiles, int relu, int fast)
{
  constexpr int OTP = 68;                                                  // output tile pitch (elements): 136 bytes
  __shared__ __attribute__((aligned(16))) bf16_t sm[TP * OTP + 7 * PITCH];  // output tile | patch; first the padded filter [64][176]
  __shared__ __attribute__((aligned(16))) float sb[128];                   // scale[64], bias[64]
  bf16_t *ot = sm, *patch = sm + TP * OTP;
  static_assert(64 * 176 <= TP * OTP + 7 * PITCH, "the padded filter passes through the tile buffers");
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fr = lane & 31, fh = lane >> 5;
  if (t < 64) { sb[t] = scale[t]; sb[64 + t] = bias[t]; }
  // the filter -> LDS as [channel][7 rows x 24] (coalesced 2-byte loads, zeros in the pads), then this lane's share of it as MFMA operands:
  // channel fr + 32 tile, k' = 8 (2 s + fh) .. + 7 of step s — one 16-byte LDS read each (per-lane 2-byte global loads at a 294-byte lane
  // stride cost the prologue 700 uncoalesced wave loads per workgroup)
  for (int i = t; i < 64 * 176 / 2; i += 256) reinterpret_cast<unsigned *>(sm)[i] = 0u;
  __syncthreads();
  for (int i = t; i < 64 * 147; i += 256) {
    const int ch = i / 147, r = i - ch * 147, ky = r / 21, j = r - ky * 21;
    sm[ch * 176 + ky * KROW + j] = w[i];
  }
  __syncthreads();
  hwbf16x8 wf[11][2];
#pragma unroll
  for (int s = 0; s < 11; ++s)
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) wf[s][tl] = *reinterpret_cast<const hwbf16x8 *>(sm + (fr + 32 * tl) * 176 + 8 * (2 * s + fh));
  __syncthreads();
  for (int i = t; i < 7 * PITCH; i += 256) patch[i] = 0;                    // (the pad beyond a row's elements stays zero)
  __syncthreads();
  PatchRegs<XF32> pr;
  if (fast && (int)blockIdx.x < ntiles) {
    const int tile = blockIdx.x, tx = tile % tiles_x, r = tile / tiles_x;
    patch_load<XF32>(pr, x, r / Ho, r % Ho, tx * TP, H, W, t);
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tx = tile % tiles_x, r = tile / tiles_x, oy = r % Ho, b = r / Ho, ox0 = tx * TP;
    if (fast) {
      patch_store(patch, pr, t);
      const int nx = tile + gridDim.x;
      if (nx < ntiles) {                                            // the next tile's pieces fly while this one is multiplied and stored
        const int ntx = nx % tiles_x, nr = nx / tiles_x;
        patch_load<XF32>(pr, x, nr / Ho, nr % Ho, ntx * TP, H, W, t);
      }
    } else {
      stage_patch<XF32>(patch, x, b, oy, ox0, H, W, t, 256, false);
    }
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tl][e] = 0.f;
    const int pix = wave * 32 + fr;
#pragma unroll
    for (int s = 0; s < 11; ++s) {
      const int g = 2 * s + fh, ky = min(g / 3, 6), j0 = (g % 3) * 8;
      const unsigned *p = reinterpret_cast<const unsigned *>(patch + ky * PITCH + PBASE + 6 * pix + j0);     // (an even element index: 4-byte aligned)
      const hwbf16x8 xf = __builtin_bit_cast(hwbf16x8, u32x4{p[0], p[1], p[2], p[3]});
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s][0], xf, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s][1], xf, acc[1], 0, 0, 0);
    }
    // C^T: lane = pixel, register e = channel (e & 3) + 8 (e >> 2) + 4 fh of the 32-channel tile: quads of 4 consecutive channels.  The
    // tile's output — 128 pixels x 128 bytes — is ONE contiguous 16 KB run of the NHWC tensor: it goes through LDS (8-byte pieces in, pitch
    // 136 bytes against bank conflicts) and leaves as 16 bytes per lane, 1 KB per wavefront instruction (direct 8-byte stores at a
    // 128-byte lane stride were this kernel's bound: eight partial writes per line)
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = tl * 32 + 8 * q + 4 * fh;
        const float4 sc = *reinterpret_cast<const float4 *>(sb + ch), bi = *reinterpret_cast<const float4 *>(sb + 64 + ch);
        float v0 = acc[tl][4 * q] * sc.x + bi.x, v1 = acc[tl][4 * q + 1] * sc.y + bi.y, v2 = acc[tl][4 * q + 2] * sc.z + bi.z,
              v3 = acc[tl][4 * q + 3] * sc.w + bi.w;
        if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
        *reinterpret_cast<uint2 *>(ot + pix * OTP + ch) = make_uint2(pk_bf16(v0, v1), pk_bf16(v2, v3));
      }
    __syncthreads();
    {
      bf16_t *yrow = y + (((int64_t)b * Ho + oy) * Wo + ox0) * 64;
      const int npix = min(TP, Wo - ox0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = t + 256 * i, p = idx >> 3, c8 = (idx & 7) * 8;      // 16-byte piece idx of the tile
        if (p < npix) {
          const uint2 lo = *reinterpret_cast<const uint2 *>(ot + p * OTP + c8), hi = *reinterpret_cast<const uint2 *>(ot + p * OTP + c8 + 4);
          *reinterpret_cast<uint4 *>(yrow + p * 64 + c8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
      }
    }
    __syncthreads();
  }
}

// dW partial sums.  The contraction runs over PIXELS, along which neither operand is contiguous: both are laid out in LDS as they are
// produced — SY[pixel][64 channels] (the masked gradient) and SX[pixel][k'] (the tile's im2col rows: 7 x 24 contiguous patch elements per
// pixel, copied 48 bytes at a time) — and transposed on the way out by ds_read_b64_tr_b16 (the scheme of conv_bf16.hip / gemm_x3.hip; lane
// map: profiles/r02_tr_read_probe.txt): lane s of a 16-lane group points at row s / 4, columns 4 (s % 4) .. + 3 of a [4 pixels][16 columns]
// block and receives column s of those 4 pixels; two reads (pixels +0 and +4) are one 8-pixel MFMA operand.  Row pitches 96 / 224
// elements (= 16 banks mod 32).  Workgroup = 4 wavefronts; wavefront w owns output channels 32 (w & 1) .. + 31 and the k' tiles
// 3 (w >> 1) .. + 2 (6 tiles of 32 cover k' < 192 >= 168): 3 accumulators, kept over all the workgroup's tiles.
constexpr int SYP = 96, SXP = 224, SUB = 64;   // pitches (elements); pixels per im2col sub-tile
typedef short v4s16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ hwbf16x8 frag_tr(const bf16_t *p, int pitch)
{
  typedef __attribute__((address_space(3))) v4s16 *lp;
  union { v4s16 h[2]; hwbf16x8 v; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p));
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p + 4 * pitch));
  return u.v;
}

template <bool XF32>
__global__ __launch_bounds__(256) void stem_wgrad(const void *__restrict__ x, const bf16_t *__restrict__ gy, const bf16_t *__restrict__ yact,
                                                  float *__restrict__ ws, int B, int H, int W, int Ho, int Wo, int tiles_x, int ntiles, int fast)
{
  __shared__ __attribute__((aligned(16))) bf16_t patch[7 * PITCH];
  __shared__ __attribute__((aligned(16))) bf16_t SY[TP * SYP];
  __shared__ __attribute__((aligned(16))) bf16_t SX[SUB * SXP];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < 7 * PITCH; i += 256) patch[i] = 0;
  for (int i = t; i < SUB * SXP / 2; i += 256) reinterpret_cast<unsigned *>(SX)[i] = 0u;       // (columns 168 .. 223 are never written: finite zeros)
  const int mt = wave & 1, nt0 = 3 * (wave >> 1);
  f32x16 acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  const int grp = lane >> 4, sl = lane & 15;
  const int frow = 8 * (grp >> 1) + (sl >> 2), fcol = 16 * (grp & 1) + 4 * (sl & 3);
  __syncthreads();
  // a thread's share of a tile's masked gradient: 4 pieces of 8 channels (pixels t / 8 + 32 i).  Loaded RAW and unconditionally (clamped pixel), masked
  // when stored to LDS: with the ReLU mask formed right behind each conditional pair of loads, the four pairs were four serial round trips
  struct GRegs { uint4 g[4], a[4]; };
  auto g_load = [&](GRegs &gr, int b, int oy, int ox0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = min((t >> 3) + 32 * i, Wo - 1 - ox0), c8 = (t & 7) * 8;
      const int64_t o = (((int64_t)b * Ho + oy) * Wo + ox0 + p) * 64 + c8;
      gr.g[i] = *reinterpret_cast<const uint4 *>(gy + o);
      gr.a[i] = yact ? *reinterpret_cast<const uint4 *>(yact + o) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto g_store = [&](const GRegs &gr, int ox0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = (t >> 3) + 32 * i;
      uint4 g = gr.g[i];
      if (yact) {
        const uint4 a = gr.a[i];
        // y is relu(.) rounded to bf16: positive <=> a non-zero, non-negative pattern
        auto m = [](unsigned gv, unsigned av) {
          const unsigned lo = ((av & 0xffffu) != 0u && !(av & 0x8000u)) ? 0xffffu : 0u, hi = ((av >> 16) != 0u && !(av & 0x80000000u)) ? 0xffff0000u : 0u;
          return gv & (lo | hi);
        };
        g = make_uint4(m(g.x, a.x), m(g.y, a.y), m(g.z, a.z), m(g.w, a.w));
      }
      if (ox0 + p >= Wo) g = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4 *>(SY + p * SYP + (t & 7) * 8) = g;
    }
  };
  PatchRegs<XF32> pr;
  GRegs gr;
  if ((int)blockIdx.x < ntiles) {
    const int tile = blockIdx.x, tx = tile % tiles_x, r = tile / tiles_x;
    if (fast) patch_load<XF32>(pr, x, r / Ho, r % Ho, tx * TP, H, W, t);
    g_load(gr, r / Ho, r % Ho, tx * TP);
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tx = tile % tiles_x, r = tile / tiles_x, oy = r % Ho, b = r / Ho, ox0 = tx * TP;
    if (fast) patch_store(patch, pr, t);
    else stage_patch<XF32>(patch, x, b, oy, ox0, H, W, t, 256, false);
    g_store(gr, ox0);
    {
      const int nx = tile + gridDim.x;
      if (nx < ntiles) {                                            // the next tile's operands fly while this one is multiplied
        const int ntx = nx % tiles_x, nr = nx / tiles_x;
        if (fast) patch_load<XF32>(pr, x, nr / Ho, nr % Ho, ntx * TP, H, W, t);
        g_load(gr, nr / Ho, nr % Ho, ntx * TP);
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int sub = 0; sub < TP / SUB; ++sub) {
      // im2col rows of pixels [SUB sub, + SUB): 7 x 24 elements = 7 x 48 bytes per pixel, copied from the patch (12 dwords -> 3 x 16 bytes)
      for (int idx = t; idx < SUB * 7; idx += 256) {
        const int p = idx / 7, ky = idx - p * 7;
        const unsigned *src = reinterpret_cast<const unsigned *>(patch + ky * PITCH + PBASE + 6 * (sub * SUB + p));
        uint4 *dst = reinterpret_cast<uint4 *>(SX + p * SXP + ky * KROW);
        dst[0] = make_uint4(src[0], src[1], src[2], src[3]);
        dst[1] = make_uint4(src[4], src[5], src[6], src[7]);
        dst[2] = make_uint4(src[8], src[9], src[10], src[11]);
      }
      __syncthreads();
#pragma unroll
      for (int st = 0; st < SUB / 16; ++st) {
        const hwbf16x8 af = frag_tr(SY + (sub * SUB + st * 16 + frow) * SYP + mt * 32 + fcol, SYP);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const hwbf16x8 bf = frag_tr(SX + (st * 16 + frow) * SXP + (nt0 + j) * 32 + fcol, SXP);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[j], 0, 0, 0);
        }
      }
      __syncthreads();
    }
  }
  // C: lane = k' row (B operand row), register e = channel (e & 3) + 8 (e >> 2) + 4 (lane / 32) of this wavefront's 32
  const int fr = lane & 31, fh = lane >> 5;
  float *o = ws + (int64_t)blockIdx.x * 64 * 192;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ch = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
      o[ch * 192 + (nt0 + j) * 32 + fr] = acc[j][e];
    }
}

// dw[o][ky][kx][c] = scale[o] * sum over workgroups of ws[wg][o][24 ky + 3 kx + c]: one block per output channel, thread (k', part) sums
// a quarter of the workgroups with 8 loads in flight, the quarters meet in LDS in a fixed order (deterministic)
__global__ __launch_bounds__(768) void stem_wgrad_reduce(const float *__restrict__ ws, const float *__restrict__ scale, void *__restrict__ dw,
                                                         int nwg, int dw_is_f32)
{
  __shared__ float part[4][192];
  const int ch = blockIdx.x, kp = threadIdx.x % 192, pt = threadIdx.x / 192;
  const int g0 = (int)((int64_t)nwg * pt / 4), g1 = (int)((int64_t)nwg * (pt + 1) / 4);
  const float *p = ws + ch * 192 + kp;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int g = g0;
  for (; g + 8 <= g1; g += 8) {
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] += p[(int64_t)(g + i) * 64 * 192];
  }
  for (; g < g1; ++g) s[0] += p[(int64_t)g * 64 * 192];
  part[pt][kp] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  const int ky = kp / KROW, j = kp - ky * KROW;
  if (pt != 0 || ky >= 7 || j >= 21) return;
  const float v = ((part[0][kp] + part[1][kp]) + (part[2][kp] + part[3][kp])) * scale[ch];
  const int o = ch * 147 + ky * 21 + j;
  if (dw_is_f32) reinterpret_cast<float *>(dw)[o] = v;
  else reinterpret_cast<bf16_t *>(dw)[o] = f2bf(v);
}

}  // namespace

extern "C" int pd_stem7x7_fwd(const void *x, int x_is_f32, const void *w_bf16, const float *scale, const float *bias, void *y_bf16, int B, int H, int W,
                              int relu, void *stream_)
{
  if (B < 0 || H <= 0 || W <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_stem7x7_fwd: B=%d H=%d W=%d", B, H, W);
  if (B == 0) return PD_OK;
  if (!x || !w_bf16 || !scale || !bias || !y_bf16 || ((uintptr_t)y_bf16 & 7)) return pd_set_error(PD_ERR_INVALID_ARG, "pd_stem7x7_fwd: null / misaligned pointer");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, tiles_x = (Wo + TP - 1) / TP;
  const int64_t nt = (int64_t)B * Ho * tiles_x;
  if (nt > 0x7fffffffLL || (int64_t)B * H * W * 3 > 0x7fffffffffLL) return pd_set_error(PD_ERR_INVALID_ARG, "pd_stem7x7_fwd: too many pixels");
  const unsigned grid = (unsigned)(nt < 768 ? nt : 768);
  hipStream_t st = (hipStream_t)stream_;
  const int fast = (W % 4 == 0) && !((uintptr_t)x & 15);            // 4-element pieces of an image row are aligned and never straddle its end
  if (x_is_f32)
    hipLaunchKernelGGL(stem_fwd<true>, dim3(grid), dim3(256), 0, st, x, (const bf16_t *)w_bf16, scale, bias, (bf16_t *)y_bf16, B, H, W, Ho, Wo, tiles_x, (int)nt, relu, fast);
  else
    hipLaunchKernelGGL(stem_fwd<false>, dim3(grid), dim3(256), 0, st, x, (const bf16_t *)w_bf16, scale, bias, (bf16_t *)y_bf16, B, H, W, Ho, Wo, tiles_x, (int)nt, relu, fast);
  return pd_check_launch("pd_stem7x7_fwd");
}

extern "C" int64_t pd_stem_wgrad_workspace_floats(void) { return (int64_t)WG_WGRAD * 64 * 192; }

extern "C" int pd_stem7x7_wgrad(const void *x, int x_is_f32, const void *gy_bf16, const void *y_bf16, const float *scale, void *dw, int dw_is_f32,
                                float *workspace, int B, int H, int W, int relu, void *stream_)
{
  if (B < 0 || H <= 0 || W <= 0) return pd_set_error(PD_ERR_INVALID_ARG, "pd_stem7x7_wgrad: B=%d H=%d W=%d", B, H, W);
  if (!x || !gy_bf16 || !scale || !dw || !workspace || (relu && !y_bf16) || ((uintptr_t)gy_bf16 & 15) || ((uintptr_t)y_bf16 & 15))
    return pd_set_error(PD_ERR_INVALID_ARG, "pd_stem7x7_wgrad: null / misaligned pointer");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, tiles_x = (Wo + TP - 1) / TP;
  const int64_t nt = (int64_t)B * Ho * tiles_x;
  if (nt > 0x7fffffffLL) return pd_set_error(PD_ERR_INVALID_ARG, "pd_stem7x7_wgrad: too many pixels");
  hipStream_t st = (hipStream_t)stream_;
  const int nwg = (int)(nt < WG_WGRAD ? (nt > 0 ? nt : 1) : WG_WGRAD);
  const void *ya = relu ? y_bf16 : nullptr;
  const int fast = (W % 4 == 0) && !((uintptr_t)x & 15);
  if (x_is_f32)
    hipLaunchKernelGGL(stem_wgrad<true>, dim3(nwg), dim3(256), 0, st, x, (const bf16_t *)gy_bf16, (const bf16_t *)ya, workspace, B, H, W, Ho, Wo, tiles_x, (int)nt, fast);
  else
    hipLaunchKernelGGL(stem_wgrad<false>, dim3(nwg), dim3(256), 0, st, x, (const bf16_t *)gy_bf16, (const bf16_t *)ya, workspace, B, H, W, Ho, Wo, tiles_x, (int)nt, fast);
  hipLaunchKernelGGL(stem_wgrad_reduce, dim3(64), dim3(768), 0, st, workspace, scale, dw, nt > 0 ? nwg : 0, dw_is_f32);
  return pd_check_launch("pd_stem7x7_wgrad");
}
