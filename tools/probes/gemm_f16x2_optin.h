// Kernels of the two-plane GEMM family that lost their same-box A/B against the product's choice: correct, tested
// (tools/probes/test_optin_kernels.py), slower or tools-only.  Not part of libpd_hip.so: csrc/gemm_f16x2.hip includes this file only when built
// with -DPD_PROBES (make -C partdistillation_amd/csrc probes -> libpd_hip_probes.so; load it through PD_LIB_PATH).
#ifndef PD_GEMM_F16X2_OPTIN_H
#define PD_GEMM_F16X2_OPTIN_H

// ---------------------------------------------------------------------------------------------------------------------------
// gemm_ra_f16x2_k256 (round 5): the K = 256 products (value / output / offset projections and their input gradients, the 1 x 1
// convolutions on 256 channels) with the A operand going global memory -> REGISTERS -> matrix cores, never through LDS.
// What bounds the tiled kernel above on these shapes is not bytes or products but the chain load -> split -> LDS store -> barrier ->
// fragment read that every 16-deep step of every workgroup walks in phase (ablation: 10.5 us fixed + 9.5 loads + 11 products of a 31 us
// launch whose traffic takes 11).  Here:
//   * a wavefront OWNS 32 rows of A for all of K and a block of NTL x 32 columns of C.  The v_mfma_f32_32x32x16_f16 operand of a lane
//     (row lane % 32, 8 consecutive k of block lane / 32) is read straight from memory: per 32-wide chunk of k a lane loads the 64
//     contiguous bytes [32 c + 16 (lane / 32), + 16) of its row (4 x global_load_dwordx4 issued back to back: the wavefront touches each
//     128-byte line of its 32 rows exactly once, as a whole), splits them into the two fp16 planes in registers and multiplies.  No A
//     tile in LDS, no barrier in the K loop, loads one chunk (4 KB per wavefront) ahead.  The contraction order inside a chunk is
//     (lane / 32, 8-k group): the weight fragments are read to match.
//   * the workgroup's NTL x 32 columns of the weight matrix are split ONCE, in the prologue, into a resident LDS image
//     [plane][k panel of 8][column][8 halves] (128 columns: 129 KB, panels padded by 16 bytes against the staging stores' bank
//     conflicts); a fragment is one ds_read_b128, 512 contiguous bytes per half-wave.
//   * 12 wavefronts per workgroup (3 per SIMD, <= 168 VGPRs): 384 rows x 128 columns; the column blocks of a row block are
//     neighbours in the XCD-chunked block order, so the second reader of the A rows finds them in that XCD's L2.
// Epilogue as the tiled kernel: row / column scales undone, bias, optional bf16 output, optional row maxima of C (atomic max).
constexpr int RA_PSTRIDE_PAD = 16;
template <int NTL, int NWAVE, bool AM, int RABL = 0>
__global__ __launch_bounds__(NWAVE * 64)
void gemm_ra_f16x2_k256(const float *__restrict__ A, const float *__restrict__ B, const float *__restrict__ bias, float *__restrict__ C,
                        int M, int N, int lda, int ldb, int ldc, int nblk_n, const float *__restrict__ a_amax,
                        const float *__restrict__ b_amax, unsigned *__restrict__ c_amax, int flags)
{
  constexpr int NT = NTL * 32, PST = NT * 16 + RA_PSTRIDE_PAD, PLANE = 32 * PST, NTH = NWAVE * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char ra_lds[];     // 2 planes, then NT inverse column scales, then NWAVE x 32 inverse row scales
  float *sib = reinterpret_cast<float *>(ra_lds + 2 * PLANE);
  float *sia = sib + NT;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fr = lane & 31, fh = lane >> 5;
  const int lb = xcd_chunk(blockIdx.x, gridDim.x);
  const int nb = lb % nblk_n, mb = lb / nblk_n;
  const int n0 = nb * NT, row0 = (mb * NWAVE + wave) * 32;
  // ---- this lane's row of A: scale, first chunk in flight before anything else
  const int arow = min(row0 + fr, M - 1);
  const float *ap = A + (int64_t)arow * lda + 16 * fh;
  constexpr int PF = 3;                                              // chunks of A in flight ahead of the one being multiplied
  // chunk order rotated per wavefront (the sum does not care): at any moment the workgroup's loads spread over all eight 128-byte
  // columns of the 1 KB rows instead of every wavefront asking for the same column — rows 1 KB apart map to few memory channels
  const int rot = __builtin_amdgcn_readfirstlane((wave + lb) & 7);
  float4 raw[PF + 1][4];
#pragma unroll
  for (int c = 0; c < PF; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) raw[c][i] = (RABL & 1) ? make_float4(1.f, 0.5f, (float)c, (float)i) : *reinterpret_cast<const float4 *>(ap + 32 * ((c + rot) & 7) + 4 * i);
  float sa = 1.f, ia = 1.f;
  if (AM) row_scale(a_amax[arow], sa, ia);
  if (lane < 32) sia[wave * 32 + fr] = ia;
  // ---- the weight block, split once into the resident image: all of a thread's loads first (one memory latency, not one per piece)
  {
    constexpr int NIT = (NT * 64 + NTH - 1) / NTH;
    float4 wv[NIT];
    float wsb[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = t + it * NTH, n = idx >> 6, k4 = idx & 63, gn = n0 + n;
      wv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      wsb[it] = 0.f;
      if (!(RABL & 16) && idx < NT * 64 && gn < N) {
        wv[it] = *reinterpret_cast<const float4 *>(B + (int64_t)gn * ldb + 4 * k4);
        if (AM) wsb[it] = b_amax[gn];
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = t + it * NTH, n = idx >> 6, k4 = idx & 63;
      if (!(RABL & 16) && idx < NT * 64) {
        float sb = 1.f, ibv = 1.f;
        if (AM && n0 + n < N) row_scale(wsb[it], sb, ibv);
        const SplitH x = split4h(wv[it], sb);
        unsigned char *p = ra_lds + (k4 >> 1) * PST + n * 16 + (k4 & 1) * 8;
        *reinterpret_cast<uint2 *>(p) = x.hi;
        *reinterpret_cast<uint2 *>(p + PLANE) = x.lo;
        if (k4 == 0) sib[n] = ibv;
      }
    }
  }
  __syncthreads();
  f32x16 acc[NTL];
#pragma unroll
  for (int j = 0; j < NTL; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  const unsigned char *bbase = ra_lds + fr * 16;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c + PF < 8) {
#pragma unroll
      for (int i = 0; i < 4; ++i) raw[(c + PF) % (PF + 1)][i] = (RABL & 1) ? make_float4(1.f, 0.5f, (float)c, (float)i) : *reinterpret_cast<const float4 *>(ap + 32 * ((c + PF + rot) & 7) + 4 * i);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const SplitH u = split4h(raw[c % (PF + 1)][2 * j], sa), v = split4h(raw[c % (PF + 1)][2 * j + 1], sa);
      const h16x8 ah = __builtin_bit_cast(h16x8, u32x4{u.hi.x, u.hi.y, v.hi.x, v.hi.y});
      const h16x8 al = __builtin_bit_cast(h16x8, u32x4{u.lo.x, u.lo.y, v.lo.x, v.lo.y});
      const unsigned char *bp = bbase + (4 * ((c + rot) & 7) + 2 * fh + j) * PST;
      h16x8 bh[NTL], bl[NTL];
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt) {
        bh[nt] = (RABL & 2) ? al : *reinterpret_cast<const h16x8 *>(bp + nt * 512);
        bl[nt] = (RABL & 2) ? ah : *reinterpret_cast<const h16x8 *>(bp + nt * 512 + PLANE);
      }
      // the three products of a term, each over the NTL independent accumulators (no back-to-back dependent matrix instructions)
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt) { if (RABL & 4) asm volatile("" ::"v"(al), "v"(bh[nt])); else mmah(acc[nt], al, bh[nt]); }
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt) { if (RABL & 4) asm volatile("" ::"v"(ah), "v"(bl[nt])); else mmah(acc[nt], ah, bl[nt]); }
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt) { if (!(RABL & 4)) mmah(acc[nt], ah, bh[nt]); }
    }
  }
  // ---- epilogue.  C of the 32 x 32 product: column lane % 32, rows (e & 3) + 8 (e >> 2) + 4 (lane / 32)
  float bv[NTL], ib[NTL];
#pragma unroll
  for (int nt = 0; nt < NTL; ++nt) {
    const int col = n0 + nt * 32 + fr;
    bv[nt] = (bias && col < N) ? bias[col] : 0.f;
    ib[nt] = sib[nt * 32 + fr];
  }
  const bool full = row0 + 32 <= M && n0 + NT <= N;
  float rmax[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int rl = (e & 3) + 8 * (e >> 2) + 4 * fh, row = row0 + rl;
    const float iar = sia[wave * 32 + rl];
    float rm = 0.f;
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) {
      const int col = n0 + nt * 32 + fr;
      const float v = acc[nt][e] * (iar * ib[nt]) + bv[nt];
      if (full || (row < M && col < N)) {
        if (RABL & 8) asm volatile("" ::"v"(v));
        else if (flags & 1) reinterpret_cast<unsigned short *>(C)[(int64_t)row * ldc + col] = f2bf_rne(v);
        else C[(int64_t)row * ldc + col] = v;
        rm = fmaxf(rm, fabsf(v));
      }
    }
    rmax[e] = rm;
  }
  if (c_amax) {
    // row maxima over this wavefront's NT columns: the halving butterfly of gemm_rows_f16x2_k256 (16 values -> 1 over the 32 lanes of a half)
#pragma unroll
    for (int n = 8, m = 16; n >= 1; n >>= 1, m >>= 1) {
      const bool up = (lane & m) != 0;
#pragma unroll
      for (int i = 0; i < n; ++i) {
        const float mine = up ? rmax[n + i] : rmax[i], send = up ? rmax[i] : rmax[n + i];
        rmax[i] = fmaxf(mine, __shfl_xor(send, m, 64));
      }
    }
    const float r = fmaxf(rmax[0], __shfl_xor(rmax[0], 1, 64));
    const int e = (lane >> 1) & 15, rl = (e & 3) + 8 * (e >> 2) + 4 * fh;
    if (!(lane & 1) && row0 + rl < M && r > 0.f) atomicMax(c_amax + row0 + rl, __float_as_uint(r));
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// gemm_kres_f16x2 (round 5): the deep-K products with 256 output columns (the encoder FFN's second Linear 256 <- 1024 and the input
// gradient of its first: 12 launches of ~95 us per step) with the K loop OUTSIDE and the accumulators of a whole row block resident.
// The tiled kernel gives these shapes 672 tiles of 128 x 128 on 256 CUs (2.6 rounds -> 3) and reads A twice (once per column tile).  Here
// one persistent workgroup per CU owns a block of 192 rows x 256 columns: 8 wavefronts = 2 row groups (96 rows: three 32-row MFMA
// tiles) x 4 column groups (64 columns), 6 accumulator tiles = 96 registers per lane — 192 x 256 fp32 results stay in registers for all
// of K.  Per 32-deep chunk the block's A rows (24 KB) and the weight panel's slice (32 KB, from L2) are read by all 512 threads, split and
// laid into double-buffered two-plane LDS images (the row stream's [k panel of 8][row][8 halves] layout: a fragment is one ds_read_b128);
// the loads run TWO chunks ahead in registers, one barrier per chunk.  Per 16-deep step and wavefront: 10 fragment reads, 18 matrix
// instructions.  A is read once, C written once.  EXPERIMENTAL, not the product's choice: see the dispatch in gemm_tn_f16x2_impl.  (First version: the weights went global -> registers per step, behind the chunk's A
// loads in the in-order load queue — every step waited for HBM: 98 us against the tiled kernel's 86.)
template <bool AM, int KABL = 0>     // KABL (tools only): 1 no split / LDS stores in the loop, 2 no products, 4 no fragment reads, 8 no global loads in the loop
__global__ __launch_bounds__(512)
void gemm_kres_f16x2(const float *__restrict__ A, const float *__restrict__ B, const float *__restrict__ bias, float *__restrict__ C, int M, int K,
                     int lda, int ldb, int ldc, int npanels, const float *__restrict__ a_amax, const float *__restrict__ b_amax)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char kr_lds[];
  unsigned char *const aimg = kr_lds, *const bimg = kr_lds + 2 * KR_ABUF;
  float *sinv = reinterpret_cast<float *>(kr_lds + 2 * (KR_ABUF + KR_BBUF));
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6), fr = lane & 31, fh = lane >> 5;
  const int rgp = w >> 2, cg = w & 3;                                  // row group (96 rows), column group (64 columns)
  const int nblk = (M + KR_RB - 1) / KR_RB, NC = K / KR_KC;
  const int arow = t >> 3, ac4 = t & 7;                                 // staging: 8 threads per row (32 k = 8 float4); A rows arow + 64 j (3), B rows arow + 64 j (4)
  for (int work = blockIdx.x; work < nblk * npanels; work += gridDim.x) {
    const int blk = work / npanels, panel = work - blk * npanels, row0 = blk * KR_RB, c0 = panel * 256, n0 = c0 + cg * 64;
    float sa[3], sb[4];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float inv = 1.f;
      sa[j] = 1.f;
      if (AM) row_scale(a_amax[min(row0 + arow + 64 * j, M - 1)], sa[j], inv);
      if (ac4 == 0) sinv[arow + 64 * j] = inv;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float inv = 1.f;
      sb[j] = 1.f;
      if (AM) row_scale(b_amax[c0 + arow + 64 * j], sb[j], inv);
    }
    float ibv[2] = {1.f, 1.f};
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      float sc = 1.f;
      if (AM) row_scale(b_amax[n0 + 32 * cb + fr], sc, ibv[cb]);
    }
    float4 RA[2][3], RB[2][4];
    auto gload = [&](int rs, int kc) {
      kc = min(kc, NC - 1);
#pragma unroll
      for (int j = 0; j < 3; ++j)
        RA[rs][j] = *reinterpret_cast<const float4 *>(A + (int64_t)min(row0 + arow + 64 * j, M - 1) * lda + kc * KR_KC + 4 * ac4);
#pragma unroll
      for (int j = 0; j < 4; ++j) RB[rs][j] = *reinterpret_cast<const float4 *>(B + (int64_t)(c0 + arow + 64 * j) * ldb + kc * KR_KC + 4 * ac4);
    };
    auto split_store = [&](int rs, int buf) {
      unsigned char *ab = aimg + buf * KR_ABUF + (ac4 >> 1) * KR_APANEL + (ac4 & 1) * 8;
      unsigned char *bb = bimg + buf * KR_BBUF + (ac4 >> 1) * KR_BPANEL + (ac4 & 1) * 8;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const SplitH v = split4h_u(RA[rs][j], sa[j]);
        unsigned char *p = ab + (arow + 64 * j) * 16;
        *reinterpret_cast<uint2 *>(p) = v.hi;
        *reinterpret_cast<uint2 *>(p + KR_APLANE) = v.lo;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const SplitH v = split4h_u(RB[rs][j], sb[j]);
        unsigned char *p = bb + (arow + 64 * j) * 16;
        *reinterpret_cast<uint2 *>(p) = v.hi;
        *reinterpret_cast<uint2 *>(p + KR_BPLANE) = v.lo;
      }
    };
    f32x16 acc[3][2];
#pragma unroll
    for (int ti = 0; ti < 3; ++ti)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ti][cb][e] = 0.f;
    gload(0, 0);
    gload(1, 1);
    split_store(0, 0);
    gload(0, 2);
    __syncthreads();
    // one staged item: A rows (j < 3) or weight rows (j - 3) of register stage rs -> the images `buf`
    auto stage_item = [&](int rs, int buf, int j) {
      if (j < 3) {
        const SplitH v = split4h_u(RA[rs][j], sa[j]);
        unsigned char *p = aimg + buf * KR_ABUF + (ac4 >> 1) * KR_APANEL + (ac4 & 1) * 8 + (arow + 64 * j) * 16;
        *reinterpret_cast<uint2 *>(p) = v.hi;
        *reinterpret_cast<uint2 *>(p + KR_APLANE) = v.lo;
      } else {
        const SplitH v = split4h_u(RB[rs][j - 3], sb[j - 3]);
        unsigned char *p = bimg + buf * KR_BBUF + (ac4 >> 1) * KR_BPANEL + (ac4 & 1) * 8 + (arow + 64 * (j - 3)) * 16;
        *reinterpret_cast<uint2 *>(p) = v.hi;
        *reinterpret_cast<uint2 *>(p + KR_BPLANE) = v.lo;
      }
    };
    auto load_item = [&](int rs, int kc, int j) {
      kc = min(kc, NC - 1);
      if (j < 3) RA[rs][j] = *reinterpret_cast<const float4 *>(A + (int64_t)min(row0 + arow + 64 * j, M - 1) * lda + kc * KR_KC + 4 * ac4);
      else RB[rs][j - 3] = *reinterpret_cast<const float4 *>(B + (int64_t)(c0 + arow + 64 * (j - 3)) * ldb + kc * KR_KC + 4 * ac4);
    };
    // The order below is the order of the instruction stream (sched_barrier between the pieces): left to itself the compiler emits the split's
    // ~100 VALU instructions as one block before or after the 36 matrix instructions, and eight wavefronts in step then run staging, fragment
    // reads and products one after the other (tools/debug/kres_ablate.py).  Step 0's products carry the split and the LDS stores of the next
    // chunk (register stage par ^ 1 -> images par ^ 1), step 1's the loads of chunk kc + 3 into the freed registers.
    auto chunk = [&](int kc, int par) {
      const unsigned char *ab = aimg + par * KR_ABUF + fh * KR_APANEL + (96 * rgp + fr) * 16;
      const unsigned char *bb = bimg + par * KR_BBUF + fh * KR_BPANEL + (64 * cg + fr) * 16;
#pragma unroll
      for (int s = 0; s < KR_KC / 16; ++s) {
        h16x8 ah[3], al[3], bh[2], bl[2];
#pragma unroll
        for (int ti = 0; ti < 3; ++ti) {
          if (KABL & 4) {
            ah[ti] = __builtin_bit_cast(h16x8, u32x4{(unsigned)s, (unsigned)ti, 1u, 2u});
            al[ti] = ah[ti];
          } else {
            ah[ti] = *reinterpret_cast<const h16x8 *>(ab + 2 * s * KR_APANEL + ti * 32 * 16);
            al[ti] = *reinterpret_cast<const h16x8 *>(ab + 2 * s * KR_APANEL + ti * 32 * 16 + KR_APLANE);
          }
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          if (KABL & 4) {
            bh[cb] = __builtin_bit_cast(h16x8, u32x4{(unsigned)s, (unsigned)cb, 3u, 4u});
            bl[cb] = bh[cb];
          } else {
            bh[cb] = *reinterpret_cast<const h16x8 *>(bb + 2 * s * KR_BPANEL + cb * 32 * 16);
            bl[cb] = *reinterpret_cast<const h16x8 *>(bb + 2 * s * KR_BPANEL + cb * 32 * 16 + KR_BPLANE);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 18; ++i) {
          const int pr = i / 6, ti = (i % 6) >> 1, cb = i & 1;
          if (KABL & 2) acc[ti][cb][0] += (float)ah[ti][0] * (float)bl[cb][1] + (float)al[ti][2] * (float)bh[cb][3];
          else if (pr == 0) mmah(acc[ti][cb], al[ti], bh[cb]);
          else if (pr == 1) mmah(acc[ti][cb], ah[ti], bl[cb]);
          else mmah(acc[ti][cb], ah[ti], bh[cb]);
          if (s == 0 && !(KABL & 1) && (i % 5) == 1) stage_item(par ^ 1, par ^ 1, i / 5);            // items 0 .. 3 after products 1, 6, 11, 16
          if (s == 0 && !(KABL & 1) && (i == 3 || i == 8 || i == 13)) stage_item(par ^ 1, par ^ 1, 4 + (i - 3) / 5);   // items 4 .. 6
          if (s == 1 && !(KABL & 8) && (i & 1) == 0 && i < 14) load_item(par ^ 1, kc + 3, i >> 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
    };
    for (int kc = 0; kc < NC; kc += 2) {
      chunk(kc, 0);
      if (kc + 1 < NC) chunk(kc + 1, 1);
    }
    // C rows of a 32 x 32 accumulator: (e & 3) + 8 (e >> 2) + 4 fh, column fr
#pragma unroll
    for (int ti = 0; ti < 3; ++ti) {
      const int rl0 = 96 * rgp + 32 * ti;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int col = n0 + 32 * cb + fr;
        const float bv = bias ? bias[col] : 0.f;
        float *cp = C + (int64_t)(row0 + rl0) * ldc + col;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int rl = (e & 3) + 8 * (e >> 2) + 4 * fh;
          if (row0 + rl0 + rl < M) cp[(int64_t)rl * ldc] = acc[ti][cb][e] * (sinv[rl0 + rl] * ibv[cb]) + bv;
        }
      }
    }
    __syncthreads();                                                     // sinv and the images are rewritten by the next block
  }
}

// W [N, K] fp32 -> two fp16 planes [2][N][K] with the rows scaled by row_scale(amax[row]) (the B operand of gemm_kpc_f16x2<.., BP>)
__global__ __launch_bounds__(256) void split_planes_f16x2(const float *__restrict__ Wm, int ldw, const float *__restrict__ amax, unsigned short *__restrict__ planes,
                                                          int N, int K)
{
  const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x, k4 = K / 4;
  if (i4 >= (int64_t)N * k4) return;
  const int row = (int)(i4 / k4), c = (int)(i4 - (int64_t)row * k4) * 4;
  float sc = 1.f, inv = 1.f;
  if (amax) row_scale(amax[row], sc, inv);
  const SplitH v = split4h_u(*reinterpret_cast<const float4 *>(Wm + (int64_t)row * ldw + c), sc);
  *reinterpret_cast<uint2 *>(planes + (int64_t)row * K + c) = v.hi;
  *reinterpret_cast<uint2 *>(planes + ((int64_t)N + row) * K + c) = v.lo;
}

#endif
