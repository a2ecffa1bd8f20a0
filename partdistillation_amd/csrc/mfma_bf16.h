// Shared device helpers for the bf16 matrix-core kernels (attention.hip, smallgemm.hip).
//
// v_mfma_f32_32x32x8_bf16_1k with X and Y row-major and each lane holding 4 consecutive contraction elements of row
// (lane % 32) [elements 4*(lane/32) .. +3 of the 8-wide step]:
//     mma(c, x, y):  c[i][j] += sum_k X[i][k] * Y[j][k]        (C = X . Y^T)
// and the result sits with lane = j (the Y row), register e = X row (e&3) + 8*(e>>2) + 4*(lane>>5)
// (checked on the hardware by tools/probes/mfma_bf16_probe.hip).
#ifndef PD_MFMA_BF16_H
#define PD_MFMA_BF16_H
#include <hip/hip_runtime.h>

namespace pdmfma {

typedef unsigned short bf16_t;
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 hwbf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// round-to-nearest-even fp32 -> bf16 pair: the compiler emits v_cvt_pk_bf16_f32 (gfx950) and, unlike inline asm, knows the
// wait states an MFMA that reads the result needs (an asm version fed stale operands to the next MFMA)
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi)
{
  const f32x2 x = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(x, hwbf16x2));
}
__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d)
{
  union { unsigned u[2]; bf16x4 v; } x;
  x.u[0] = pk_bf16(a, b); x.u[1] = pk_bf16(c, d);
  return x.v;
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ void mma(f32x16 &c, bf16x4 x, bf16x4 y) { c = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(x, y, c, 0, 0, 0); }
__device__ __forceinline__ bf16x4 lds4(const bf16_t *p) { return *reinterpret_cast<const bf16x4 *>(p); }
__device__ __forceinline__ bf16x4 gather4(const bf16_t *p, int stride)      // 4 elements `stride` apart
{
  bf16x4 r;
  r[0] = (short)p[0]; r[1] = (short)p[stride]; r[2] = (short)p[2 * stride]; r[3] = (short)p[3 * stride];
  return r;
}

}  // namespace pdmfma
#endif
