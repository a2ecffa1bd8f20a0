// Device helpers of the fp16 two-plane form of an fp32 product (gemm_f16x2.hip, the weight-gradient kernels of gemm_x3.hip):
//     x' = 2^s x,  hi = fp16(x'),  lo = fp16(x' - hi);   a b ~= hi_a hi_b + hi_a lo_b + lo_a hi_b   (fp32 accumulate)
#ifndef PD_F16X2_H
#define PD_F16X2_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pdh2 {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short h16_t;
// scale / inverse scale of a row from its absolute maximum: biased exponent e of the maximum, clamped to [20, 250];
// scale 2^(141 - e) puts the maximum into [2^14, 2^15); zero / tiny rows get the largest scale (harmless), inf / nan propagate
__device__ __forceinline__ void row_scale(float amax, float &s, float &inv)
{
  int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);
  e = e < 20 ? 20 : (e > 250 ? 250 : e);
  s = __uint_as_float((unsigned)(268 - e) << 23);
  inv = __uint_as_float((unsigned)(e - 14) << 23);
}

struct SplitH { uint2 hi, lo; };                                   // 4 consecutive k of one row, per plane
__device__ __forceinline__ void split2h(float x0, float x1, unsigned &h, unsigned &l)
{
  const f32x2 x = {x0, x1};
  const h16x2 hh = __builtin_convertvector(x, h16x2);              // round to nearest even
  const f32x2 r = {x0 - (float)hh[0], x1 - (float)hh[1]};         // exact
  const h16x2 ll = __builtin_convertvector(r, h16x2);
  h = __builtin_bit_cast(unsigned, hh);
  l = __builtin_bit_cast(unsigned, ll);
}
__device__ __forceinline__ SplitH split4h_p(float4 v, float s)
{
  SplitH o;
  split2h(v.x * s, v.y * s, o.hi.x, o.lo.x);
  split2h(v.z * s, v.w * s, o.hi.y, o.lo.y);
  return o;
}
// split4h with the scale multiply and the residual subtraction as ONE-LANE-ONE-VALUE instructions (v_mul_f32 / v_sub_f32).  The compiler turns
// the form above into v_pk_mul_f32 / v_pk_fma_f32 — half the instructions, which is what a wavefront that interleaves its own split with its
// own matrix instructions wants (the tiled kernel: 92 vs 104 us on 256 <- 1024) — but the packed fp32 forms contend with the matrix pipe: a
// wavefront that ONLY splits makes no progress while other wavefronts of its SIMD multiply (stamps in gemm_kpc_f16x2, tools/debug/kpc_trace.py:
// a half chunk's split + stores 2 550 cycles next to running matrix instructions, 1 050 with this form).
__device__ __forceinline__ float np_mul(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float np_sub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ void split2h_u(float x0, float x1, unsigned &h, unsigned &l)
{
  const f32x2 x = {x0, x1};
  const h16x2 hh = __builtin_convertvector(x, h16x2);
  const f32x2 r = {np_sub(x0, (float)hh[0]), np_sub(x1, (float)hh[1])};
  const h16x2 ll = __builtin_convertvector(r, h16x2);
  h = __builtin_bit_cast(unsigned, hh);
  l = __builtin_bit_cast(unsigned, ll);
}
__device__ __forceinline__ SplitH split4h_u(float4 v, float s)
{
  SplitH o;
  split2h_u(np_mul(v.x, s), np_mul(v.y, s), o.hi.x, o.lo.x);
  split2h_u(np_mul(v.z, s), np_mul(v.w, s), o.hi.y, o.lo.y);
  return o;
}
// split4h: the packed form unless a translation unit is built with -DPD_F16X2_UNPACKED_SPLIT (A/B builds)
__device__ __forceinline__ SplitH split4h(float4 v, float s)
{
#ifdef PD_F16X2_UNPACKED_SPLIT
  return split4h_u(v, s);
#else
  return split4h_p(v, s);
#endif
}
__device__ __forceinline__ void mmah(f32x16 &c, h16x8 x, h16x8 y) { c = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0); }

}  // namespace pdh2
#endif
