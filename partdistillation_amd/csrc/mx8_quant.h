// Device helpers of the MX-fp8 operand format (include/pd_mx8.h): the shared exponent of a 32-element block and the conversion of
// scaled values to fp8, for the kernels that emit such operands (csrc/mx8.hip: the standalone pass and the GEMM epilogue;
// csrc/swin_rows.hip: the LayerNorm rows).
#ifndef PD_MX8_QUANT_H
#define PD_MX8_QUANT_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pd_mx8.h"

namespace pdmx {
// shared exponent of a block with absolute maximum `amax`: the smallest X with amax 2^-X <= FMAX = 1.75 * 2^EMAX, clamped to
// [-126, 126] (both 2^X and 2^-X are normal floats).  Returns the E8M0 byte; mult = 2^-X.
template <int FMT>
__device__ __forceinline__ unsigned mx_exponent(float amax, float &mult)
{
  constexpr int EMAX = FMT == PD_MX8_E4M3 ? 8 : 15;
  const unsigned bits = __float_as_uint(amax);
  int X = (int)((bits >> 23) & 255u) - 127 - EMAX + ((bits & 0x7fffffu) > 0x600000u ? 1 : 0);
  X = min(max(X, -126), 126);
  mult = __uint_as_float((unsigned)(127 - X) << 23);
  return (unsigned)(X + 127);
}

// four values -> four fp8 bytes (|v mult| <= the format maximum by the choice of X, an exact power-of-two product: no clamp; a NaN stays one)
template <int FMT>
__device__ __forceinline__ unsigned mx_pack4(float a, float b, float c, float d, float mult)
{
  unsigned w = 0u;
  if (FMT == PD_MX8_E4M3) {
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a * mult, b * mult, w, false); w = __builtin_amdgcn_cvt_pk_fp8_f32(c * mult, d * mult, w, true);
  } else {
    w = __builtin_amdgcn_cvt_pk_bf8_f32(a * mult, b * mult, w, false); w = __builtin_amdgcn_cvt_pk_bf8_f32(c * mult, d * mult, w, true);
  }
  return w;
}

template <int FMT>
__device__ __forceinline__ uint2 mx_pack8(const float (&v)[8], float mult)
{
  return make_uint2(mx_pack4<FMT>(v[0], v[1], v[2], v[3], mult), mx_pack4<FMT>(v[4], v[5], v[6], v[7], mult));
}
}  // namespace pdmx
#endif
