// GELU (exact-erf form, torch.nn.GELU(approximate="none"); reference swin.py:31 Mlp.act) and its derivative for the GEMM epilogues.
//
//   Phi(x) = 0.5 erfc(-x / sqrt 2),   gelu(x) = x Phi(x),   gelu'(x) = Phi(x) + x phi(x),   phi(x) = exp(-x^2 / 2) / sqrt(2 pi)
//
// erfc through Abramowitz & Stegun 7.1.26 — erfc(z) = (a1 t + .. + a5 t^5) exp(-z^2), t = 1 / (1 + p z), z >= 0, |error| <= 1.5e-7
// absolute — whose exponential is the one phi needs anyway: 14 VALU instructions (one reciprocal, one exponential) where erff + __expf
// took ~50.  The epilogue of a 14 112 x 3 072 GELU' GEMM was VALU-bound on the erf (112 us against 50 for the same product without
// the gate); the result is rounded to bf16 (2^-9 relative), 10^4 times coarser than the approximation.
#ifndef PD_GELU_H
#define PD_GELU_H
#include <hip/hip_runtime.h>

namespace pdgelu {
__device__ __forceinline__ void parts(float x, float &Phi, float &phi)
{
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  const float e = __expf(-0.5f * x * x);
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float hc = 0.5f * p * t * e;                       // 0.5 erfc(|x| / sqrt 2)
  Phi = x >= 0.f ? 1.f - hc : hc;
  phi = e * 0.39894228040143267794f;
}
__device__ __forceinline__ float gelu(float x)
{
  float Phi, phi;
  parts(x, Phi, phi);
  return x * Phi;
}
__device__ __forceinline__ float gelu_grad(float x)
{
  float Phi, phi;
  parts(x, Phi, phi);
  return fmaf(x, phi, Phi);
}
}  // namespace pdgelu
#endif
