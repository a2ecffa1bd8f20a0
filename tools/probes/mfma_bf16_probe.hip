// Probe of v_mfma_f32_32x32x8_bf16_1k operand / result layouts (development tool; run on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 mfma_bf16_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned short f2bf(float f) { return (unsigned short)(__float_as_uint(f) >> 16); }   // exact for small ints

// C[i][j] = sum_k X[i][k] Y[j][k], X, Y: [32][8] row-major floats (small ints); out[i*32+j]
__global__ void probe(const float *X, const float *Y, float *out, float *out_pk)
{
  const int lane = threadIdx.x, r = lane & 31, hh = lane >> 5;
  bf16x4 a, b;
  for (int i = 0; i < 4; ++i) { a[i] = (short)f2bf(X[r * 8 + 4 * hh + i]); b[i] = (short)f2bf(Y[r * 8 + 4 * hh + i]); }
  f32x16 c;
  for (int e = 0; e < 16; ++e) c[e] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c, 0, 0, 0);
  for (int e = 0; e < 16; ++e) {
    const int i = (e & 3) + 8 * (e >> 2) + 4 * hh;
    out[i * 32 + r] = c[e];
  }
  unsigned pk;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(1.0f + lane), "v"(-2.0f));
  out_pk[lane] = __uint_as_float(pk);
}

int main()
{
  float hX[256], hY[256], hO[1024], hP[64];
  for (int i = 0; i < 256; ++i) { hX[i] = (float)((i * 7) % 13 - 6); hY[i] = (float)((i * 5) % 11 - 5); }
  float *dX, *dY, *dO, *dP;
  hipMalloc(&dX, sizeof hX); hipMalloc(&dY, sizeof hY); hipMalloc(&dO, sizeof hO); hipMalloc(&dP, sizeof hP);
  hipMemcpy(dX, hX, sizeof hX, hipMemcpyHostToDevice); hipMemcpy(dY, hY, sizeof hY, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dX, dY, dO, dP);
  hipMemcpy(hO, dO, sizeof hO, hipMemcpyDeviceToHost); hipMemcpy(hP, dP, sizeof hP, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      float ref = 0;
      for (int k = 0; k < 8; ++k) ref += hX[i * 8 + k] * hY[j * 8 + k];
      if (fabsf(ref - hO[i * 32 + j]) > 1e-3f) { if (bad < 5) printf("mismatch C[%d][%d] = %g want %g\n", i, j, hO[i * 32 + j], ref); ++bad; }
    }
  printf("layout assumption (C = X.Y^T, lane = Y row, reg e = X row (e&3)+8(e>>2)+4hh): %s (%d bad)\n", bad ? "WRONG" : "ok", bad);
  unsigned u; memcpy(&u, &hP[3], 4);
  printf("cvt_pk_bf16(4.0, -2.0) = 0x%08x  (lo half should be 0x4080 = 4.0, hi 0xc000 = -2.0)\n", u);
  return 0;
}
