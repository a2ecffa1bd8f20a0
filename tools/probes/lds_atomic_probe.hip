// Probe: throughput of LDS atomics (f32 add, u32 add) vs plain LDS read-modify-write on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, int stride)
{
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0.f;
  __syncthreads();
  unsigned a = (threadIdx.x * stride) & 8191;
  float v = 1.0f + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      unsigned addr = (a + u * 256 * stride) & 8191;
      if (MODE == 0) atomicAdd(&lds[addr], v);
      else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned *>(&lds[addr]), (unsigned)it);
      else if (MODE == 2) lds[addr] += v;
      else if (MODE == 3) __hip_atomic_fetch_add(&lds[addr], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
}

int main()
{
  float *out;
  hipMalloc(&out, 1024 * 256 * 4);
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  const int iters = 200, blocks = 1024;
  for (int stride : {1, 4, 33}) {
    for (int mode = 0; mode < 4; ++mode) {
      auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, stride);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, stride);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, stride);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters, stride);
      };
      launch();
      hipDeviceSynchronize();
      hipEventRecord(s);
      launch();
      hipEventRecord(e);
      hipEventSynchronize(e);
      float ms;
      hipEventElapsedTime(&ms, s, e);
      double ops = (double)blocks * 256 * iters * 16;
      printf("stride %2d mode %d (%s): %.3f ms  %.1f Glane-ops/s  (%.2f lane-ops/clk/CU @2.1GHz)\n", stride, mode,
             mode == 0 ? "ds_add_f32 atomicAdd" : mode == 1 ? "ds_add_u32" : mode == 2 ? "plain RMW" : "hip_atomic f32 wg",
             ms, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.1);
    }
  }
  return 0;
}
