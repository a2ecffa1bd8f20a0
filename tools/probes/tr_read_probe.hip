// what ds_read_b64_tr_b16 (gfx950 LDS transpose read) returns: every lane points at its own 8 bytes (4 x b16) of LDS;
// LDS[j] = j, lane L's address = 8 L bytes, so a returned value v came from lane v / 4's segment, element v % 4.
// hipcc --offload-arch=gfx950 -O2 tr_read_probe.hip -o tr_read_probe && ./tr_read_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void probe(uint16_t *out)
{
  __shared__ __attribute__((aligned(16))) uint16_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const uint32_t addr = (uint32_t)(uintptr_t)lds + threadIdx.x * 8;
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[threadIdx.x * 4 + 0] = v.x & 0xffff; out[threadIdx.x * 4 + 1] = v.x >> 16;
  out[threadIdx.x * 4 + 2] = v.y & 0xffff; out[threadIdx.x * 4 + 3] = v.y >> 16;
}

int main()
{
  uint16_t *d, h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) printf("  (lane %2d, elem %d)", h[l * 4 + e] / 4, h[l * 4 + e] % 4);
    printf("\n");
  }
  return 0;
}
