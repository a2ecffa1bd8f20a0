// How fast can ONE workgroup pull a weight stream it reads exactly once?  (the bound of csrc/declayer.hip: 13 workgroups x 2.7 MB)
//   pattern 0: the MFMA-operand pattern of declayer.hip (16 rows x 64 contiguous bytes per wave instruction, row pitch 512 B)
//   pattern 1: 1 KB contiguous per wave instruction (a pre-arranged "fragment order" copy of the weights would read like this)
//   pattern 2: pattern 0 with row pitch 4096 B (linear2's [256, 2048] weight)
//   AHEAD: 16-load blocks in flight beside the one being consumed;  WAVES per workgroup 8 or 16;  nt: non-temporal loads
// build: hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip ; run: ./stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int PATTERN, int AHEAD, bool NT>
__global__ __launch_bounds__(1024) void stream(const uint4 *__restrict__ w, size_t window_u4, int nblk, unsigned *out)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // every workgroup reads the SAME window (like the decoder's row blocks read the same weights); wave w takes blocks w, w + nw, ...
  uint4 buf[AHEAD + 1][16];
  auto addr = [&](int blk, int i) -> const uint4 * {
    if (PATTERN == 1) return w + (size_t)blk * 1024 + i * 64 + lane;                       // 16 KB blocks, 1 KB per instruction
    const int pitch_u4 = PATTERN == 0 ? 32 : 256;                                            // 512 B / 4096 B rows
    // block = 32 rows x 512 B: i = t * 8 + s: row 16 t + (lane & 15), 64-byte piece s, 16-byte part lane >> 4
    const size_t row = (size_t)blk * 32 + 16 * (i >> 3) + (lane & 15);
    return w + (PATTERN == 0 ? row * pitch_u4 + (i & 7) * 4 + (lane >> 4)
                             : (row % 256) * pitch_u4 + (row / 256) * 32 + (i & 7) * 4 + (lane >> 4));
  };
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  auto ld = [&](const uint4 *p) -> uint4 {
    if (!NT) return *p;
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  };
  unsigned acc = 0;
  int issued = 0;
  for (int a = 0; a < AHEAD; ++a, ++issued) {
    const int blk = wave + issued * nw;
    if (blk < nblk)
#pragma unroll
      for (int i = 0; i < 16; ++i) buf[a][i] = ld(addr(blk, i));
  }
  int slot = 0;
  for (int b = wave; b < nblk; b += nw) {
    const int nb = wave + issued * nw;
    const int ns = (slot + AHEAD) % (AHEAD + 1);
    if (nb < nblk) {
#pragma unroll
      for (int s2 = 0; s2 <= AHEAD; ++s2)
        if (s2 == ns)
#pragma unroll
          for (int i = 0; i < 16; ++i) buf[s2][i] = ld(addr(nb, i));
    }
    ++issued;
#pragma unroll
    for (int s2 = 0; s2 <= AHEAD; ++s2)
      if (s2 == slot)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc ^= buf[s2][i].x ^ buf[s2][i].y ^ buf[s2][i].z ^ buf[s2][i].w;
    slot = (slot + 1) % (AHEAD + 1);
  }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

template <int PATTERN, int AHEAD, bool NT>
void run(const char *name, const uint4 *buf, size_t total_u4, unsigned *out, int wgs, int waves, size_t bytes)
{
  const int nblk = (int)(bytes / 16384);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20;
  float best = 1e9f, sum = 0.f;
  for (int it = 0; it < iters; ++it) {
    const uint4 *w = buf + ((size_t)it * (bytes / 16 + 4096)) % (total_u4 - bytes / 16);      // a fresh window each launch
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream<PATTERN, AHEAD, NT>), dim3(wgs), dim3(64 * waves), 0, 0, w, bytes / 16, nblk, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it >= 2) { best = ms < best ? ms : best; sum += ms; }
  }
  printf("%-44s wgs %3d waves %2d  %5.1f us avg  %5.1f us best  -> %6.1f GB/s per workgroup\n", name, wgs, waves, sum / (iters - 2) * 1e3, best * 1e3,
         bytes / (sum / (iters - 2) * 1e-3) / 1e9);
}

int main()
{
  const size_t total = 256u << 20;
  uint4 *buf; unsigned *out;
  hipMalloc(&buf, total); hipMalloc(&out, 4096);
  hipMemset(buf, 1, total);
  const size_t bytes = 2752512;        // 168 blocks of 16 KB = 2.7 MB
  for (int wgs : {1, 13, 26, 104}) {
    run<0, 1, false>("rows 512 B, 1 block ahead", buf, total / 16, out, wgs, 8, bytes);
    run<0, 2, false>("rows 512 B, 2 blocks ahead", buf, total / 16, out, wgs, 8, bytes);
    run<0, 1, false>("rows 512 B, 1 block ahead, 16 waves", buf, total / 16, out, wgs, 16, bytes);
    run<1, 1, false>("contiguous 1 KB, 1 block ahead", buf, total / 16, out, wgs, 8, bytes);
    run<1, 2, false>("contiguous 1 KB, 2 blocks ahead", buf, total / 16, out, wgs, 8, bytes);
    run<1, 1, false>("contiguous 1 KB, 1 block ahead, 16 waves", buf, total / 16, out, wgs, 16, bytes);
    run<1, 1, true>("contiguous 1 KB, 1 ahead, nt", buf, total / 16, out, wgs, 8, bytes);
    run<2, 1, false>("rows 4096 B, 1 block ahead", buf, total / 16, out, wgs, 8, bytes);
  }
  return 0;
}
