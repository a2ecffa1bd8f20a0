// Workgroup g of a launch runs on XCD g % 8 (MI355X: 8 XCDs, each with its own 4 MB L2; the L2s do not share lines).  Kernels whose
// neighbouring logical blocks share operands want them on ONE XCD, i.e. logical blocks handed out XCD-major: XCD x works through a
// contiguous run of logical indices.
#ifndef PD_XCD_H
#define PD_XCD_H
#include <hip/hip_runtime.h>

// position of workgroup g among the n workgroups [first, first + n) of a (sub-)launch in XCD-major order:
// (workgroups of the range on lower-numbered XCDs) + (rank among this XCD's).  A bijection onto [0, n) for ANY first / n.
__device__ __forceinline__ int pd_xcd_major(int first, int n, int g)
{
  const int x = g & 7;
  int pos = 0;
#pragma unroll
  for (int y = 0; y < 7; ++y) {
    const int f = first + ((y - first) & 7);               // the range's first workgroup on XCD y
    if (y < x && f < first + n) pos += (first + n - 1 - f) / 8 + 1;
  }
  return pos + (g - (first + ((x - first) & 7))) / 8;
}

// the whole launch: logical block of workgroup bid of nb.  nb % 8 == 0 is the closed form (bid % 8) * (nb / 8) + bid / 8; other counts used to
// fall back to the identity — the FPN 3 x 3 filter gradient's 252 workgroups (nine tap tiles per pixel slice on nine different XCDs) moved
// 8.7 x its operands through the fabric that way.
__device__ __forceinline__ int pd_xcd_chunk(int bid, int nb) { return (nb & 7) == 0 ? (bid & 7) * (nb >> 3) + (bid >> 3) : pd_xcd_major(0, nb, bid); }
#endif
