// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950 by system identification: which (lane, byte) of the A / B registers is which
// (row, k), which lane's scale byte applies to which (row, k block), what the op-select immediates pick.
//   hipcc -w --offload-arch=gfx950 -O2 -o /tmp/mx8 tools/probes/mfma_mx8_probe.hip && /tmp/mx8
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// raw: per-lane register images in, per-lane accumulators out
template <int FA, int FB, int OA, int OB>
__global__ void raw(const uint32_t *A, const uint32_t *B, const uint32_t *sa, const uint32_t *sb, float *D)
{
  const int l = threadIdx.x;
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (int)A[l * 8 + i]; b[i] = (int)B[l * 8 + i]; }
  v16f c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, FA, FB, OA, (int)sa[l], OB, (int)sb[l]);
  for (int i = 0; i < 16; ++i) D[l * 16 + i] = c[i];
}

static uint32_t *dA, *dB, *dsa, *dsb;
static float *dD;
static uint8_t hA[64 * 32], hB[64 * 32];
static uint32_t hsa[64], hsb[64];
static float hD[64 * 16];

template <int FA, int FB, int OA, int OB>
static void go()
{
  (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  (void)hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((raw<FA, FB, OA, OB>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
  (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
}

static void unit_scales() { for (int l = 0; l < 64; ++l) hsa[l] = hsb[l] = 0x7f7f7f7fu; }

int main()
{
  (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dsa, sizeof hsa); (void)hipMalloc(&dsb, sizeof hsb); (void)hipMalloc(&dD, sizeof hD);
  const uint8_t ONE = 0x38;   // 1.0 in e4m3
  // ---- 1. D layout and A rows: A one-hot at (lane la, byte 0), B all ones -> non-zero D entries
  printf("== A one-hot (lane, byte 0) x B all-ones: non-zero accumulators\n");
  for (int la : {0, 1, 5, 31, 32, 33, 63}) {
    memset(hA, 0, sizeof hA); memset(hB, ONE, sizeof hB); unit_scales();
    hA[la * 32] = ONE;
    go<0, 0, 0, 0>();
    int regs_mask = 0, lanes_lo = 64, lanes_hi = -1; float val = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) if (hD[l * 16 + i] != 0.f) { regs_mask |= 1 << i; lanes_lo = l < lanes_lo ? l : lanes_lo; lanes_hi = l > lanes_hi ? l : lanes_hi; val = hD[l * 16 + i]; }
    printf("  A lane %2d: regs mask 0x%04x lanes %d..%d value %g\n", la, regs_mask, lanes_lo, lanes_hi, val);
  }
  printf("== B one-hot (lane, byte 0) x A all-ones\n");
  for (int lb : {0, 1, 5, 31, 32, 33, 63}) {
    memset(hB, 0, sizeof hB); memset(hA, ONE, sizeof hA); unit_scales();
    hB[lb * 32] = ONE;
    go<0, 0, 0, 0>();
    int regs_mask = 0, lanes_lo = 64, lanes_hi = -1; float val = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) if (hD[l * 16 + i] != 0.f) { regs_mask |= 1 << i; lanes_lo = l < lanes_lo ? l : lanes_lo; lanes_hi = l > lanes_hi ? l : lanes_hi; val = hD[l * 16 + i]; }
    printf("  B lane %2d: regs mask 0x%04x lanes %d..%d value %g\n", lb, regs_mask, lanes_lo, lanes_hi, val);
  }
  // ---- 2. k pairing: A one-hot (row 0: lane 0 or 32, byte j) x B one-hot (col 0: lane 0 or 32, byte j') -> D[0][0] (lane 0 reg 0)
  printf("== k pairing: for A (half, byte) which B (half, byte) gives a product\n");
  int ident = 1;
  for (int ka = 0; ka < 64; ++ka) {
    int found = -1, nfound = 0;
    for (int kb = 0; kb < 64; ++kb) {
      memset(hA, 0, sizeof hA); memset(hB, 0, sizeof hB); unit_scales();
      hA[(ka >> 5) * 32 * 32 + (ka & 31)] = ONE;
      hB[(kb >> 5) * 32 * 32 + (kb & 31)] = ONE;
      go<0, 0, 0, 0>();
      float s = 0; for (int i = 0; i < 64 * 16; ++i) s += fabsf(hD[i]);
      if (s != 0.f) { found = kb; ++nfound; }
    }
    if (found != ka || nfound != 1) { ident = 0; printf("  A k-slot %d pairs with B k-slot %d (%d matches)\n", ka, found, nfound); }
  }
  printf("  k pairing is %s\n", ident ? "the IDENTITY: slot (half h, byte j) of A meets slot (h, j) of B" : "NOT the identity");
  // ---- 3. scales: all-ones data (D = 64), one lane of scale A doubled (128) in ALL four bytes -> which D entries change
  printf("== scale A: lane x holds 0x80808080, others 0x7f7f7f7f, opsel 0; data all ones\n");
  for (int x : {0, 3, 31, 32, 35, 63}) {
    memset(hA, ONE, sizeof hA); memset(hB, ONE, sizeof hB); unit_scales();
    hsa[x] = 0x80808080u;
    go<0, 0, 0, 0>();
    int regs_mask = 0, lanes_lo = 64, lanes_hi = -1; float val = 0, base = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) { if (hD[l * 16 + i] != 64.f) { regs_mask |= 1 << i; lanes_lo = l < lanes_lo ? l : lanes_lo; lanes_hi = l > lanes_hi ? l : lanes_hi; val = hD[l * 16 + i]; } else base = 64.f; }
    printf("  scale-A lane %2d: changed regs mask 0x%04x lanes %d..%d value %g (base %g)\n", x, regs_mask, lanes_lo, lanes_hi, val, base);
  }
  printf("== scale B: lane x doubled\n");
  for (int x : {0, 3, 31, 32, 35, 63}) {
    memset(hA, ONE, sizeof hA); memset(hB, ONE, sizeof hB); unit_scales();
    hsb[x] = 0x80808080u;
    go<0, 0, 0, 0>();
    int regs_mask = 0, lanes_lo = 64, lanes_hi = -1; float val = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) if (hD[l * 16 + i] != 64.f) { regs_mask |= 1 << i; lanes_lo = l < lanes_lo ? l : lanes_lo; lanes_hi = l > lanes_hi ? l : lanes_hi; val = hD[l * 16 + i]; }
    printf("  scale-B lane %2d: changed regs mask 0x%04x lanes %d..%d value %g\n", x, regs_mask, lanes_lo, lanes_hi, val);
  }
  // ---- 4. which k a scale covers: scale A lane 0 doubled, A row 0 one-hot at k-slot j, B all ones -> D[0][*] = 2 or 1
  printf("== scale A lane 0 / lane 32 doubled: which k slots of row 0 it multiplies\n");
  for (int x : {0, 32}) {
    printf("  lane %2d: ", x);
    for (int ka = 0; ka < 64; ++ka) {
      memset(hA, 0, sizeof hA); memset(hB, ONE, sizeof hB); unit_scales();
      hsa[x] = 0x80808080u;
      hA[(ka >> 5) * 32 * 32 + (ka & 31)] = ONE;
      go<0, 0, 0, 0>();
      float mx = 0; for (int i = 0; i < 64 * 16; ++i) mx = fmaxf(mx, hD[i]);
      printf("%g", mx);
    }
    printf("\n");
  }
  // ---- 5. op-select: scale A lane 0 = bytes {0x80, 0x81, 0x82, 0x83} (x2, x4, x8, x16), A row 0 k-slot 0 one-hot
  printf("== op-select of scale A (bytes x2 x4 x8 x16 from byte 0 up): product seen\n");
  memset(hA, 0, sizeof hA); memset(hB, ONE, sizeof hB); unit_scales();
  hsa[0] = 0x83828180u; hA[0] = ONE;
  { go<0, 0, 0, 0>(); float mx = 0; for (int i = 0; i < 64 * 16; ++i) mx = fmaxf(mx, hD[i]); printf("  opsel 0 -> x%g\n", mx); }
  { go<0, 0, 1, 0>(); float mx = 0; for (int i = 0; i < 64 * 16; ++i) mx = fmaxf(mx, hD[i]); printf("  opsel 1 -> x%g\n", mx); }
  { go<0, 0, 2, 0>(); float mx = 0; for (int i = 0; i < 64 * 16; ++i) mx = fmaxf(mx, hD[i]); printf("  opsel 2 -> x%g\n", mx); }
  { go<0, 0, 3, 0>(); float mx = 0; for (int i = 0; i < 64 * 16; ++i) mx = fmaxf(mx, hD[i]); printf("  opsel 3 -> x%g\n", mx); }
  printf("== op-select of scale B likewise\n");
  memset(hB, 0, sizeof hB); memset(hA, ONE, sizeof hA); unit_scales();
  hsb[0] = 0x83828180u; hB[0] = ONE;
  { go<0, 0, 0, 0>(); float mx = 0; for (int i = 0; i < 64 * 16; ++i) mx = fmaxf(mx, hD[i]); printf("  opsel 0 -> x%g\n", mx); }
  { go<0, 0, 0, 1>(); float mx = 0; for (int i = 0; i < 64 * 16; ++i) mx = fmaxf(mx, hD[i]); printf("  opsel 1 -> x%g\n", mx); }
  { go<0, 0, 0, 2>(); float mx = 0; for (int i = 0; i < 64 * 16; ++i) mx = fmaxf(mx, hD[i]); printf("  opsel 2 -> x%g\n", mx); }
  { go<0, 0, 0, 3>(); float mx = 0; for (int i = 0; i < 64 * 16; ++i) mx = fmaxf(mx, hD[i]); printf("  opsel 3 -> x%g\n", mx); }
  // ---- 6. e5m2 decode: A = 0x3c (1.0 in e5m2) with cbsz 1
  memset(hA, 0, sizeof hA); memset(hB, ONE, sizeof hB); unit_scales(); hA[0] = 0x3c;
  { go<1, 0, 0, 0>(); float mx = 0; for (int i = 0; i < 64 * 16; ++i) mx = fmaxf(mx, hD[i]); printf("== cbsz 1: A byte 0x3c reads as %g (e5m2 1.0 expected)\n", mx); }
  memset(hB, 0, sizeof hB); memset(hA, ONE, sizeof hA); unit_scales(); hB[0] = 0x3c;
  { go<0, 1, 0, 0>(); float mx = 0; for (int i = 0; i < 64 * 16; ++i) mx = fmaxf(mx, hD[i]); printf("== blgp 1: B byte 0x3c reads as %g (e5m2 1.0 expected)\n", mx); }
  return 0;
}
