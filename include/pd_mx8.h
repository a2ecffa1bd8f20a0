/*
 * pd_mx8.h — C-ABI of the MX-fp8 GEMM family of libpd_hip.so (csrc/mx8.hip): BASELINE config 5's "fp8 MFMA GEMMs".
 *
 * What it replaces: the qkv / proj / Mlp fc1 / fc2 Linears of the Swin backbones, forward and input gradient
 * (part_distillation/modeling/backbone/swin.py:34-36 Mlp.fc1 / fc2, :127-129 WindowAttention.qkv / proj) — in the reference bf16 / fp16
 * library GEMMs under AMP; config 5 asks for them on the fp8 matrix cores.
 *
 * Format: OCP Microscaling (MX) fp8 — the format gfx950's matrix cores consume natively (v_mfma_scale_f32_32x32x64_f8f6f4, twice the
 * bf16 rate; the plain fp8 instructions run at the bf16 rate):
 *   elements   fp8 e4m3 (activations, weights) or e5m2 (gradients), one byte each, row-major [rows][k] along the CONTRACTION axis k
 *   scales     one E8M0 byte (value 2^(byte - 127)) per 32 consecutive elements of a row: [rows][k / 32]
 *   x[r][c] ~= 2^(s[r][c / 32] - 127) * fp8(q[r][c])
 * The shared exponent of a block is the smallest X with amax(block) * 2^-X <= format maximum (448 | 57 344), clamped to [-126, 126]:
 * nothing saturates (the OCP recipe floor(log2 amax) - emax clips the top of a block whose amax has a mantissa above 1.75).
 * Quantisation is local to 32 elements — no tensor-wide amax pass, no scale history — so it rides in the epilogue of whatever kernel
 * produces the operand (pd_mx8_gemm's out_q / out_s; a standalone pass for operands other kernels produce).
 *
 * All pointers are device pointers; `stream` = hipStream_t; returns 0 or PD_ERR_* (pd_msda.h) with pd_last_error() set.
 */
#ifndef PD_MX8_H
#define PD_MX8_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PD_MX8_E4M3 0
#define PD_MX8_E5M2 1

/* x [rows][cols] bf16, row stride ldx elements (cols % 32 == 0, ldx % 8 == 0, x 16-byte aligned)
 * -> q [rows][cols] fp8 bytes (8-byte aligned), s [rows][cols / 32] E8M0 bytes.  One streaming pass. */
int pd_mx8_quantize_bf16(const void *x, int64_t rows, int32_t cols, int64_t ldx, int32_t format, uint8_t *q, uint8_t *s, void *stream);

/* `count` contiguous bf16 tensors (numel % 32 == 0 each; a [n][k] weight with k % 32 == 0 is one) in ONE launch: the weights of a whole
 * stage, re-quantised after every optimizer step.  table_host_pinned / table_device: caller-provided staging of
 * pd_mx8_quantize_table_bytes(count) bytes each; the pinned one must stay untouched until the asynchronous copy has executed. */
typedef struct PdMx8Tensor {
  const void *x;                      /* bf16, contiguous, 16-byte aligned */
  uint8_t *q, *s;                     /* [numel], [numel / 32] */
  int64_t numel;
} PdMx8Tensor;
int64_t pd_mx8_quantize_table_bytes(int32_t count);
int pd_mx8_quantize_grouped(const PdMx8Tensor *list, int32_t count, int32_t format, void *table_host_pinned, void *table_device, void *stream);

/* out[m][n] = epilogue( sum_k A[m][k] W[n][k] )  with MX operands, fp32 accumulation, ONE rounding to bf16:
 *   + bias[n]  ->  (out_pre = that)  ->  act (PD_IG_ACT_NONE | PD_IG_ACT_GELU of pd_igemm.h)  ->  * gelu'(gate[m][n]) (PD_IG_GATE_GELU)
 *   -> out (bf16) and, when out_q is given, the same values again as MX fp8 along n (the next GEMM's operand).
 * nn.Linear forward: A = activations (e4m3), W = weight as it lies [out][in];  input gradient: A = dY (e5m2 or e4m3), W = the [in][out]
 * transpose.  m >= 1, n % 64 == 0, k % 128 == 0; q pointers 16-byte aligned, rows dense. */
typedef struct PdMx8Gemm {
  const uint8_t *a_q, *a_s;           /* [m][k], [m][k / 32] */
  const uint8_t *w_q, *w_s;           /* [n][k] e4m3, [n][k / 32] */
  const void *bias;                   /* nullable: fp32 [n], or bf16 [n] with bias_bf16 != 0 */
  const void *gate;                   /* nullable: bf16 [m][n], see gate_mode */
  void *out;                          /* bf16 [m][n] */
  void *out_pre;                      /* nullable: bf16 [m][n], the value before act */
  uint8_t *out_q, *out_s;             /* nullable: [m][n] fp8 of out_format, [m][n / 32] */
  int32_t m, n, k;
  int32_t a_format;                   /* PD_MX8_E4M3 | PD_MX8_E5M2 */
  int32_t act, gate_mode, bias_bf16;
  int32_t out_format;
} PdMx8Gemm;
int pd_mx8_gemm(const PdMx8Gemm *p, void *stream);
int pd_mx8_gemm_supported(int32_t m, int32_t n, int32_t k);

#ifdef __cplusplus
}
#endif
#endif /* PD_MX8_H */
