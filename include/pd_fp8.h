/*
 * pd_fp8.h — C-ABI of the fp8 quantisation kernels of libpd_hip.so (BASELINE config 5: "fp8 MFMA GEMMs").
 *
 * The GEMMs of config 5 — the Swin-L qkv / proj / MLP Linears, reference modeling/backbone/swin.py:58-70 (Mlp),
 * :128-131,146,171-173 (WindowAttention qkv / proj) — run as plain library fp8 GEMMs (hipBLASLt, e4m3 x e4m3 forward,
 * e5m2 x e4m3 input gradient, fp32 accumulate); what the library does not do is get the operands there.  These kernels
 * do: per-tensor "current scaling" (scale = format_max / amax(|x|)) computed ON THE DEVICE, no host round trip:
 *   pd_fp8_amax      amax[0] = max(amax[0], max |x|)              (caller zero-fills amax; one pass over x)
 *   pd_fp8_quantize  out = sat(x * scale) in OCP e4m3 / e5m2 (v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32, round to nearest
 *                    even, saturating at +-448 / +-57344), scale_inv[0] = amax / format_max = the GEMM's de-scale
 * x: fp32 or bf16 (`dtype` = PD_F32 / PD_BF16 of pd_msda.h), contiguous, n % 8 == 0, 16-byte aligned.
 * amax == 0 quantises with scale 1.  `stream` = hipStream_t; returns 0 or PD_ERR_*.
 */
#ifndef PD_FP8_H
#define PD_FP8_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PD_FP8_E4M3 0
#define PD_FP8_E5M2 1

int pd_fp8_amax(const void *x, int64_t n, int dtype, float *amax, void *stream);
int pd_fp8_quantize(const void *x, int64_t n, int dtype, const float *amax, int format, uint8_t *out, float *scale_inv,
                    void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PD_FP8_H */
