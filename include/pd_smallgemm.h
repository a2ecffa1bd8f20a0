/*
 * pd_smallgemm.h — C-ABI of the bf16 matrix-core GEMMs for SKINNY activations (tens to a few hundred rows) of
 * libpd_hip.so: the query side of the masked-attention decoder multiplies [Q*B = 200, 256..2048] activations by
 * 256 x 256 .. 2048 x 256 weights ~330 times per training step
 *   reference transformer_decoder/mask2former_transformer_decoder.py:44-54, 102-114, 167-171 (attention in/out
 *   projections, FFN) and :198-204 (the 3-layer mask-embedding MLP).
 * A general GEMM library maps such a product to ONE 256 x 256 tile = one workgroup (measured 14-18 us each, and ~20 us
 * of host time per call to pick it); these kernels cut the same product into 32 x 128 tiles so 14..128 workgroups share
 * it, take bias / ReLU / ReLU-mask / bias-gradient in the same pass, and are launched directly.
 *
 * All matrices row-major bf16 with leading dimensions in ELEMENTS (multiples of 8, 16-byte aligned bases); fp32
 * accumulation on v_mfma_f32_32x32x8_bf16_1k; results rounded to nearest even.  `stream` = hipStream_t; 0 or PD_ERR_*.
 */
#ifndef PD_SMALLGEMM_H
#define PD_SMALLGEMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Y[M,N] = X[M,K] . W[N,K]^T (+ bias[N]) (then ReLU if relu != 0)      nn.Linear forward.   K % 64 == 0, N % 4 == 0. */
int pd_sgemm_tn_bf16(const void *X, const void *W, const void *bias, void *Y, int M, int N, int K, int ldx, int ldw, int ldy,
                     int relu, void *stream);

/* dX[M,K] (+)= dY[M,N] . W[N,K], then dX *= (relu_ref[M,K] > 0) when relu_ref != NULL (ld = ldx)
 * nn.Linear input gradient, optionally through the ReLU that produced this layer's input.   N % 64 == 0, K % 4 == 0. */
int pd_sgemm_nn_bf16(const void *dY, const void *W, const void *relu_ref, void *dX, int M, int N, int K, int ldy, int ldw,
                     int ldx, int accumulate, void *stream);

/* The two products above for a LONG contraction over few rows (the decoder FFN: K = 2048 in pd_sgemm_tn, N = 2048 in pd_sgemm_nn; 14
 * output tiles that each walked the whole contraction as a chain of load latencies): the contraction is cut into 256-wide slices that
 * run as separate workgroups with all their loads in flight, fp32 partial tiles go to `workspace` and the last workgroup of a tile
 * to arrive sums them in slice order (deterministic) and applies the epilogue.  Contraction % 256 == 0 and >= 512.
 *   workspace: >= pd_sgemm_split_workspace_floats(M, output columns, contraction) floats; tickets: pd_sgemm_split_tickets(M, output
 *   columns) int32, ZERO before the first use (every launch leaves them zero); neither may be shared by launches that can overlap. */
int64_t pd_sgemm_split_workspace_floats(int M, int out_cols, int contraction);
int64_t pd_sgemm_split_tickets(int M, int out_cols);
int pd_sgemm_tn_splitk_bf16(const void *X, const void *W, const void *bias, void *Y, float *workspace, int64_t workspace_floats, int *tickets,
                            int M, int N, int K, int ldx, int ldw, int ldy, int relu, void *stream);
int pd_sgemm_nn_splitn_bf16(const void *dY, const void *W, const void *relu_ref, void *dX, float *workspace, int64_t workspace_floats,
                            int *tickets, int M, int N, int K, int ldy, int ldw, int ldx, int accumulate, void *stream);

/* `batch` equally shaped products Y_b [M, N] = X_b [M, K] W_b [N, K]^T (bf16, fp32 accumulation, K <= 256 and % 64 == 0, N % 4 == 0) in one
 * launch; element strides between consecutive problems.  The matcher's point logits of all (image, head) problems
 * (reference matcher.py:108-125: out_mask = point_sample(pred_masks) = mask_embed . point_sample(mask_features)). */
int pd_sgemm_tn_batched_bf16(const void *X, const void *W, void *Y, int M, int N, int K, int ldx, int ldw, int ldy, int batch,
                             int64_t stride_x, int64_t stride_w, int64_t stride_y, void *stream);

/* Up to four independent pd_sgemm_tn_bf16 products with a common K <= 256 (K % 64 == 0) in ONE launch: the q / k / v projections of an
 * attention block (two inputs, three slices of the packed in_proj weight; reference mask2former_transformer_decoder.py:44-54, 102-114
 * through nn.MultiheadAttention).  Rows may differ per problem (the cross-attention's keys / values run over the memory tokens). */
typedef struct PdSgemmTnDesc {
  const void *X, *W, *bias;   /* bias nullable */
  void *Y;
  int32_t M, N, ldx, ldw, ldy;
} PdSgemmTnDesc;
int pd_sgemm_tn_multi_bf16(const PdSgemmTnDesc *descs, int count, int K, void *stream);

/* One prediction head of the masked-attention decoder in one launch (reference mask2former_transformer_decoder.py:449-459 +
 * :198-204; the in-loop mask prediction carries no gradient, :457):  dec_out[R,256] = LayerNorm(tgt) in fp32 with mean / rstd [R];
 * when ef != NULL also e = W3 relu(W2 relu(W1 bf16(dec_out) + b1) + b2) + b3 (bf16 weights [256,256] / biases [256], fp32 accumulation,
 * bf16 roundings where the three Linears would round) written batch-major: ef[b][q][:] = e[q B + b][:], R = Q B rows, as fp32
 * (the bf16 values widened) or, ef_bf16 != 0, as bf16 (what an autocast bmm would cast them back to). */
int pd_decoder_head_bf16(const float *tgt, const float *ln_w, const float *ln_b, float eps, const void *w1, const void *b1, const void *w2,
                         const void *b2, const void *w3, const void *b3, float *dec_out, float *mean, float *rstd, void *ef, int ef_bf16, int R,
                         int B, int C, void *stream);

/* dW[N,K] = dY[M,N]^T . X[M,K];  dB[N] (fp32, nullable) = column sums of dY       nn.Linear weight / bias gradient.
 * Any M >= 0 (rows past M count as zeros); N % 4 == 0, K % 4 == 0. */
int pd_sgemm_wgrad_bf16(const void *dY, const void *X, void *dW, float *dB, int M, int N, int K, int ldy, int ldx, int ldw,
                        void *stream);

/* The same product for MANY rows (the 10^4..10^5 tokens of a Swin stage: swin.py:58-70, 128-131 Linear weight gradients):
 * the M rows are cut into slices that run as separate workgroups, fp32 partial tiles are summed by a second kernel
 * (deterministic, no atomics).  workspace: fp32, at least pd_sgemm_wgrad_split_workspace(M, N, K) elements, 16-byte
 * aligned.  M > 0; N % 4 == K % 4 == ldw % 4 == 0; dB (fp32 [N], nullable) = column sums of dY. */
int64_t pd_sgemm_wgrad_split_workspace(int M, int N, int K);
int pd_sgemm_wgrad_split_bf16(const void *dY, const void *X, void *dW, float *dB, float *workspace, int M, int N, int K, int ldy,
                              int ldx, int ldw, void *stream);

/* MANY pd_sgemm_wgrad_bf16 problems as one launch (the decoder's backward pass queues its ~70 weight gradients and runs them at
 * its end: a single one is a 7 us launch on a handful of CUs).  descs: host array; table_host_pinned / table_device: caller-
 * provided staging of pd_sgemm_wgrad_grouped_table_bytes(count) bytes each — the function fills the pinned one, copies it with
 * an asynchronous memcpy on `stream`, launches; the pinned buffer must stay untouched until the copy has executed. */
typedef struct PdSgemmWgradDesc {
  const void *dY, *X;
  void *dW;
  float *dB;
  int32_t M, N, K, ldy, ldx, ldw;
} PdSgemmWgradDesc;
int64_t pd_sgemm_wgrad_grouped_table_bytes(int count);
int pd_sgemm_wgrad_grouped_bf16(const PdSgemmWgradDesc *descs, int count, void *table_host_pinned, void *table_device, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PD_SMALLGEMM_H */
