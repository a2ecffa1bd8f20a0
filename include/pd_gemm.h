/*
 * pd_gemm.h — C-ABI of the fp32 MFMA GEMMs of libpd_hip.so.
 *
 * The reference keeps the pixel decoder in fp32 under AMP
 * (pixel_decoder/msdeformattn.py:318 `@autocast(enabled=False)`, :324/:348 `.float()`), so its six encoder layers
 * run ~1.1 TFLOP of fp32 nn.Linear work per step at config 2 (ops/modules/ms_deform_attn.py:102-107,130;
 * msdeformattn.py:120-124).  These kernels do that work on the matrix cores with v_mfma_f32_32x32x2_f32 (exact fp32
 * FMA chains, no TF32/bf16 rounding).  Row-major operands, device pointers, `stream` = hipStream_t.
 */
#ifndef PD_GEMM_H
#define PD_GEMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* C[M,N] = A[M,K] . B[N,K]^T (+ bias[N]) (then ReLU if relu != 0).  K % 4 == 0, lda/ldb/ldc % 4 == 0, 16-byte
 * aligned pointers.  nn.Linear forward (A = x, B = weight) and its input gradient (A = dy, B = weight^T). */
int pd_gemm_tn_f32(const float *A, const float *B, const float *bias, float *C, int M, int N, int K, int lda, int ldb,
                   int ldc, int relu, void *stream);

/* The same product with the fp32 operands split exactly into 3 bf16 values each and 6 of the 9 partial products run on the
 * bf16 matrix cores (fp32 accumulate; the dropped terms are <= 2^-23 |a b|, one fp32 rounding of the product): fp32-level
 * results at 2.7x the fp32 matrix rate.  Same arguments and constraints as pd_gemm_tn_f32. */
int pd_gemm_tn_f32x3(const float *A, const float *B, const float *bias, float *C, int M, int N, int K, int lda, int ldb,
                     int ldc, int relu, void *stream);

/* dW[N,K] = dY[M,N]^T . X[M,K]  (nn.Linear weight gradient; contraction over the M rows, split over workgroups and
 * combined with fp32 atomics: dW is zero-filled by the library first) and, when dB != NULL, the bias gradient
 * dB[N] = column sums of dY from the same pass.  N % 4 == 0, K % 4 == 0. */
int pd_gemm_wgrad_f32(const float *dY, const float *X, float *dW, float *dB, int M, int N, int K, int ldy, int ldx,
                      int ldw, void *stream);

/* Same, accumulating: dW += dY^T . X and dB += column sums, into buffers the caller has initialised (one memset for
 * all the weight gradients of a backward pass instead of one or two per GEMM). */
int pd_gemm_wgrad_acc_f32(const float *dY, const float *X, float *dW, float *dB, int M, int N, int K, int ldy, int ldx,
                          int ldw, void *stream);

/* The FFN's ReLU backward folded into the GEMM epilogues (256 x 256 tiles: N % 256 == 0, M >= 1024):
 *   pd_gemm_tn_f32x3_relu_bits   C = relu(A B^T + bias) and `bits` <- one bit per element of C (C > 0), laid out in the kernel's
 *                                own accumulator order; `bits` holds pd_gemm_tn_f32x3_relu_bits_words(M, N) 32-bit words
 *   pd_gemm_tn_f32x3_relumask    C = (A B^T) where the recorded bit is set, else 0;  colsum[N] += column sums of that C
 * over the same [M, N]: the gradient w.r.t. the FFN's hidden pre-activation and the bias gradient of its first Linear
 * (reference msdeformattn.py:120-124 through autograd) without a separate pass over the [M, N] tensor. */
int64_t pd_gemm_tn_f32x3_relu_bits_words(int M, int N);
/* Weight operand split ONCE per optimizer step instead of by every row tile of every GEMM:
 *   pd_split3_bf16        W fp32 [N, K] (row stride ldw) -> planes bf16 [3][N][K] (hi, mid, lo; exact: hi + mid + lo == W), or with
 *                         transpose != 0 the planes [3][K][N] of W^T (the operand of the input-gradient GEMM)
 *   pd_gemm_tn_f32x3_pre  pd_gemm_tn_f32x3 with B given as such planes [3][N][K]; mode 0: C = A B^T + bias, 1: relu (+ sign bits when
 *                         bits != NULL), 2: masked by bits, colsum += column sums.  N % 256 == 0, K % 16 == 0, M >= 1024. */
int pd_split3_bf16(const float *W, int N, int K, int ldw, int transpose, void *planes, void *stream);
int pd_gemm_tn_f32x3_pre(const float *A, const void *Bplanes, const float *bias, float *C, uint32_t *bits, float *colsum, int M, int N, int K,
                         int lda, int ldc, int mode, void *stream);
int pd_gemm_tn_f32x3_relu_bits(const float *A, const float *B, const float *bias, float *C, uint32_t *bits, int M, int N, int K, int lda,
                               int ldb, int ldc, void *stream);
int pd_gemm_tn_f32x3_relumask(const float *A, const float *B, const uint32_t *bits, float *C, float *colsum, int M, int N, int K, int lda,
                              int ldb, int ldc, void *stream);

/* pd_gemm_wgrad_acc_f32 with the 3-way bf16 split (see pd_gemm_tn_f32x3): dW += dY^T X, dB += column sums of dY (exact fp32
 * adds), accumulated into caller-initialised buffers.  Any N, K, M >= 0.  Operands whose rows allow 16-byte loads (N, K, ldy, ldx
 * multiples of 4, 16-byte aligned bases) take the ds_read_b64_tr_b16 transpose-read kernel, anything else the scalar-staged one. */
int pd_gemm_wgrad_acc_f32x3(const float *dY, const float *X, float *dW, float *dB, int M, int N, int K, int ldy, int ldx, int ldw,
                            void *stream);
/* The same with a caller-provided fp32 workspace (>= pd_gemm_wgrad_f32x3_ws_floats(N, K) elements, 16-byte aligned, free to be
 * reused by the next call on the same stream): the workgroups that share an output tile leave their partial tiles there with
 * plain stores and a second kernel sums them into dW, instead of 8.4 M fp32 atomics per call (30-110 us).  A NULL or too small
 * workspace falls back to the atomics. */
int64_t pd_gemm_wgrad_f32x3_ws_floats(int N, int K);
int pd_gemm_wgrad_acc_f32x3_ws(const float *dY, const float *X, float *dW, float *dB, float *workspace, int64_t workspace_floats, int M, int N,
                               int K, int ldy, int ldx, int ldw, void *stream);

/* Several pd_gemm_wgrad_acc_f32x3 problems as ONE launch (+ one reduce launch): dW_i[N_i,K_i] += dY_i[M_i,N_i]^T X_i[M_i,K_i],
 * dB_i += column sums of dY_i (dB nullable).  Weight gradients have no consumer until the optimizer runs, so the encoder queues
 * its 30 per step during the backward pass and runs them together (reference: the autograd of ops/modules/ms_deform_attn.py:102-130
 * and msdeformattn.py:120-124): whole rounds of workgroups once instead of 30 times and ~17 partial tiles per output tile instead
 * of 32-128.  Every problem: N, K, ldy, ldx multiples of 4, 16-byte aligned operands; count <= 256.
 *   table_host_pinned / table_device: pd_gemm_wgrad_f32x3_grouped_table_bytes(count) bytes each (pinned host staging the call
 *   fills + its device copy); workspace: >= pd_gemm_wgrad_f32x3_grouped_ws_floats(descs, count) floats (-1: bad problem). */
typedef struct PdGemmWgradDesc {
  const float *dY, *X;
  float *dW, *dB;
  int M, N, K, ldy, ldx, ldw;
  const float *y_amax, *x_amax;   /* absolute row maxima of dY / X [M] (nullable); read by the _f16x2 form only */
} PdGemmWgradDesc;
int64_t pd_gemm_wgrad_f32x3_grouped_table_bytes(int max_count);
int64_t pd_gemm_wgrad_f32x3_grouped_ws_floats(const PdGemmWgradDesc *descs, int count);
int pd_gemm_wgrad_f32x3_grouped(const PdGemmWgradDesc *descs, int count, void *table_host_pinned, void *table_device, float *workspace,
                                int64_t workspace_floats, void *stream);

/* Weight (and bias) gradient of pd_conv3x3_nhwc_f32x3's convolution (3 x 3, stride 1, pad 1; the fp32 FPN output convolution of
 * the pixel decoder, reference msdeformattn.py:238-257, 348-357), ACCUMULATED into caller-initialised buffers:
 *   dWk[co][tap][ci] += sum over pixels dY[p][co] X[p + off(tap)][ci],   dB[co] += sum over pixels dY[p][co]   (dB nullable)
 * dY [B,H,W,Co], X [B,H,W,Ci] NHWC fp32; dWk [Co][3][3][Ci] (the channels-last storage of the [Co][Ci][3][3] filter gradient).
 * Ci % 128 == 0, Co % 4 == 0.  workspace as in pd_gemm_wgrad_acc_f32x3_ws with (N, K) = (Co, 9 Ci); same 3-way bf16 split. */
int pd_conv3x3_wgrad_nhwc_f32x3(const float *dY, const float *X, float *dWk, float *dB, float *workspace, int64_t workspace_floats, int B, int H,
                                int W, int Ci, int Co, void *stream);

/* 3 x 3, stride 1, pad 1 convolution as an implicit GEMM on the same 3-way bf16 split (fp32-level results): the fp32 FPN
 * output convolution of the pixel decoder (reference pixel_decoder/msdeformattn.py:268-277, run at 1/4 resolution: 77 GFLOP per
 * 1024^2 image).  X [B,H,W,Ci] and Y [B,H,W,Co] channels-last, Wk [Co][3][3][Ci] (the channels-last filter), bias [Co] or NULL;
 * Ci % 16 == 0.  The input gradient is the same call on dY with the filter flipped and transposed ([Ci][3][3][Co]). */
int pd_conv3x3_nhwc_f32x3(const float *X, const float *Wk, const float *bias, float *Y, int B, int H, int W, int Ci, int Co,
                          void *stream);

/* fp32 GEMM on the fp16 matrix cores, two planes per operand and three products per term (csrc/gemm_f16x2.hip): every operand
 * row is scaled by a power of two taken from its absolute maximum, split as hi = fp16(x'), lo = fp16(x' - hi) and a b is
 * accumulated as hi hi + hi lo + lo hi in fp32 — per-element error <= max(2^-22 |x|, 2^-39 max_row |x|), i.e. the normwise
 * accuracy of an fp32 GEMM at half the matrix work of pd_gemm_tn_f32x3.  C[M,N] = A[M,K] B[N,K]^T:
 *   mode 0: + bias;  1: relu(+ bias) and, when bits != NULL, its sign bits (pd_gemm_tn_f16x2_bits_words(M, N) words, the
 *   kernel's accumulator order; N % 256 == 0, M >= 1024);  2: masked by `bits`, colsum[N] += column sums (bits, colsum required).
 *   a_amax[M] / b_amax[N]: absolute row maxima of A / B (fp32, any value >= the true maximum within a factor 2^7 keeps full
 *   accuracy; NULL = the operand is O(1), no scaling).  c_amax[M] (nullable, ZERO-FILLED by the caller): receives the absolute
 *   row maxima of C (atomic max) — the a_amax of the GEMM that consumes C.
 * K, lda, ldb multiples of 4, A and B 16-byte aligned.  Reference: the fp32 Linears of the pixel decoder,
 * pixel_decoder/msdeformattn.py:120-135,318 and ops/modules/ms_deform_attn.py:102-130.
 * pd_row_amax_f32: out[r] = max_c |X[r][c]| for operands whose producer does not emit the maxima. */
/* The weight-gradient kernels in the same two-plane form: the contraction runs over the rows, so each workgroup scales its slab
 * of rows by powers of two from the largest of the slab's row maxima (y_amax / x_amax [M], nullable = O(1) operand) and undoes
 * them on its partial tile.  Same arguments otherwise as pd_gemm_wgrad_acc_f32x3_ws / pd_gemm_wgrad_f32x3_grouped /
 * pd_conv3x3_wgrad_nhwc_f32x3; N, K, ldy, ldx multiples of 4, 16-byte aligned operands. */
/* workspace sizes of the two-plane weight gradients (they take 256 x 256 output tiles where the output has more than one 128-tile in
 * both directions: half the operand re-reads of the 128 x 128 kernel, which runs at the memory system's limit on them) */
int64_t pd_gemm_wgrad_f16x2_ws_floats(int N, int K);
int pd_gemm_wgrad_f16x2_takes_wide_tiles(int N, int K);   /* 1: this output shape runs on 256 x 256 tiles (a grouped launch does when ALL its problems do) */
int64_t pd_gemm_wgrad_f16x2_grouped_ws_floats(const PdGemmWgradDesc *descs, int count);
int pd_gemm_wgrad_acc_f16x2_ws(const float *dY, const float *X, float *dW, float *dB, const float *y_amax, const float *x_amax,
                               float *workspace, int64_t workspace_floats, int M, int N, int K, int ldy, int ldx, int ldw, void *stream);
int pd_gemm_wgrad_f16x2_grouped(const PdGemmWgradDesc *descs, int count, void *table_host_pinned, void *table_device, float *workspace,
                                int64_t workspace_floats, void *stream);
int pd_conv3x3_wgrad_nhwc_f16x2(const float *dY, const float *X, float *dWk, float *dB, const float *y_amax, const float *x_amax,
                                float *workspace, int64_t workspace_floats, int B, int H, int W, int Ci, int Co, void *stream);
/* host-only query (tools / bench.py labels): the kernel pd_gemm_tn_f16x2 takes for a problem — 0 = 128 x 128 tiles, 1 = 256 x 256 tiles,
 * 2 = the row stream (K = 256), 3 = the opt-in register-operand kernel */
int pd_gemm_tn_f16x2_which(int M, int N, int K, int mode, int has_bits, int has_amax);

/* pd_conv3x3_nhwc_f32x3 in the two-plane form: x_amax [B H W] = absolute maxima of X's pixels over their channels, w_amax [Co] =
 * row maxima of Wk (both nullable), y_amax [B H W] (nullable, zero-filled by the caller) receives those of Y.  Ci % 16 == 0. */
int pd_conv3x3_nhwc_f16x2(const float *X, const float *Wk, const float *bias, float *Y, const float *x_amax, const float *w_amax, float *y_amax,
                          int B, int H, int W, int Ci, int Co, void *stream);
int64_t pd_gemm_tn_f16x2_bits_words(int M, int N);
int pd_gemm_tn_f16x2(const float *A, const float *B, const float *bias, float *C, uint32_t *bits, float *colsum, const float *a_amax,
                     const float *b_amax, float *c_amax, int M, int N, int K, int lda, int ldb, int ldc, int mode, void *stream);
int pd_row_amax_f32(const float *X, int rows, int cols, int ld, float *out, void *stream);
/* pd_gemm_tn_f16x2 (mode 0, no row maxima out) with the result rounded to bf16 on the way out (C_bf16 [M, ldc] bf16): the input gradient of a
 * 1 x 1 convolution whose input was a bf16 backbone map — autograd's cast of the fp32 gradient (a pass of its own) is the epilogue. */
int pd_gemm_tn_f16x2_bf16out(const float *A, const float *B, const float *bias, void *C_bf16, const float *a_amax, const float *b_amax, int M, int N,
                             int K, int lda, int ldb, int ldc, void *stream);
/* X bf16 [rows, cols] contiguous (cols % 8 == 0) -> Y fp32 [rows, cols] = X and row_amax[r] = max_c |X[r, c]| in one pass: the reference's
 * `features[f].float()` in front of the pixel decoder's 1 x 1 convolutions (msdeformattn.py:324, 338) fused with the row-maxima pass. */
int pd_cast_bf16_f32_amax(const void *X_bf16, int rows, int cols, float *Y, float *row_amax, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PD_GEMM_H */
