/*
 * pd_kmeans.h — C-ABI of the device K-means (Lloyd) iteration of libpd_hip.so, batched over images.
 *
 * Replaces `sklearn.cluster.KMeans(n_clusters=K, random_state=0).fit(data)` on the CPU in the reference's pixel grouping
 * (proposal_generation_model.py:202-211; scikit-learn 1.7 `_kmeans_single_lloyd` semantics: an iteration assigns every
 * point to argmin_k |c_k|^2 - 2 x.c_k (first minimum) and moves the centres to the cluster means; it stops when no label
 * changed, or when the squared centre shift <= tol).  The convergence test lives on the device, so the host issues
 * iterations without reading anything back and looks at `done` only every few iterations.
 *
 * Points of all images are concatenated: X fp32 [N, C] (C % 4 == 0, C <= 2048), K <= 8 centres per image (kernels instantiated for 4 and 8); `blocks` is an
 * int32 [n_blocks, 3] table (image, first point, point count <= 64) so that no workgroup straddles two images.
 */
#ifndef PD_KMEANS_H
#define PD_KMEANS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * E-step + accumulation for every image b with done[b] == 0:
 *   labels[n] = argmin_k cnorm[b,k] - 2 x_n . centers[b,k]     changed[b] += #(labels[n] != previous labels[n])
 *   sums[b,k,:] += x_n,  counts[b,k] += 1   for the new label (sums / counts / changed must be zero on entry)
 */
int pd_kmeans_assign(const float *X, const int32_t *blocks, int n_blocks, const float *centers, const float *cnorm,
                     const int32_t *done, int32_t *labels, float *sums, float *counts, int32_t *changed, int C, int K, void *stream);

/*
 * The same E-step without atomics (the product path): workgroup i STORES the sums / counts of its slab to
 * partial_sums[i, K, C] / partial_counts[i, K]; pd_kmeans_reduce then adds up, for every image b that is not done, the slabs
 * block_range[b] = (first block, number of blocks) — the image's blocks are consecutive in `blocks` — into sums[b] / counts[b]
 * (overwritten).  ~170 workgroups per image flushing K*C same-address atomics each cost 60 us per iteration at config 4.
 */
int pd_kmeans_assign_partial(const float *X, const int32_t *blocks, int n_blocks, const float *centers, const float *cnorm,
                             const int32_t *done, int32_t *labels, float *partial_sums, float *partial_counts, int32_t *changed,
                             int C, int K, void *stream);
int pd_kmeans_reduce(const float *partial_sums, const float *partial_counts, const int32_t *block_range, const int32_t *done,
                     float *sums, float *counts, int B, int K, int C, void *stream);

/*
 * pd_kmeans_reduce and pd_kmeans_update in ONE launch (what the product issues per iteration after pd_kmeans_assign_partial):
 * centres, norms, iteration count and `done` as pd_kmeans_update would leave them, every floating-point sum in a fixed order
 * (bit-reproducible).  scratch: fp32, pd_kmeans_reduce_update_scratch_floats(B, K, C) elements; ticket: int32 [B], zero on
 * first use (left zero).
 */
int64_t pd_kmeans_reduce_update_scratch_floats(int B, int K, int C);
int pd_kmeans_reduce_update(const float *partial_sums, const float *partial_counts, const int32_t *block_range, float *centers,
                            float *cnorm, int32_t *changed, const float *tol, int32_t *done, int32_t *n_iter, float *scratch,
                            int32_t *ticket, int B, int K, int C, void *stream);

/*
 * The E-step with distance bounds (Hamerly 2010), exact: same labels / partial sums / counts / `changed` as pd_kmeans_assign_partial, but a point
 * whose bounds prove that its fp32 argmin is still its label is not read, and a slab without a changed label keeps the partial sums of an
 * earlier call.  State kept by the caller between the calls of one run: ub, lb, xnorm fp32 [N] (contents irrelevant while labels[n] < 0) and
 * cshift fp32 [B, 2, 8] = per image (how far centre k moved in the last update, the largest move of any OTHER centre), zero before the first
 * update, afterwards written by pd_kmeans_reduce_update_shift (pd_kmeans_reduce_update with that one more output).  `labels` must be -1 before
 * the first call of a run; partial_sums / partial_counts must be the buffers of the previous call.
 */
int pd_kmeans_assign_bounded(const float *X, const int32_t *blocks, int n_blocks, const float *centers, const float *cnorm, const int32_t *done,
                             int32_t *labels, float *partial_sums, float *partial_counts, int32_t *changed, float *ub, float *lb, float *xnorm,
                             const float *cshift, int C, int K, void *stream);
int pd_kmeans_reduce_update_shift(const float *partial_sums, const float *partial_counts, const int32_t *block_range, float *centers, float *cnorm,
                                  int32_t *changed, const float *tol, int32_t *done, int32_t *n_iter, float *scratch, int32_t *ticket, float *cshift,
                                  int B, int K, int C, void *stream);

/*
 * M-step + convergence for every image b with done[b] == 0:  centers[b,k] = sums / counts (unchanged when the cluster is
 * empty), cnorm recomputed, n_iter[b] += 1, done[b] = (changed[b] == 0) || (sum_k |new - old|^2 <= tol[b]); then sums,
 * counts and changed are cleared for the next pd_kmeans_assign.
 */
int pd_kmeans_update(float *centers, float *cnorm, float *sums, float *counts, int32_t *changed, const float *tol, int32_t *done,
                     int32_t *n_iter, int B, int K, int C, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PD_KMEANS_H */
