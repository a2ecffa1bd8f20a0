/*
 * pd_criterion.h — C-ABI of the matcher / criterion kernels of libpd_hip.so: the assignment solver, and the arithmetic between the
 * sampled point logits and the losses of modeling/criterion.py / modeling/matcher.py, which the reference (and rounds 1-3 here) run
 * as chains of eager elementwise / reduction / top-k launches over small tensors.
 *
 *   pd_lsa_batched            scipy.optimize.linear_sum_assignment + the cost-ordered pair sort, reference matcher.py:159-163
 *                             (there: C.cpu() -> SciPy -> topk; one D2H sync per image per decoder layer).
 *   pd_matcher_costs          the Hungarian cost matrix of every (image, head) problem in one pass over the point logits —
 *                             reference matcher.py:108-158 (batch_sigmoid_ce_loss_jit :38-62, batch_dice_loss_jit :13-35, cost_class
 *                             :122, the weighted sum :150-154).  Replaces softplus / sigmoid / two row sums, two batched GEMMs with
 *                             n_targets output columns, a softmax gather and ~20 elementwise launches; the fp32 copies of the
 *                             logits and their sigmoids (2 x 100 MB at BASELINE config 2) are never written.
 *   pd_match_point_logits     the logits of all Q masks at the matcher's points straight from the mask features (matcher.py:108-125):
 *                             sampler + batched product in one kernel, the sampled features stay in LDS.
 *   pd_mask_point_losses_*    sigmoid_ce_loss (criterion.py:50-69) and dice_loss (:25-47) of the matched masks at their sampled
 *                             points, per mask, and their gradient with respect to the point logits.
 *   pd_point_sample_u8        the target masks at the matcher's / the loss's points, read as stored bytes.
 *   pd_uncertain_points       get_uncertain_point_coords_with_randomness (criterion.py:181-189, detectron2 point_rend): of K
 *                             oversampled points per mask keep the k with the smallest |logit| (calculate_uncertainty :72-88 is
 *                             -|logit|), followed by the mask's random points — replaces abs, neg, a multi-block top-k, a gather
 *                             and a concatenation.
 *
 * Device pointers, fp32 unless stated; `stream` = hipStream_t; returns 0 or a negative PD_ERR_* (pd_msda.h).
 */
#ifndef PD_CRITERION_H
#define PD_CRITERION_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Solve `nbatch` independent rectangular assignment problems (minimise).
 *   cost      float32 [nbatch, nrows, ncols_max] row-major; problem b uses columns [0, ncols[b])
 *   ncols     int32   [nbatch]   (0 <= ncols[b] <= ncols_max)
 *   out_rows  int64   [nbatch, ncols_max]  selected row  (query)  of pair k, pairs sorted by ascending cost
 *   out_cols  int64   [nbatch, ncols_max]  selected col  (target) of pair k; entries k >= min(nrows, ncols[b]) are -1
 * The solver is the float64 shortest-augmenting-path algorithm SciPy uses
 * (Crouse 2016), including its tie-breaking scan order, so that the result is
 * SciPy's for the same matrix.  Limits: nrows <= 4096, ncols_max <= 64.
 */
int pd_lsa_batched(const float *cost, const int32_t *ncols, int64_t *out_rows, int64_t *out_cols,
                   int nbatch, int nrows, int ncols_max, void *stream);

/*
 * Problem p = b * heads + d (image b, decoder output d) has Q queries and n_targets (padded) target columns.
 *   x        [problems, Q, n] point logits of the queries, dtype PD_F32 / PD_BF16 (converted to fp32 on load)
 *   t        target masks sampled at the same points: t[b * t_image_stride + d * t_head_stride + j * t_target_stride + i], i < n
 *            (so the sampler's [B, n_targets, heads * n] output is read in place)
 *   prob     [problems, Q, classes] class probabilities (softmax / sigmoid of the logits), labels int64 [B, n_targets]
 *   cost     [problems, Q, n_targets] =
 *              w_mask  * (sum_i softplus(x_i) - sum_i x_i t_ji) / n
 *            + w_dice  * (1 - (2 sum_i sigmoid(x_i) t_ji + 1) / (sum_i sigmoid(x_i) + sum_i t_ji + 1))
 *            - w_class * prob[label_j]
 * softplus as torch's (beta 1, threshold 20).  Padded columns (all-zero targets, label 0) get finite costs nobody reads.
 */
int pd_matcher_costs(const void *x, int dtype, const float *t, int64_t t_image_stride, int64_t t_head_stride, int64_t t_target_stride,
                     const float *prob, const int64_t *labels, float *cost, int problems, int heads, int Q, int n, int n_targets,
                     int classes, float w_mask, float w_class, float w_dice, void *stream);

/*
 * The matcher's point logits of every (image, head) problem in ONE launch, without the sampled features in memory (reference matcher.py:108-125:
 * point_sample(out_mask, point_coords) with out_mask = mask_embed . mask_features — bilinear sampling is linear in the map, so the logits are
 * mask_embed . point_sample(mask_features)):
 *   feat_nhwc [B, H, W, C] fp32 mask features (channels last), coords [B, heads * points, 2] (x, y) in [0, 1] — problem p = b * heads + d owns
 *   points [d * points, (d + 1) * points) of image b —, emb [B * heads, Q, C] bf16  ->  out [B * heads, Q, points] bf16
 *   out[p, q, i] = bf16( sum_c emb[p, q, c] * bf16( grid_sample(feat[b], coords[p, i])[c] ) )      (bilinear, zeros padding, align_corners = False)
 * — bit-identical to pd_point_sample_nhwc_f32_bf16 followed by pd_sgemm_tn_batched_bf16.  C == 256, Q <= 128, points % 4 == 0.
 */
int pd_match_point_logits(const float *feat_nhwc, const float *coords, const void *emb, void *out, int B, int heads, int Q, int points,
                          int H, int W, int C, void *stream);

/*
 * x, y [rows, n] (point logits, sampled target masks in [0, 1]) ->
 *   bce[r]  = mean_i ( max(x, 0) - x y + log1p(exp(-|x|)) )                       (F.binary_cross_entropy_with_logits, mean over points)
 *   dice[r] = 1 - (2 sum_i s_i y_i + 1) / (sum_i s_i + sum_i y_i + 1),  s = sigmoid(x)
 *   stats   [rows, 3] = (sum s y, sum s, sum y): what the backward needs besides x and y
 */
int pd_mask_point_losses_fwd(const float *x, const float *y, float *bce, float *dice, float *stats, int rows, int n, void *stream);
/* dx[r, i] = d_bce[r] * (s_i - y_i) / n - d_dice[r] * (2 y_i den - num) / den^2 * s_i (1 - s_i),  num = 2 sum s y + 1, den = sum s + sum y + 1 */
int pd_mask_point_losses_bwd(const float *x, const float *y, const float *stats, const float *d_bce, const float *d_dice, float *dx,
                             int rows, int n, void *stream);

/*
 * logits [rows, K], coords [rows, K, 2], random_coords [rows, n_random, 2] (nullable when n_random = 0) ->
 * out [rows, k + n_random, 2]: the coordinates of the k points with the smallest |logit| — every point below the k-th smallest value,
 * then as many of the points equal to it as are needed —, then the row's random coordinates.  Which ties are taken and the order of the
 * k points are fixed by the kernel's thread layout (deterministic; the losses are sums over the points).  1 <= k <= K <= PD_UNCERTAIN_MAX_K.
 */
#define PD_UNCERTAIN_MAX_K 40960
int pd_uncertain_points(const float *logits, const float *coords, const float *random_coords, float *out, int rows, int K, int k,
                        int n_random, void *stream);

/*
 * Bilinear samples (F.grid_sample: bilinear, zeros padding, align_corners=False — detectron2 point_sample) of one-byte 0 / non-0 masks
 * maps [M, H, W] as they are stored (the reference samples `.float()` copies: matcher.py:130-139, criterion.py:196-199):
 *   out[r, p] = sample of map map_idx[r] (row r when map_idx is NULL) at coords[r / coords_div, p]   (coords [ceil(rows / coords_div), P, 2] in [0, 1])
 */
int pd_point_sample_u8(const uint8_t *maps, const int64_t *map_idx, const float *coords, float *out, int rows, int P, int H, int W,
                       int coords_div, void *stream);

/*
 * The criterion's three loss vectors [3, H] (rows: loss_ce, loss_mask, loss_dice; column h = head in criterion order: 0 = final output,
 * 1.. = aux_outputs[h - 1]) in one launch, and their gradient in one:
 *   vec[0, h] = sum_{b, q} w[t] (logsumexp(x) - x[t]) / sum_{b, q} w[t],  x = logits[b, d_of_h[h], q, :], t = tclass[b, d_of_h[h], q]
 *               (F.cross_entropy(src_logits.transpose(1, 2), target_classes, empty_weight), criterion.py:126-145)
 *   vec[1, h] = sum_n bce[h, n] / num_masks,  vec[2, h] = sum_n dice[h, n] / num_masks        (criterion.py:203-206; bce / dice [H, Nh])
 *   logits: element (b, d, q, k) at b * image_stride + d * head_stride + q * K1 + k (fp32); tclass int64 [B, H, Q] with d in the middle;
 *   class_weight [K1]; d_of_h int64 [H]; num_masks: device scalar.  lse [B, H, Q] and den [H] (indexed by d) are what the backward keeps.
 *   backward: d_logits (same strides as logits), d_bce, d_dice [H, Nh] from dvec [3, H].
 */
int pd_loss_vectors_fwd(const float *logits, int64_t image_stride, int64_t head_stride, const int64_t *tclass, const float *class_weight,
                        const int64_t *d_of_h, const float *bce, const float *dice, const float *num_masks, float *vec, float *lse, float *den, int B,
                        int H, int Q, int K1, int Nh, void *stream);
int pd_loss_vectors_bwd(const float *logits, int64_t image_stride, int64_t head_stride, const int64_t *tclass, const float *class_weight,
                        const int64_t *d_of_h, const float *num_masks, const float *lse, const float *den, const float *dvec, float *d_logits,
                        float *d_bce, float *d_dice, int B, int H, int Q, int K1, int Nh, void *stream);

/*
 * Mask logits of the matched (query, target) pairs and their gradients: the rows the criterion keeps of
 * einsum("bqc,bchw->bqhw", mask_embed, mask_features) (mask2former_transformer_decoder.py:441-459; criterion.py:147-160 src_masks = pred_masks[src_idx]).
 *   tok        [B, T, C]  channels-last mask features as tokens (T = h w), fp32, C == PD_PAIR_LOGITS_CHANNELS
 *   e          [N, C]     the pairs' mask embeddings, fp32, grouped by image: image b owns rows [img_start[b], img_start[b + 1])
 *   img_start  HOST int32 [B + 1], img_start[0] = 0, img_start[B] = N, B <= PD_PAIR_LOGITS_MAX_IMAGES
 *   out_row    int64 [N] (device) row of out / g that pair i of e is written to / read from (a permutation; NULL: row i)
 *   out, g     [N, T]     out[out_row[i], t] = sum_c e[i, c] tok[b(i), t, c]
 *   d_tok      [B, T, C]  d_tok[b, t, c] = sum_{i in image b} g[out_row[i], t] e[i, c]   (zeros for an image without pairs)
 *   d_e        [N, C]     d_e[i, c] = sum_t g[out_row[i], t] tok[b(i), t, c]; workspace: pd_pair_logits_workspace_floats(T, C, N) floats (partial sums
 *                         per slab of tokens, added in slab order: deterministic)
 * Arithmetic: fp32 products, fp32 accumulation (v_mfma_f32_16x16x4_f32).  All pointers 16-byte aligned.
 */
#define PD_PAIR_LOGITS_CHANNELS 256
#define PD_PAIR_LOGITS_MAX_IMAGES 32
int pd_pair_logits_fwd(const float *tok, const float *e, const int32_t *img_start, const int64_t *out_row, float *out, int B, int T, int C, int N,
                       void *stream);
int pd_pair_logits_bwd_tok(const float *g, const float *e, const int32_t *img_start, const int64_t *out_row, float *d_tok, int B, int T, int C, int N,
                           void *stream);
int64_t pd_pair_logits_workspace_floats(int T, int C, int N);
int pd_pair_logits_bwd_rows(const float *g, const float *tok, const int32_t *img_start, const int64_t *out_row, float *d_e, float *workspace, int B,
                            int T, int C, int N, void *stream);

#ifdef __cplusplus
}
#endif
#endif
