/*
 * pd_grouping.h — C-ABI of the pixel-grouping (part-proposal generation) kernel of libpd_hip.so.
 *
 * Reference: proposal_generation_model.py:131-146 + 224-237 (twin: pixel_grouping_model.py:183-218).  The reference
 * upsamples the C-channel backbone features of an image to full resolution ([C, H, W] fp32: 4.8-6.4 GB per 1024^2 image),
 * gathers the object's pixels, ships them to the CPU and takes argmax_k of `feature . centroid_k` (metric "dot") or of
 * `2 feature . centroid_k - |feature|^2 - |centroid_k|^2` (metric "l2").  Bilinear interpolation is linear with weights
 * that sum to one, so the per-centroid scores  s_k = F . c_k  (dot)  /  2 F . c_k - |c_k|^2  (l2; the -|feature|^2 term
 * is the same for every k) can be formed at feature resolution ([K, h, w], K = 4) and interpolated instead:
 * ~1 MB of label map per image instead of gigabytes of features.
 */
#ifndef PD_GROUPING_H
#define PD_GROUPING_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * labels[y, x] (uint8, [H, W]) = mask[y, x] ? 1 + argmax_k bilinear(scores_k)(y, x) : 0          (first maximum wins)
 * scores: fp32 [K, h, w] of one image (K <= 32); bilinear = F.interpolate(size=(Hp, Wp), mode="bilinear",
 * align_corners=False) evaluated on the top-left H x W crop (H <= Hp, W <= Wp: the un-padded image);
 * mask: uint8 [H, W] (nonzero = object pixel).
 */
int pd_scores_argmax_u8(const float *scores, const uint8_t *mask, uint8_t *labels, int K, int h, int w, int Hp, int Wp, int H,
                        int W, void *stream);

/*
 * Per-pixel assignment of the K selected query masks at inference (reference proposal_model.py:220-302:
 * F.interpolate of [Q, H/4, W/4] logits to the padded image size, `* object mask` (:372-378), `_unique_assignment`
 * :263-299).  The reference materialises [Q, H, W] fp32 three times (0.4 GB per image and copy at 1024^2, Q = 100);
 * here the low-resolution logits (26 MB) are interpolated inside the one pass that consumes them:
 *   v_k   = bilinear(logits_k)(y, x) * (object ? object[y, x] != 0 : 1)
 *   arg   = argmax_k scores[k] * sigmoid(v_k)   (first maximum)              int16 [H, W]
 *   obj   = max_k v_k > 0                                                     uint8 [H, W]
 *   positive[k] += #pixels with v_k > 0         (int32 [K], caller zeroes)
 * logits fp32 [K, h, w]; bilinear as in pd_scores_argmax_u8 (upsample to (Hp, Wp), top-left H x W crop).
 */
int pd_mask_assign(const float *logits, const float *scores, const uint8_t *object, int16_t *arg, uint8_t *obj, int32_t *positive,
                   int K, int h, int w, int Hp, int Wp, int H, int W, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PD_GROUPING_H */
