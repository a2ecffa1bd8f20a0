/*
 * pd_input.h — C-ABI of the device input pipeline of libpd_hip.so.
 *
 * Replaces the per-image CPU work of the reference's dataloader workers,
 *   data/dataset_mappers/proposal_dataset_mapper.py:171-235 (`_forward`, `_transform_annotations`):
 *   T.apply_transform_gens([RandomFlip, RandomCrop, ResizeScale, FixedSizeCrop]) on the image (detectron2 ResizeTransform =
 *   Pillow BILINEAR on uint8), transform_instance_annotations + annotations_to_instances on every pseudo-label
 *   (pycocotools RLE decode -> dense mask -> flip / crop / Pillow NEAREST resize / crop / pad -> BitMasks).
 * The host keeps what is host work: file decode, the random parameter draws, the RLE string -> run lengths parse and the
 * (tiny) Pillow coefficient tables; everything that touches pixels runs here:
 *
 *   pd_resample_rows_u8   horizontal pass of Pillow's 8-bit resample (Resample.c ImagingResampleHorizontal_8bpc):
 *                         tmp[r][x][c] = clip8((2^21 + sum_k src[row0 + r][col(xmin[x] + k)][c] * kk[x][k]) >> 22),
 *                         col(j) = x0 + j, mirrored (W - 1 - col) when `flip` — flip and first crop are just addressing
 *   pd_resample_cols_u8   vertical pass + second crop + pad + HWC -> CHW:
 *                         out[c][y][x] = y < vh && x < vw ? clip8((2^21 + sum_k tmp[ymin[y] + k - r0][x][c] * kk[y][k]) >> 22) : pad
 *   pd_rle_sample_u8      all masks of the image, straight from their run lengths (no dense full-resolution mask):
 *                         out[i][y][x] = inside ? parity(search(starts_i, colmajor(src_x[x], src_y[y]))) : 0, and
 *                         area[i] += popcount — src_x / src_y = the composed nearest-neighbour index tables
 *
 * src: uint8 [H, W, 3] (HWC, as decoded).  Tables int32.  kk: Pillow's 22-bit fixed-point coefficients [n, ksize].
 * starts: int32, for mask i the entries [offsets[i], offsets[i+1]) are the EXCLUSIVE prefix sums of its COCO run lengths
 * (column-major, first run = zeros).  `stream` = hipStream_t; returns 0 or PD_ERR_*.
 */
#ifndef PD_INPUT_H
#define PD_INPUT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int pd_resample_rows_u8(const uint8_t *src, int H, int W, int row0, int rows, int x0, int flip, const int32_t *xmin,
                        const int32_t *cnt, const int32_t *kk, int ksize, int out_w, uint8_t *tmp, void *stream);

int pd_resample_cols_u8(const uint8_t *tmp, int tmp_rows, int tmp_w, int r0, const int32_t *ymin, const int32_t *cnt,
                        const int32_t *kk, int ksize, int vh, int vw, int S, int pad_value, uint8_t *out, void *stream);

int pd_rle_sample_u8(const int32_t *starts, const int32_t *offsets, int n_masks, int H, int W, int flip, const int32_t *src_x,
                     const int32_t *src_y, int vh, int vw, int S, uint8_t *out, int32_t *area, void *stream);

/*
 * out [B, H, W, 3] fp32 (the channels-last storage of the [B, 3, H, W] batch) = (images[b] - mean) / std for B same-size planar uint8
 * images [3, H, W] — the model's preprocess (reference proposal_model.py:251-253 / part_distillation_model.py: `(x - pixel_mean) /
 * pixel_std` per image) in one launch.  images: HOST array of B device pointers; mean3 / std3: HOST arrays of 3 floats.
 */
#define PD_NORMALIZE_MAX_IMAGES 16
int pd_normalize_u8_nhwc(const uint8_t *const *images, int B, int H, int W, const float *mean3, const float *std3, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PD_INPUT_H */
