"""ORACLE (test infrastructure, never imported by the product): CPU restatement of the training input pipeline the
reference runs per image in its dataloader workers — data/dataset_mappers/proposal_dataset_mapper.py:171-235
(`_forward`, `_transform_annotations`) with the augmentation list of its scripts (`["crop","scale","flip"]`,
sh_files/proposal_learning/train_multi.sh:7 -> :64-88: RandomFlip, RandomCrop, ResizeScale(MIN_SCALE..MAX_SCALE ->
IMAGE_SIZE), FixedSizeCrop(IMAGE_SIZE, pad 128)) and the on-disk pseudo-label format (COCO RLE dicts,
proposal_generation_model.py:188-199).

Third-party pieces that are ABSENT from /root/reference and from this image:
  * detectron2 0.6 `data/transforms` (RandomFlip / RandomCrop / ResizeScale / FixedSizeCrop parameter draws and their
    Transform composition) — restated from its published source in `draw_params` / `apply`; PARITY UNPINNED for the
    draws (numpy RNG call order), the geometry is pinned indirectly through Pillow below.
  * pycocotools `mask.decode` — the published RLE format (maskApi.c rleFrString / rleDecode) restated in
    `rle_string_to_counts` / `rle_decode`; pinned by round trips against the product's own encoder, which is pinned to
    the reference's golden in tests/test_oracle_propgen.py.
PRESENT and used as the pin: Pillow — detectron2's ResizeTransform resizes uint8 images with
`Image.fromarray(img).resize((w, h), Image.BILINEAR)` and segmentation masks with Image.NEAREST;
`resize_bilinear_u8` / `resize_nearest` restate Pillow's ImagingResample (8-bit fixed-point, horizontal then vertical pass)
and nearest sampling and are checked bit-exact against Pillow itself (tests/test_input_pipeline.py)."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


# ------------------------------------------------------------------------------------------------ COCO RLE
def rle_string_to_counts(s):
    """maskApi.c rleFrString: 5 data bits + continuation bit per char (offset 48), sign extension, delta coding from the
    third count on"""
    if isinstance(s, bytes):
        s = s.decode("ascii")
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return np.asarray(counts, dtype=np.int64)


def rle_decode(counts, h, w):
    """maskApi.c rleDecode: runs alternate 0 / 1 starting with 0, pixels in column-major order -> bool [h, w]"""
    flat = np.zeros(h * w, dtype=bool)
    pos, v = 0, False
    for c in counts:
        if v:
            flat[pos:pos + c] = True
        pos += int(c)
        v = not v
    return flat.reshape(w, h).T


# ------------------------------------------------------------------------------------------------ Pillow resampling
def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def resample_coeffs(in_size, out_size, box0=0.0, box1=None):
    """Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter (support 1):
    -> (xmin [out], count [out], kk int32 [out, ksize])"""
    box1 = float(in_size) if box1 is None else box1
    scale = filterscale = (box1 - box0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmin_a, cnt_a = np.zeros(out_size, dtype=np.int32), np.zeros(out_size, dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = box0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            w = _bilinear((x + xmin - center + 0.5) * ss)
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        for x in range(ksize):                              # normalize_coeffs_8bpc: round half away from zero
            v = k[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(v - 0.5) if v < 0 else int(v + 0.5)
        xmin_a[xx], cnt_a[xx] = xmin, xmax
    return xmin_a, cnt_a, kk


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img, out_h, out_w):
    """Pillow Image.resize((out_w, out_h), BILINEAR) of a uint8 [H, W, C] array: horizontal pass into an 8-bit image
    (only when the width changes), then the vertical pass (only when the height changes)"""
    H, W, C = img.shape
    cur = img
    if out_w != W:
        xmin, cnt, kk = resample_coeffs(W, out_w)
        tmp = np.zeros((H, out_w, C), dtype=np.uint8)
        for xx in range(out_w):
            acc = np.full((H, C), 1 << (PRECISION_BITS - 1), dtype=np.int64)
            for x in range(cnt[xx]):
                acc += cur[:, xmin[xx] + x, :].astype(np.int64) * int(kk[xx, x])
            tmp[:, xx, :] = _clip8(acc)
        cur = tmp
    if out_h != H:
        ymin, cnt, kk = resample_coeffs(H, out_h)
        tmp = np.zeros((out_h, cur.shape[1], C), dtype=np.uint8)
        for yy in range(out_h):
            acc = np.full((cur.shape[1], C), 1 << (PRECISION_BITS - 1), dtype=np.int64)
            for y in range(cnt[yy]):
                acc += cur[ymin[yy] + y, :, :].astype(np.int64) * int(kk[yy, y])
            tmp[yy, :, :] = _clip8(acc)
        cur = tmp
    return cur


def nearest_index(in_size, out_size):
    """Pillow NEAREST resize (Geometry.c ImagingScaleAffine): source index of every output index.  Pillow tabulates the
    positions by REPEATED ADDITION in double precision (xo = 0.5 * scale; xo += scale), and that rounding is part of the
    result; positions outside the source keep the output's initial 0 (cannot happen for a plain resize)."""
    scale = float(in_size) / float(out_size)
    idx = np.zeros(out_size, dtype=np.int64)
    xo = 0.0 + scale * 0.5
    for x in range(out_size):
        idx[x] = min(max(int(xo), 0), in_size - 1)
        xo += scale
    return idx


def resize_nearest(mask, out_h, out_w):
    return mask[nearest_index(mask.shape[0], out_h)][:, nearest_index(mask.shape[1], out_w)]


# ------------------------------------------------------------------------------------------------ detectron2 transforms
def draw_params(rng, in_h, in_w, image_size, min_scale, max_scale, crop_type=None, crop_size=None, flip_prob=0.5, flip=True):
    """parameter draws of detectron2 0.6 for the list [RandomFlip, RandomCrop, ResizeScale, FixedSizeCrop] in that order
    (augmentation_impl.py: RandomFlip.get_transform, RandomCrop.get_transform / get_crop_size, ResizeScale._get_resize,
    FixedSizeCrop._get_crop / _get_pad).  `rng` = numpy RandomState standing in for the global np.random."""
    p = {"in_h": in_h, "in_w": in_w, "size": image_size}
    p["flip"] = bool(flip and rng.uniform() < flip_prob)
    h, w = in_h, in_w
    p["crop1"] = (0, 0, w, h)                                             # x0, y0, w, h in the (flipped) input
    if crop_type is not None:
        if crop_type == "relative":
            ch, cw = int(h * crop_size[0] + 0.5), int(w * crop_size[1] + 0.5)
        elif crop_type == "relative_range":
            cs = np.asarray(crop_size, dtype=np.float32)
            chf, cwf = cs + rng.rand(2) * (1 - cs)
            ch, cw = int(h * chf + 0.5), int(w * cwf + 0.5)
        elif crop_type == "absolute":
            ch, cw = min(crop_size[0], h), min(crop_size[1], w)
        else:
            raise NotImplementedError(crop_type)
        y0 = rng.randint(h - ch + 1)
        x0 = rng.randint(w - cw + 1)
        p["crop1"] = (int(x0), int(y0), int(cw), int(ch))
        h, w = ch, cw
    s = rng.uniform(min_scale, max_scale)
    out_scale = min(image_size * s / h, image_size * s / w)
    rh, rw = int(np.round(h * out_scale)), int(np.round(w * out_scale))
    p["resize"] = (rh, rw)
    max_off = np.maximum(np.array([rh, rw]) - image_size, 0)
    off = np.round(np.multiply(max_off, rng.uniform(0.0, 1.0))).astype(int)
    p["crop2"] = (int(off[1]), int(off[0]))                               # x, y offset into the resized image
    return p


def apply(image, masks, p, pad_value=128):
    """image uint8 [H, W, 3], masks bool [n, H, W] -> (image uint8 [S, S, 3], masks bool [n, S, S], padding_mask bool [S, S])
    through HFlipTransform, CropTransform, ResizeTransform (Pillow), CropTransform, PadTransform (right / bottom)"""
    S = p["size"]
    if p["flip"]:
        image, masks = image[:, ::-1], masks[:, :, ::-1]
    x0, y0, cw, ch = p["crop1"]
    image, masks = image[y0:y0 + ch, x0:x0 + cw], masks[:, y0:y0 + ch, x0:x0 + cw]
    rh, rw = p["resize"]
    image = resize_bilinear_u8(np.ascontiguousarray(image), rh, rw)
    masks = np.stack([resize_nearest(m, rh, rw) for m in masks]) if len(masks) else np.zeros((0, rh, rw), dtype=bool)
    ox, oy = p["crop2"]
    vh, vw = min(rh - oy, S), min(rw - ox, S)
    out = np.full((S, S, 3), pad_value, dtype=np.uint8)
    out[:vh, :vw] = image[oy:oy + vh, ox:ox + vw]
    om = np.zeros((len(masks), S, S), dtype=bool)
    om[:, :vh, :vw] = masks[:, oy:oy + vh, ox:ox + vw]
    padding = np.ones((S, S), dtype=bool)
    padding[:vh, :vw] = False
    return out, om, padding


def filter_instances(masks, min_area_ratio):
    """proposal_dataset_mapper.py:225-235 after filter_empty_instances(by_box=False): non-empty masks whose share of the
    total mask area exceeds min_area_ratio -> kept indices"""
    area = masks.reshape(len(masks), -1).sum(1).astype(np.float64)
    nonempty = np.nonzero(area > 0)[0]
    if len(nonempty) == 0:
        return nonempty
    a32 = area[nonempty].astype(np.float32)
    ratio = a32 / np.float32(a32.sum())
    return nonempty[ratio > min_area_ratio]
