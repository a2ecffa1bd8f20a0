"""``DeviceProposalMapper``: the reference's ProposalDatasetMapper (data/dataset_mappers/proposal_dataset_mapper.py:24-235)
with the pixel work on the GPU (include/pd_input.h) — SURVEY §8 f3.

Host side (cheap, data-dependent control flow): the random draws of detectron2's RandomFlip / RandomCrop / ResizeScale /
FixedSizeCrop in the order the reference builds them (:64-88; restated from detectron2 0.6, which is not in this image),
the COCO RLE string -> run lengths parse, Pillow's coefficient tables for the region that survives the crops.
Device side: the Pillow-exact two-pass bilinear resample of the image with flip / crops / pad folded into the addressing,
and every pseudo-label mask sampled straight from its run lengths (decode + flip + crop + nearest resize + crop + pad in
one kernel, no dense full-resolution mask), plus the mask areas for the reference's area-ratio filter (:225-235).
The uploaded image is the decoded uint8 HWC array; the result is what the reference's mapper returns
({"image" [3,S,S] uint8, "padding_mask", "instances" (gt_masks BitMasks, gt_classes), "height", "width"}), on the device."""
import math

import numpy as np
import torch

from .. import lib as _lib
from ..compat import BitMasks, Instances
from ..utils import rle as _rle

PRECISION_BITS = 32 - 8 - 2


def resample_coeffs(in_size, out_size, first, count):
    """Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc (bilinear) for output indices [first, first + count),
    vectorised with the same double-precision operation order (the weight sum is a sequential cumsum)."""
    if in_size == out_size:                                   # Pillow skips the pass: identity taps
        idx = np.arange(first, first + count, dtype=np.int32)
        return idx, np.ones(count, dtype=np.int32), np.full((count, 1), 1 << PRECISION_BITS, dtype=np.int32)
    scale = filterscale = (float(in_size) - 0.0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = 0.0 + (np.arange(first, first + count, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.int64)[None, :]
    arg = np.abs(((x + xmin[:, None]) - center[:, None] + 0.5) * ss)
    w = np.where((x < xmax[:, None]) & (arg < 1.0), 1.0 - arg, 0.0)
    ww = np.cumsum(w, axis=1)[:, -1:]
    k = np.where(ww != 0.0, w / np.where(ww != 0.0, ww, 1.0), w)
    v = k * (1 << PRECISION_BITS)
    kk = np.where(v < 0, np.trunc(v - 0.5), np.trunc(v + 0.5)).astype(np.int32)
    return xmin.astype(np.int32), xmax.astype(np.int32), kk


def nearest_index(in_size, out_size):
    """Pillow NEAREST resize positions (Geometry.c ImagingScaleAffine): tabulated by repeated addition in double precision"""
    scale = float(in_size) / float(out_size)
    steps = np.full(out_size, scale, dtype=np.float64)
    steps[0] = 0.0 + scale * 0.5
    return np.clip(np.cumsum(steps).astype(np.int64), 0, in_size - 1).astype(np.int32)


class DeviceProposalMapper:
    def __init__(self, image_size, min_scale=0.1, max_scale=2.0, crop_type=None, crop_size=None, flip=True, min_area_ratio=0.0,
                 min_object_area_ratio=0.0, device="cuda", rng=None, pad_value=128, num_repeats=100):
        self.image_size, self.min_scale, self.max_scale = int(image_size), float(min_scale), float(max_scale)
        self.crop_type, self.crop_size, self.flip = crop_type, crop_size, flip
        self.min_area_ratio, self.min_object_area_ratio = min_area_ratio, min_object_area_ratio
        self.device, self.pad_value, self.num_repeats = torch.device(device), int(pad_value), num_repeats
        self.rng = rng if rng is not None else np.random                 # detectron2 draws from the global numpy RNG

    @classmethod
    def from_config(cls, cfg, is_train=True, device=None):
        names = list(cfg.CUSTOM_DATASETS.AUG_NAME_LIST)
        for n in names:
            if n not in ("flip", "crop", "scale"):
                raise NotImplementedError(f"augmentation '{n}' (reference :68-72: colour jitter / rotation) is not in the device pipeline")
        scale = "scale" in names
        return cls(cfg.INPUT.IMAGE_SIZE, cfg.INPUT.MIN_SCALE if scale else 1.0, cfg.INPUT.MAX_SCALE if scale else 1.0,
                   cfg.INPUT.CROP.TYPE if "crop" in names else None, tuple(cfg.INPUT.CROP.SIZE) if "crop" in names else None,
                   "flip" in names, cfg.PROPOSAL_LEARNING.MIN_AREA_RATIO, cfg.PROPOSAL_LEARNING.MIN_OBJECT_AREA_RATIO,
                   device or cfg.MODEL.DEVICE)

    # ------------------------------------------------------------------ host: parameter draws (detectron2 0.6 augmentation_impl.py)
    def draw(self, in_h, in_w, weak=False):
        rng, S = self.rng, self.image_size
        p = {"in_h": in_h, "in_w": in_w, "size": S, "flip": bool(self.flip and rng.uniform() < 0.5)}
        h, w = in_h, in_w
        p["crop1"] = (0, 0, w, h)
        if self.crop_type is not None and not weak:
            if self.crop_type == "relative":
                ch, cw = int(h * self.crop_size[0] + 0.5), int(w * self.crop_size[1] + 0.5)
            elif self.crop_type == "relative_range":
                cs = np.asarray(self.crop_size, dtype=np.float32)
                chf, cwf = cs + rng.rand(2) * (1 - cs)
                ch, cw = int(h * chf + 0.5), int(w * cwf + 0.5)
            elif self.crop_type == "absolute":
                ch, cw = min(self.crop_size[0], h), min(self.crop_size[1], w)
            else:
                raise NotImplementedError(self.crop_type)
            y0 = rng.randint(h - ch + 1)
            x0 = rng.randint(w - cw + 1)
            p["crop1"] = (int(x0), int(y0), int(cw), int(ch))
            h, w = ch, cw
        s = rng.uniform(1.0, 1.0) if weak else rng.uniform(self.min_scale, self.max_scale)
        out_scale = min(S * s / h, S * s / w)
        rh, rw = int(np.round(h * out_scale)), int(np.round(w * out_scale))
        p["resize"] = (rh, rw)
        off = np.round(np.multiply(np.maximum(np.array([rh, rw]) - S, 0), rng.uniform(0.0, 1.0))).astype(int)
        p["crop2"] = (int(off[1]), int(off[0]))
        return p

    # ------------------------------------------------------------------ device: pixels
    def _dev(self, a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device, non_blocking=True)

    def transform(self, image, segmentations, p):
        """image uint8 [H, W, 3] (numpy or tensor), segmentations = list of COCO RLE dicts {"size": [H, W], "counts": str}
        -> (image uint8 [3,S,S], masks bool [n,S,S], padding_mask bool [S,S], areas int32 [n]) on the device"""
        if self.device.type != "cuda":
            raise RuntimeError("the device input pipeline runs on the GPU only (no CPU fallback in partdistillation_amd)")
        L, S = _lib.load(), p["size"]
        st = _lib.current_stream()
        img = image if torch.is_tensor(image) else torch.from_numpy(np.ascontiguousarray(image))
        assert img.dtype == torch.uint8 and img.dim() == 3 and img.shape[2] == 3, img.shape
        img = img.to(self.device, non_blocking=True).contiguous()
        H, W = int(img.shape[0]), int(img.shape[1])
        (x0, y0, cw, ch), (rh, rw), (ox, oy), flip = p["crop1"], p["resize"], p["crop2"], int(p["flip"])
        vh, vw = min(rh - oy, S), min(rw - ox, S)
        # image: horizontal pass over the source rows the surviving output rows need, then vertical pass + crop + pad
        ymin, ycnt, ykk = resample_coeffs(ch, rh, oy, vh)
        xmin, xcnt, xkk = resample_coeffs(cw, rw, ox, vw)
        r0, r1 = int(ymin.min()), int((ymin + ycnt).max())
        tmp = torch.empty((r1 - r0, vw, 3), dtype=torch.uint8, device=self.device)
        out = torch.empty((3, S, S), dtype=torch.uint8, device=self.device)
        tabs = [self._dev(t) for t in (xmin, xcnt, xkk, ymin, ycnt, ykk)]
        _lib.check(L.pd_resample_rows_u8(img.data_ptr(), H, W, y0 + r0, r1 - r0, x0, flip, tabs[0].data_ptr(), tabs[1].data_ptr(),
                                         tabs[2].data_ptr(), xkk.shape[1], vw, tmp.data_ptr(), st))
        _lib.check(L.pd_resample_cols_u8(tmp.data_ptr(), r1 - r0, vw, r0, tabs[3].data_ptr(), tabs[4].data_ptr(), tabs[5].data_ptr(),
                                         ykk.shape[1], vh, vw, S, self.pad_value, out.data_ptr(), st))
        padding = torch.ones((S, S), dtype=torch.bool, device=self.device)
        padding[:vh, :vw] = False
        # masks: straight from the run lengths
        n = len(segmentations)
        masks = torch.empty((n, S, S), dtype=torch.uint8, device=self.device)
        area = torch.zeros(n, dtype=torch.int32, device=self.device)
        if n:
            starts, offsets = [], [0]
            for seg in segmentations:
                assert tuple(seg["size"]) == (H, W), (seg["size"], (H, W))
                counts = seg["counts"]
                counts = _rle.string_to_counts(counts) if isinstance(counts, (str, bytes)) else np.asarray(counts)
                cs = np.concatenate([[0], np.cumsum(counts[:-1])]) if len(counts) else np.zeros(1)
                starts.append(cs.astype(np.int32))
                offsets.append(offsets[-1] + len(cs))
            sx = (x0 + nearest_index(cw, rw)[ox:ox + vw]).astype(np.int32)
            sy = (y0 + nearest_index(ch, rh)[oy:oy + vh]).astype(np.int32)
            d = [self._dev(t) for t in (np.concatenate(starts), np.asarray(offsets, dtype=np.int32), sx, sy)]
            _lib.check(L.pd_rle_sample_u8(d[0].data_ptr(), d[1].data_ptr(), n, H, W, flip, d[2].data_ptr(), d[3].data_ptr(), vh, vw, S,
                                          masks.data_ptr(), area.data_ptr(), st))
        return out, masks.view(torch.bool) if n else masks.bool(), padding, area

    def select(self, masks, area):
        """reference :217-235: drop empty masks (filter_empty_instances(by_box=False)), then masks whose share of the total
        mask area is <= min_area_ratio -> indices kept"""
        a = area.float()
        nonempty = (area > 0).nonzero().flatten()
        if nonempty.numel() == 0:
            return nonempty
        ratio = a[nonempty] / a[nonempty].sum()
        return nonempty[ratio > self.min_area_ratio]

    def __call__(self, dataset_dict):
        """dataset_dict: {"image": decoded uint8 HWC array (or "file_name" readable by Pillow), "pseudo_annotations":
        [{"segmentation": COCO RLE dict, "category_id"?}], ...} -> the reference mapper's output dict, tensors on the device"""
        image = dataset_dict.get("image")
        if image is None:
            from PIL import Image
            image = np.asarray(Image.open(dataset_dict["file_name"]).convert("RGB"))
        annos = dataset_dict["pseudo_annotations"]
        segs = [a["segmentation"] for a in annos]
        classes = torch.tensor([a.get("category_id", -1) for a in annos], dtype=torch.int64)
        H, W = int(image.shape[0]), int(image.shape[1])
        for attempt in range(self.num_repeats + 1):
            p = self.draw(H, W, weak=attempt == self.num_repeats)            # last resort: the weak augmentation (:160-164)
            img, masks, padding, area = self.transform(image, segs, p)
            keep = self.select(masks, area)
            if keep.numel() > 0 or attempt == self.num_repeats:
                break
        inst = Instances((self.image_size, self.image_size))
        inst.gt_masks = BitMasks(masks[keep])
        inst.gt_classes = classes.to(self.device)[keep]
        out = {k: v for k, v in dataset_dict.items() if k not in ("pseudo_annotations", "image")}
        out.update(image=img, padding_mask=padding, instances=inst, height=self.image_size, width=self.image_size)
        return out
