"""Pixel-grouping part-proposal generation (reference proposal_generation_model.py:28-237; SURVEY §8f-1, BASELINE
config 4): backbone features of the object's pixels are clustered into K = 4 groups per image and every object pixel
of the full-resolution image is labelled with its nearest centroid; the per-label masks are stored as COCO RLEs and
become the pseudo part labels the proposal model trains on.

Same class name / registry / config keys / result dictionary as the reference.  What is done differently:
  * K-means runs on the device (functions/kmeans.py, sklearn Lloyd semantics) instead of sklearn on the CPU;
  * the reference upsamples the C-channel features to full resolution (4.8-6.4 GB per 1024^2 image), moves the object's
    rows to the CPU and takes the arg-max of their products with the centroids.  Bilinear interpolation is linear, so the
    K score maps are formed at feature resolution and pd_scores_argmax_u8 interpolates THEM while writing the uint8
    label map (include/pd_grouping.h): ~1 MB per image instead of gigabytes;
  * per-label binary masks are never stacked: the RLEs are produced from the label map."""
import os
from typing import List, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import lib as _lib
from .compat import META_ARCH_REGISTRY, ImageList, build_backbone, configurable
from .functions.kmeans import kmeans_lloyd_batched
from .utils import rle


def sem_seg_postprocess(result, img_size, output_height, output_width):
    """detectron2.modeling.postprocessing.sem_seg_postprocess: crop the padding, resize to the original resolution."""
    result = result[:, : img_size[0], : img_size[1]].expand(1, -1, -1, -1)
    return F.interpolate(result, size=(output_height, output_width), mode="bilinear", align_corners=False)[0]


@META_ARCH_REGISTRY.register()
class ProposalGenerationModel(nn.Module):
    @configurable
    def __init__(self, *, backbone, size_divisibility: int, dataset_name: str, pixel_mean: Tuple[float], pixel_std: Tuple[float],
                 distance_metric: str = "l2", backbone_feature_key_list: List[str] = ("res4",), num_superpixel_clusters: int = 4,
                 feature_normalize: bool = False, wandb_vis_period: int = 100, debug: bool = False, save_path: str = None):
        super().__init__()
        assert distance_metric in ("dot", "l2")
        self.backbone = backbone
        if size_divisibility < 0:
            size_divisibility = self.backbone.size_divisibility
        self.size_divisibility = size_divisibility
        self.register_buffer("pixel_mean", torch.Tensor(pixel_mean).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.Tensor(pixel_std).view(-1, 1, 1), False)
        self.distance_metric = distance_metric
        self.backbone_feature_key_list = list(backbone_feature_key_list)
        self.num_superpixel_clusters = num_superpixel_clusters
        self.feature_normalize = feature_normalize
        self.dataset_name, self.wandb_vis_period, self.debug = dataset_name, wandb_vis_period, debug
        self.root_save_path = save_path            # the reference takes it from the dataset's metadata (:63-72)
        self.num_test_iterations = 0
        self.kmeans_generator = None               # torch.Generator for the k-means++ seeding (None = global device RNG)
        self.init_centroids = None                 # test hook: callable(image index, centred=False) -> [K,C] initial centres

    @classmethod
    def from_config(cls, cfg):
        pg = cfg.PROPOSAL_GENERATION
        return {"backbone": build_backbone(cfg), "size_divisibility": cfg.MODEL.MASK_FORMER.SIZE_DIVISIBILITY,
                "dataset_name": pg.DATASET_NAME, "pixel_mean": cfg.MODEL.PIXEL_MEAN, "pixel_std": cfg.MODEL.PIXEL_STD,
                "distance_metric": pg.DISTANCE_METRIC, "backbone_feature_key_list": pg.BACKBONE_FEATURE_KEY_LIST,
                "num_superpixel_clusters": pg.NUM_SUPERPIXEL_CLUSTERS, "feature_normalize": pg.FEATURE_NORMALIZE,
                "wandb_vis_period": cfg.WANDB.VIS_PERIOD_TEST, "debug": pg.DEBUG}

    @property
    def device(self):
        return self.pixel_mean.device

    # ------------------------------------------------------------------ pieces (reference :100-127)
    def prepare_mask(self, inputs, images):
        h_pad, w_pad = images.tensor.shape[-2:]
        out = []
        for x in inputs:
            gt = x["instances"].to(self.device).gt_masks.tensor
            padded = torch.zeros((gt.shape[0], h_pad, w_pad), dtype=gt.dtype, device=gt.device)
            padded[:, : gt.shape[1], : gt.shape[2]] = gt
            out.append({"masks": padded})
        return out

    def _prepare_features(self, features):
        keys = self.backbone_feature_key_list
        H, W = features[keys[0]].shape[-2:]
        feat = torch.cat([F.interpolate(features[k].float(), size=(H, W), mode="bilinear", align_corners=False) for k in keys], dim=1)
        return F.normalize(feat, dim=1, p=2) if self.feature_normalize else feat

    def _scores(self, feat, centroids):
        """[K,h,w] score maps whose bilinear interpolation has the same arg-max over k as the reference's
        `_measure_distance(interpolated features, centroids)` (:214-218)."""
        C, h, w = feat.shape
        s = centroids @ feat.reshape(C, h * w)                                         # [K, hw]
        if self.distance_metric == "l2":
            s = 2.0 * s - (centroids * centroids).sum(1)[:, None]
        return s.reshape(-1, h, w).float().contiguous()

    def _label_map(self, scores, object_mask_resized, pad_hw, image_size, height, width):
        """uint8 [height, width]: 1 + arg-max centroid on the object's pixels, 0 elsewhere."""
        K, h, w = scores.shape
        Hp, Wp = pad_hw
        assert scores.dtype == torch.float32 and scores.is_contiguous()
        if (height, width) == tuple(image_size) and scores.is_cuda:
            m8 = object_mask_resized.to(torch.uint8).contiguous()
            labels = torch.empty((height, width), dtype=torch.uint8, device=scores.device)
            _lib.check(_lib.load().pd_scores_argmax_u8(scores.data_ptr(), m8.data_ptr(), labels.data_ptr(), K, h, w, Hp, Wp, height,
                                                       width, _lib.current_stream()))
            return labels
        if not scores.is_cuda:
            raise RuntimeError("pd_scores_argmax_u8 runs on the GPU only (no CPU fallback in partdistillation_amd)")
        # the image was resized by the data pipeline: two chained interpolations of the K (not C) maps
        up = F.interpolate(scores[None], size=(Hp, Wp), mode="bilinear", align_corners=False)[0]
        up = sem_seg_postprocess(up, image_size, height, width)
        return torch.where(object_mask_resized, up.argmax(0).to(torch.uint8) + 1, torch.zeros((), dtype=torch.uint8, device=up.device))

    # ------------------------------------------------------------------ forward (reference :131-181)
    @torch.no_grad()
    def forward(self, batched_inputs):
        assert not self.training, "proposal generation is eval-only."
        images = [(x["image"].to(self.device) - self.pixel_mean) / self.pixel_std for x in batched_inputs]
        images = ImageList.from_tensors(images, self.size_divisibility)
        targets = self.prepare_mask(batched_inputs, images)
        backbone_out = self.backbone(images.tensor)               # may run under the caller's autocast
        with torch.autocast(device_type=self.device.type, enabled=False):
            return self._group(batched_inputs, images, targets, self._prepare_features(backbone_out))

    def _group(self, batched_inputs, images, targets, features):
        """clustering + labelling, always fp32 (the kernels take fp32 score maps)"""
        pad_hw = tuple(images.tensor.shape[-2:])
        K = self.num_superpixel_clusters
        prep, datas, inits = [], [], []
        for i, (inp, feat, image_size, tgt) in enumerate(zip(batched_inputs, features, images.image_sizes, targets)):
            height, width = inp.get("height", image_size[0]), inp.get("width", image_size[1])
            masks = tgt["masks"]
            mask_resized = sem_seg_postprocess(masks.float(), image_size, height, width)[0].bool()
            mask_low = F.interpolate(masks[None].float(), size=feat.shape[-2:], mode="nearest")[0, 0].bool()
            data = feat[:, mask_low].t().contiguous()                                      # [N, C] object pixels at 1/8 res
            ok = data.shape[0] > K
            prep.append((ok, mask_resized, height, width))
            if ok:
                datas.append(data)
                inits.append(self.init_centroids(i) if self.init_centroids is not None else None)
        results = [None] * len(prep)
        if datas:                                                                          # all images advance together
            centroids, n_iters = kmeans_lloyd_batched(datas, K, inits=inits, generator=self.kmeans_generator)
            j = 0
            for i, (ok, mask_resized, height, width) in enumerate(prep):
                if not ok:
                    continue
                labels = self._label_map(self._scores(features[i], centroids[j]), mask_resized, pad_hw, images.image_sizes[i], height, width)
                results[i] = self._result(batched_inputs[i], labels, mask_resized, centroids[j], n_iters[j])
                j += 1
        self.num_test_iterations += 1
        return results

    def _result(self, inp, labels, object_mask, centroids, n_iter):
        H, W = labels.shape
        counts = torch.bincount(labels.flatten().long(), minlength=self.num_superpixel_clusters + 1)
        # run-length form of the column-major label map on the device; only the run table crosses to the host — in ONE copy with the label
        # counts and the object's area (was five blocking reads per image)
        flat = labels.t().contiguous().flatten()
        starts = torch.cat([flat.new_zeros(1, dtype=torch.long), (flat[1:] != flat[:-1]).nonzero().flatten() + 1])
        lengths = torch.diff(starts, append=starts.new_tensor([flat.numel()]))
        packed = torch.cat([counts, object_mask.sum().reshape(1), flat[starts].long(), lengths]).cpu().numpy()
        nk = counts.numel()
        counts_h, area, runs = packed[:nk].tolist(), int(packed[nk]), packed[nk + 1:]
        values, lengths = runs[:runs.size // 2].astype("uint8"), runs[runs.size // 2:]
        present = [l for l in range(1, len(counts_h)) if counts_h[l] > 0]
        res = {"file_name": inp.get("file_name"), "file_path": inp.get("file_path"), "class_code": inp.get("class_code"),
               "class_name": inp.get("class_name"), "part_mask": rle.runs_to_coco_json(values, lengths, (H, W), present),
               "object_ratio": area / (H * W), "height": H, "width": W,
               "class_index": inp.get("gt_object_class")}
        if self.root_save_path is not None and res["class_code"] is not None and res["file_name"] is not None:
            d = os.path.join(self.root_save_path, res["class_code"])
            os.makedirs(d, exist_ok=True)
            torch.save(res, os.path.join(d, res["file_name"]))
        res.update({"labels": labels, "present_labels": present, "centroids": centroids, "kmeans_iterations": n_iter})
        return res

    @staticmethod
    def binary_masks(result):
        """the reference's `pseudo_label` tensor [P, H, W] bool (one mask per label present), built on demand."""
        return torch.stack([result["labels"] == l for l in result["present_labels"]])
