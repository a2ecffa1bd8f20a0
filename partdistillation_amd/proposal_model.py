"""``ProposalModel`` meta-architecture — training branch (reference
part_distillation/proposal_model.py:30-217, 305-338): normalise, pad-batch,
backbone, MaskFormer head, Hungarian set criterion, loss weighting.

Registered under the reference's name in ``META_ARCH_REGISTRY`` and built from
the same config keys (``from_config``).  The evaluation branch (:205-302, 340-430) lives in
``inference.py``; the wandb visualisation (:451-475) is out of scope."""
from typing import Tuple

import torch
from torch import nn

from .compat import META_ARCH_REGISTRY, ImageList, build_backbone, build_sem_seg_head, configurable
from .modeling.criterion import SetCriterion
from .modeling.matcher import HungarianMatcher


def build_criterion(cfg, num_classes, match_points=None, loss_points=None):
    mf = cfg.MODEL.MASK_FORMER
    matcher = HungarianMatcher(cost_class=mf.CLASS_WEIGHT, cost_mask=mf.MASK_WEIGHT, cost_dice=mf.DICE_WEIGHT,
                               num_points=match_points or mf.TRAIN_NUM_POINTS)
    weight_dict = {"loss_ce": mf.CLASS_WEIGHT, "loss_mask": mf.MASK_WEIGHT, "loss_dice": mf.DICE_WEIGHT}
    if mf.DEEP_SUPERVISION:
        base = dict(weight_dict)
        for i in range(mf.DEC_LAYERS - 1):
            weight_dict.update({f"{k}_{i}": v for k, v in base.items()})
    return SetCriterion(num_classes, matcher=matcher, weight_dict=weight_dict, eos_coef=mf.NO_OBJECT_WEIGHT,
                        losses=["labels", "masks"], num_points=loss_points or mf.TRAIN_NUM_POINTS,
                        oversample_ratio=mf.OVERSAMPLE_RATIO, importance_sample_ratio=mf.IMPORTANCE_SAMPLE_RATIO)


class _MaskFormerTrainBase(nn.Module):
    """What ProposalModel and PartDistillationModel share on the training path."""

    def _init_common(self, backbone, sem_seg_head, criterion, num_queries, num_classes, size_divisibility, pixel_mean,
                     pixel_std):
        self.backbone, self.sem_seg_head, self.criterion = backbone, sem_seg_head, criterion
        self.num_queries, self.num_classes = num_queries, num_classes
        if size_divisibility < 0:
            size_divisibility = self.backbone.size_divisibility
        self.size_divisibility = size_divisibility
        self.register_buffer("pixel_mean", torch.Tensor(pixel_mean).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.Tensor(pixel_std).view(-1, 1, 1), False)
        self.num_train_iterations = 0
        # the batched criterion samples mask features instead of masks: the decoder need not form the dense masks in training
        predictor = getattr(sem_seg_head, "predictor", None)
        if hasattr(predictor, "dense_masks"):
            predictor.dense_masks = not getattr(criterion, "batched", False)

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess(self, batched_inputs):
        imgs = [x["image"].to(self.device, non_blocking=True) for x in batched_inputs]
        d = max(int(self.size_divisibility), 1)
        h, w = imgs[0].shape[-2:]
        if imgs[0].is_cuda and all(i.shape == imgs[0].shape for i in imgs) and h % d == 0 and w % d == 0 and imgs[0].dim() == 3:
            # same-size images that need no padding (the training crops): (x - mean) / std written straight into the channels-last batch
            # the backbone wants — B + 1 launches instead of 2 B (normalise) + 1 + B (pad-and-copy) + 1 (layout)
            out = torch.empty((len(imgs),) + tuple(imgs[0].shape), dtype=torch.float32, device=self.device, memory_format=torch.channels_last)
            if all(i.dtype == torch.uint8 and i.is_contiguous() for i in imgs) and imgs[0].shape[0] == 3 and len(imgs) <= 16:
                # decoded uint8 pictures (the data pipeline's and the benchmark's form): the whole batch by one launch
                import ctypes
                from . import lib as _lib
                if getattr(self, "_norm_host", None) is None:                    # the constants, read from the buffers once
                    self._norm_host = ((ctypes.c_float * 3)(*self.pixel_mean.flatten().tolist()), (ctypes.c_float * 3)(*self.pixel_std.flatten().tolist()))
                ptrs = (ctypes.c_void_p * len(imgs))(*[i.data_ptr() for i in imgs])
                with torch.cuda.device(self.device):
                    _lib.check(_lib.load().pd_normalize_u8_nhwc(ptrs, len(imgs), int(h), int(w), self._norm_host[0], self._norm_host[1], out.data_ptr(),
                                                                 _lib.current_stream()))
                return ImageList(out, [(int(h), int(w))] * len(imgs))
            for i, im in enumerate(imgs):
                torch.sub(im, self.pixel_mean, out=out[i])
            out.div_(self.pixel_std)
            return ImageList(out, [(int(h), int(w))] * len(imgs))
        images = [(x - self.pixel_mean) / self.pixel_std for x in imgs]
        return ImageList.from_tensors(images, self.size_divisibility)

    def _pad_pseudo_masks(self, inputs, images):
        """zero-pad every image's gt masks to the padded batch size (reference :313-338)."""
        h_pad, w_pad = images.tensor.shape[-2:]
        out = []
        for x in inputs:
            inst = x["instances"].to(self.device)
            if not inst.has("gt_masks"):
                raise ValueError("pseudo label without masks.")
            m = inst.gt_masks.tensor if hasattr(inst.gt_masks, "tensor") else inst.gt_masks
            if tuple(m.shape[-2:]) != (h_pad, w_pad):
                padded = torch.zeros((m.shape[0], h_pad, w_pad), dtype=m.dtype, device=m.device)
                padded[:, : m.shape[1], : m.shape[2]] = m
                m = padded
            out.append((inst, m))
        return out

    @staticmethod
    def _share_padded_masks(targets):
        """pad the per-image masks to [B, n_max, H, W] ONCE per step; the criterion reads it for each of the
        prediction heads instead of re-padding ten times (reference criterion.py:167-169 via utils/misc.py:52-74)."""
        from .utils.misc import nested_tensor_from_tensor_list
        if len(targets) and all(t["masks"].shape[0] > 0 for t in targets):
            ms = [t["masks"] for t in targets]
            if all(m.shape == ms[0].shape for m in ms):             # nothing to pad (the usual case: same crop size, same number of parts):
                padded = torch.stack(ms)                            # one copy instead of two fills + the padding mask nobody reads
            else:
                padded, _ = nested_tensor_from_tensor_list(ms).decompose()
            targets[0]["_padded_masks"] = padded
        return targets

    def _weighted(self, losses):
        """scale by weight_dict, drop unknown keys (reference :192-196).  When the criterion returns stacked per-head
        vectors the scaling is three vector multiplies and the step's total is three sums (LossDict.total)."""
        wd = self.criterion.weight_dict
        vectors = getattr(losses, "vectors", None)
        if vectors is None:
            return {k: v * wd[k] for k, v in losses.items() if k in wd}
        from .modeling.criterion_batched import LossDict
        out, total = LossDict(), 0.0
        cache = self.__dict__.setdefault("_wvec", {})
        stacked = getattr(losses, "stacked", None)
        if stacked is not None:                                       # one node made the three vectors: one multiply, one sum
            names = list(vectors)
            ck = ("stacked", tuple(names), stacked.shape[1], str(stacked.device))
            if ck not in cache:
                cache[ck] = torch.tensor([[wd.get(k, 0.0) for k in [n] + [f"{n}_{i}" for i in range(stacked.shape[1] - 1)]] for n in names],
                                         dtype=stacked.dtype, device=stacked.device)
            wv = stacked * cache[ck]
            for n, row in zip(names, wv.unbind(0)):
                for k, part in zip([n] + [f"{n}_{i}" for i in range(stacked.shape[1] - 1)], row.unbind(0)):
                    if k in wd:
                        out[k] = part
            out.total = wv.sum()
            out.indices, out.points = getattr(losses, "indices", None), getattr(losses, "points", None)
            return out
        for name, vec in vectors.items():
            keys = [name] + [f"{name}_{i}" for i in range(vec.shape[0] - 1)]
            ck = (name, vec.shape[0], str(vec.device))
            if ck not in cache:
                cache[ck] = torch.tensor([wd.get(k, 0.0) for k in keys], dtype=vec.dtype, device=vec.device)
            wv = vec * cache[ck]
            total = total + wv.sum()
            for k, part in zip(keys, wv.unbind(0)):
                if k in wd:
                    out[k] = part
        out.total = total
        out.indices = getattr(losses, "indices", None)                # matched (query, target) pairs of every (image, head)
        out.points = getattr(losses, "points", None)                  # the importance-sampled loss points of every matched pair
        return out


def _count_masks(m):
    """m.sum(0, keepdim=True) of the reference's prepare_targets (int64 count of the part masks covering a pixel).  For bool masks the
    bytes are summed as uint8 with an int64 accumulator: ATen otherwise first writes an int64 copy of the masks (33 MB per image)."""
    if m.dtype == torch.bool:
        return m.view(torch.uint8).sum(0, keepdim=True, dtype=torch.int64)
    return m.sum(0, keepdim=True)


class _PseudoTargets(dict):
    """a training target; "object_masks" (the union count of the part masks, reference proposal_model.py prepare_targets) is formed
    when somebody reads it: the proposal criterion does not, and the bool -> int64 cast + sum over [n, 1024, 1024] cost 95 us per step"""

    def __missing__(self, key):
        if key == "object_masks":
            v = _count_masks(self["masks"])
            self[key] = v
            return v
        raise KeyError(key)


@META_ARCH_REGISTRY.register()
class ProposalModel(_MaskFormerTrainBase):
    @configurable
    def __init__(self, *, backbone, sem_seg_head: nn.Module, criterion: nn.Module, num_queries: int, num_classes: int,
                 size_divisibility: int, pixel_mean: Tuple[float], pixel_std: Tuple[float], test_topk_per_image: int,
                 dataset_name: str = "", use_wandb: bool = True, wandb_vis_period_train: int = 200,
                 wandb_vis_period_test: int = 20, wandb_vis_topk: int = 200, use_unique_per_pixel_label: bool = False,
                 minimum_pseudo_mask_score: float = 0.0, minimum_pseudo_mask_ratio: float = 0.0,
                 apply_masking_with_object_mask: bool = True):
        super().__init__()
        self._init_common(backbone, sem_seg_head, criterion, num_queries, num_classes, size_divisibility, pixel_mean,
                          pixel_std)
        self.test_topk_per_image, self.wandb_vis_topk = test_topk_per_image, wandb_vis_topk
        self.use_wandb = use_wandb                                   # accepted for config parity; never used here
        self.use_unique_per_pixel_label = use_unique_per_pixel_label
        self.minimum_pseudo_mask_score = minimum_pseudo_mask_score
        self.minimum_pseudo_mask_ratio = minimum_pseudo_mask_ratio
        self.apply_masking_with_object_mask = apply_masking_with_object_mask

    @classmethod
    def from_config(cls, cfg):
        backbone = build_backbone(cfg)
        sem_seg_head = build_sem_seg_head(cfg, backbone.output_shape())
        criterion = build_criterion(cfg, sem_seg_head.num_classes)
        return dict(backbone=backbone, sem_seg_head=sem_seg_head, criterion=criterion,
                    num_queries=cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES,
                    size_divisibility=cfg.MODEL.MASK_FORMER.SIZE_DIVISIBILITY, pixel_mean=cfg.MODEL.PIXEL_MEAN,
                    pixel_std=cfg.MODEL.PIXEL_STD, num_classes=cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES,
                    wandb_vis_period_train=cfg.WANDB.VIS_PERIOD_TRAIN, wandb_vis_period_test=cfg.WANDB.VIS_PERIOD_TEST,
                    wandb_vis_topk=cfg.WANDB.VIS_TOPK, use_wandb=not cfg.WANDB.DISABLE_WANDB,
                    dataset_name=cfg.DATASETS.TRAIN[0] if len(cfg.DATASETS.TRAIN) else "",
                    test_topk_per_image=cfg.TEST.DETECTIONS_PER_IMAGE,
                    use_unique_per_pixel_label=cfg.PROPOSAL_LEARNING.USE_PER_PIXEL_LABEL,
                    apply_masking_with_object_mask=cfg.PROPOSAL_LEARNING.APPLY_MASKING_WITH_OBJECT_MASK,
                    minimum_pseudo_mask_ratio=cfg.PROPOSAL_LEARNING.MIN_AREA_RATIO,
                    minimum_pseudo_mask_score=cfg.PROPOSAL_LEARNING.MIN_SCORE)

    def prepare_targets(self, inputs, images):
        if not self.training:
            from .inference import prepare_gt_targets
            return prepare_gt_targets(self, inputs, images)
        return self._prepare_pseudo_targets(inputs, images)

    def _prepare_pseudo_targets(self, inputs, images):
        targets = []
        for inst, m in self._pad_pseudo_masks(inputs, images):
            targets.append(_PseudoTargets({"labels": torch.zeros(m.shape[0], dtype=torch.long, device=self.device),   # class-agnostic
                                           "masks": m}))
        return targets

    def forward(self, batched_inputs):
        images = self.preprocess(batched_inputs)
        if not self.training:                                          # evaluation branch (reference :205-217)
            from .inference import inference
            features = self.backbone(images.tensor)
            targets = self.prepare_targets(batched_inputs, images)
            self.num_test_iterations = getattr(self, "num_test_iterations", 0) + 1
            return inference(self, batched_inputs, targets, images, self.sem_seg_head(features))
        # targets before the backbone (they depend on the inputs only): the num_masks all-reduce of the criterion is in flight
        # while the whole forward runs
        targets = self._share_padded_masks(self.prepare_targets(batched_inputs, images))
        self.criterion.prefetch_num_masks(targets, self.device)
        features = self.backbone(images.tensor)
        outputs = self.sem_seg_head(features)
        losses = self._weighted(self.criterion(outputs, targets))
        self.num_train_iterations += 1
        return losses
