"""``PartDistillationModel`` meta-architecture (reference part_distillation/part_distillation_model.py:32-236,
405-428; evaluation branch :239-288, 322-398, 430-501 in ``inference.py``): as ProposalModel
but the targets carry real part labels and the image's object class, and the
targets are handed to the head (``sem_seg_head(features, mask=targets)``, :205)
so the float64 class head can slice that object's K part columns."""
from typing import Tuple

import torch
from torch import nn

from .compat import META_ARCH_REGISTRY, build_backbone, build_sem_seg_head, configurable
from .proposal_model import _MaskFormerTrainBase, build_criterion


@META_ARCH_REGISTRY.register()
class PartDistillationModel(_MaskFormerTrainBase):
    host_reads_object_class = True          # the decoder picks class-head rows from gt_object_class on the host (hipGraph signature)

    @configurable
    def __init__(self, *, backbone, sem_seg_head: nn.Module, criterion: nn.Module, num_queries: int, num_classes: int,
                 size_divisibility: int, pixel_mean: Tuple[float], pixel_std: Tuple[float], test_topk_per_image: int,
                 dataset_name: str = "", use_wandb: bool = True, wandb_vis_period_train: int = 200,
                 wandb_vis_period_test: int = 20, wandb_vis_topk: int = 200, use_unique_per_pixel_label: bool = False,
                 minimum_pseudo_mask_score: float = 0.0, minimum_pseudo_mask_ratio: float = 0.0,
                 apply_masking_with_object_mask: bool = True, use_oracle_classifier: bool = False,
                 num_part_classes: int = 8, num_object_classes: int = 1000):
        super().__init__()
        self._init_common(backbone, sem_seg_head, criterion, num_queries, num_classes, size_divisibility, pixel_mean,
                          pixel_std)
        self.test_topk_per_image = test_topk_per_image
        self.use_wandb = use_wandb
        self.use_unique_per_pixel_label = use_unique_per_pixel_label
        self.minimum_pseudo_mask_score = minimum_pseudo_mask_score
        self.minimum_pseudo_mask_ratio = minimum_pseudo_mask_ratio
        self.apply_masking_with_object_mask = apply_masking_with_object_mask
        self.use_oracle_classifier = use_oracle_classifier
        self.num_part_classes, self.num_object_classes = num_part_classes, num_object_classes
        # evaluation branch (reference :57-99): attribute names as the reference
        self.mode, self.fg_score_threshold, self.wandb_vis_topk = "", 0.1, wandb_vis_topk
        self.min_pseudo_mask_ratio, self.min_pseudo_mask_score = minimum_pseudo_mask_ratio, minimum_pseudo_mask_score
        self.majority_vote_mapping = {}
        self.current_test_iteration = 0
        # mode "save" writes one label file per image below this directory (reference :97-99; created on first use)
        self.root_save_path = "pseudo_labels/part_labels/part_distillation_predictions/{}/{}_{}/".format(
            dataset_name, minimum_pseudo_mask_score, minimum_pseudo_mask_ratio)

    def update_majority_vote_mapping(self, mapping_dict):
        """reference :160-163: object class id -> LongTensor [num_part_classes] of merged part labels (from part ranking)"""
        for cid, mapping in mapping_dict.items():
            self.majority_vote_mapping[int(cid)] = mapping.to(self.device)

    @classmethod
    def from_config(cls, cfg):
        backbone = build_backbone(cfg)
        sem_seg_head = build_sem_seg_head(cfg, backbone.output_shape())
        pd = cfg.PART_DISTILLATION
        criterion = build_criterion(cfg, pd.NUM_PART_CLASSES, match_points=cfg.MODEL.MASK_FORMER.TRAIN_NUM_POINTS_MATCH,
                                    loss_points=cfg.MODEL.MASK_FORMER.TRAIN_NUM_POINTS_LOSS)
        return dict(backbone=backbone, sem_seg_head=sem_seg_head, criterion=criterion,
                    num_queries=cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES,
                    size_divisibility=cfg.MODEL.MASK_FORMER.SIZE_DIVISIBILITY, pixel_mean=cfg.MODEL.PIXEL_MEAN,
                    pixel_std=cfg.MODEL.PIXEL_STD, num_classes=cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES,
                    wandb_vis_period_train=cfg.WANDB.VIS_PERIOD_TRAIN, wandb_vis_period_test=cfg.WANDB.VIS_PERIOD_TEST,
                    wandb_vis_topk=cfg.WANDB.VIS_TOPK, use_wandb=not cfg.WANDB.DISABLE_WANDB,
                    dataset_name=cfg.DATASETS.TRAIN[0] if len(cfg.DATASETS.TRAIN) else "",
                    test_topk_per_image=cfg.TEST.DETECTIONS_PER_IMAGE, use_unique_per_pixel_label=pd.USE_PER_PIXEL_LABEL,
                    apply_masking_with_object_mask=pd.APPLY_MASKING_WITH_OBJECT_MASK,
                    minimum_pseudo_mask_ratio=pd.MIN_AREA_RATIO, minimum_pseudo_mask_score=pd.MIN_SCORE,
                    use_oracle_classifier=pd.USE_ORACLE_CLASSIFIER, num_part_classes=pd.NUM_PART_CLASSES,
                    num_object_classes=pd.NUM_OBJECT_CLASSES)

    def _prepare_pseudo_targets(self, inputs, images):
        from .proposal_model import _count_masks
        targets = []
        for x, (inst, m) in zip(inputs, self._pad_pseudo_masks(inputs, images)):
            targets.append({"labels": inst.gt_classes.long().to(self.device), "masks": m,
                            "object_masks": _count_masks(m), "gt_object_class": int(x["gt_object_class"])})
        return targets

    def forward(self, batched_inputs):
        images = self.preprocess(batched_inputs)
        if self.training:                # targets first: the num_masks all-reduce overlaps the whole forward (see ProposalModel)
            targets = self._share_padded_masks(self._prepare_pseudo_targets(batched_inputs, images))
            self.criterion.prefetch_num_masks(targets, self.device)
        features = self.backbone(images.tensor)
        if not self.training:                                           # evaluation branch (reference :227-236)
            from .inference import pd_inference, prepare_pd_gt_targets
            if self.mode == "save":                                     # pseudo-label export runs on the pseudo targets (:397-399)
                targets = [{"labels": t["labels"], "masks": t["masks"], "object_mask": t["object_masks"],
                            "gt_object_class": torch.as_tensor(t["gt_object_class"])} for t in self._prepare_pseudo_targets(batched_inputs, images)]
            else:
                targets = prepare_pd_gt_targets(self, batched_inputs, images)
            head_targets = [{"gt_object_class": int(t["gt_object_class"])} for t in targets]
            self.current_test_iteration += 1
            return pd_inference(self, batched_inputs, targets, images, self.sem_seg_head(features, mask=head_targets))
        outputs = self.sem_seg_head(features, mask=targets)
        losses = self._weighted(self.criterion(outputs, targets))
        self.num_train_iterations += 1
        return losses
