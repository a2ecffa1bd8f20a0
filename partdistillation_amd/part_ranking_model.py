"""``PartRankingModel`` meta-architecture (reference part_distillation/part_ranking_model.py:34-566):
evaluation-only — the part-proposal network's forward, then either
  mode "cluster": per-image part proposals with their (L2-normalised) decoder query features, for the clustering module
                  (evaluation/clustering_module.py) that turns them into per-object-class K-means centroids, or
  mode "" / "eval": nearest-centroid part classes for the proposals (register_classifier), merged per class, optionally
                  renamed through the majority-vote mapping;  mode "save": the same, and every image's labelled parts are
                  written as the COCO-RLE label file the part-distillation stage trains on (:262-279).
All post-processing stays on the device (inference.py: rank_*)."""
from typing import Tuple

import torch
from torch import nn

from .compat import META_ARCH_REGISTRY, ImageList, build_backbone, build_sem_seg_head, configurable


@META_ARCH_REGISTRY.register()
class PartRankingModel(nn.Module):
    @configurable
    def __init__(self, *, backbone, sem_seg_head: nn.Module, num_queries: int, size_divisibility: int, pixel_mean: Tuple[float],
                 pixel_std: Tuple[float], test_topk_per_image: int, wandb_vis_period: int = 20, wandb_vis_topk: int = 200,
                 use_wandb: bool = False, apply_masking_with_object_mask: bool = True,
                 use_unique_per_pixel_label_during_clustering: bool = True, use_unique_per_pixel_label_during_labeling: bool = True,
                 proposal_key: str = "decoder_output", classifier_metric: str = "l2", dataset_name: str = "", num_clusters: int = 8,
                 proposal_features_norm: bool = True, min_pseudo_mask_ratio_1: float = 0.0, min_pseudo_mask_score_1: float = 0.0,
                 min_pseudo_mask_ratio_2: float = 0.0, min_pseudo_mask_score_2: float = 0.0, fg_score_threshold: float = 0.1,
                 debug: bool = False):
        super().__init__()
        self.backbone, self.sem_seg_head, self.num_queries = backbone, sem_seg_head, num_queries
        if size_divisibility < 0:
            size_divisibility = self.backbone.size_divisibility
        self.size_divisibility = size_divisibility
        self.register_buffer("pixel_mean", torch.Tensor(pixel_mean).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.Tensor(pixel_std).view(-1, 1, 1), False)
        self.use_wandb, self.wandb_vis_period, self.wandb_vis_topk = use_wandb, wandb_vis_period, wandb_vis_topk
        self.mode = ""
        self.test_topk_per_image = test_topk_per_image
        self.proposal_features_norm, self.proposal_key, self.classifier_metric = proposal_features_norm, proposal_key, classifier_metric
        self.apply_masking_with_object_mask = apply_masking_with_object_mask
        self.use_unique_per_pixel_label_during_clustering = use_unique_per_pixel_label_during_clustering
        self.use_unique_per_pixel_label_during_labeling = use_unique_per_pixel_label_during_labeling
        self.min_pseudo_mask_score_1, self.min_pseudo_mask_ratio_1 = min_pseudo_mask_score_1, min_pseudo_mask_ratio_1
        self.min_pseudo_mask_score_2, self.min_pseudo_mask_ratio_2 = min_pseudo_mask_score_2, min_pseudo_mask_ratio_2
        self.fg_score_threshold, self.num_clusters = fg_score_threshold, num_clusters
        self.dataset_name, self.debug = dataset_name, debug
        # mode "save" writes one label file per image below this directory (reference :99-100; created on first use)
        self.root_save_path = "pseudo_labels/part_labels/part_masks_with_class/{}/{}_{}/".format(
            dataset_name.replace("_pre_labeling", "") if not debug else "debug", classifier_metric, num_clusters)
        self.classifier = {}                     # object class id -> centroids [num_clusters, C] on the device
        self.majority_vote_mapping = {}
        predictor = getattr(sem_seg_head, "predictor", None)
        if hasattr(predictor, "dense_masks"):
            predictor.dense_masks = True

    @classmethod
    def from_config(cls, cfg):
        backbone = build_backbone(cfg)
        pr = cfg.PART_RANKING
        return dict(backbone=backbone, sem_seg_head=build_sem_seg_head(cfg, backbone.output_shape()),
                    num_queries=cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES, size_divisibility=cfg.MODEL.MASK_FORMER.SIZE_DIVISIBILITY,
                    pixel_mean=cfg.MODEL.PIXEL_MEAN, pixel_std=cfg.MODEL.PIXEL_STD, wandb_vis_period=cfg.WANDB.VIS_PERIOD_TEST,
                    wandb_vis_topk=cfg.WANDB.VIS_TOPK, use_wandb=not cfg.WANDB.DISABLE_WANDB,
                    test_topk_per_image=cfg.TEST.DETECTIONS_PER_IMAGE,
                    apply_masking_with_object_mask=pr.APPLY_MASKING_WITH_OBJECT_MASK,
                    use_unique_per_pixel_label_during_clustering=pr.USE_PER_PIXEL_LABEL_DURING_CLUSTERING,
                    use_unique_per_pixel_label_during_labeling=pr.USE_PER_PIXEL_LABEL_DURING_LABELING,
                    proposal_key=pr.PROPOSAL_KEY, classifier_metric=pr.CLASSIFIER_METRIC,
                    dataset_name=cfg.DATASETS.TEST[0] if len(cfg.DATASETS.TEST) else "", num_clusters=pr.NUM_CLUSTERS,
                    proposal_features_norm=pr.PROPOSAL_FEATURE_NORM, min_pseudo_mask_ratio_1=pr.MIN_AREA_RATIO_1,
                    min_pseudo_mask_score_1=pr.MIN_SCORE_1, min_pseudo_mask_ratio_2=pr.MIN_AREA_RATIO_2,
                    min_pseudo_mask_score_2=pr.MIN_SCORE_2, debug=pr.DEBUG)

    @property
    def device(self):
        return self.pixel_mean.device

    def num_classes(self, k):
        return self.classifier[int(k)].shape[0]

    def register_classifier(self, centroids_dict):
        """reference :441-445: object class id -> centroids [num_clusters, C] (the clustering module's output)"""
        for cid, centroids in centroids_dict.items():
            self.classifier[int(cid)] = centroids.to(self.device, torch.float32)

    def update_majority_vote_mapping(self, mapping_dict):
        for cid, mapping in mapping_dict.items():
            self.majority_vote_mapping[int(cid)] = mapping.to(self.device)

    def forward(self, batched_inputs):
        assert not self.training, "part ranking is eval-only."
        from .inference import rank_inference, rank_prepare_targets
        images = [(x["image"].to(self.device, non_blocking=True) - self.pixel_mean) / self.pixel_std for x in batched_inputs]
        images = ImageList.from_tensors(images, self.size_divisibility)
        outputs = self.sem_seg_head(self.backbone(images.tensor))
        targets = rank_prepare_targets(self, batched_inputs, images)
        return rank_inference(self, batched_inputs, targets, images, outputs, vis=False)
