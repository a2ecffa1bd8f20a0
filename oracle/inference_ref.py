"""ORACLE (test infrastructure, never imported by the product): CPU restatement of the reference's ProposalModel
evaluation branch — proposal_model.py:220-258 (inference), :263-299 (_unique_assignment), :340-366
(_prepare_gt_targets), :372-378 (masking_with_object_mask), :381-412 (instance_inference), :418-430 (match_gt_labels),
utils/utils.py:35-42 (mask IoU through pycocotools: |a & b| / |a | b| in float64) and detectron2's sem_seg_postprocess —
in the ORIGINAL dense form ([Q, H, W] fp32 masks).  Pinned against tests/golden/infer.pt (the real reference run)."""
import torch
import torch.nn.functional as F


def sem_seg_postprocess(result, img_size, output_height, output_width):
    result = result[:, : img_size[0], : img_size[1]].expand(1, -1, -1, -1)
    return F.interpolate(result, size=(output_height, output_width), mode="bilinear", align_corners=False)[0]


def mask_iou(pr, gt):
    a, b = pr.flatten(1).double(), gt.flatten(1).double()
    inter = a @ b.t()
    union = a.sum(1)[:, None] + b.sum(1)[None, :] - inter
    return torch.where(union > 0, inter / union.clamp_min(1), torch.zeros_like(inter))


def prepare_gt_targets(inputs, pad_hw):
    out = []
    for i in inputs:
        pm = torch.zeros((i["part_masks"].shape[0],) + tuple(pad_hw), dtype=i["part_masks"].dtype)
        pm[:, : i["part_masks"].shape[1], : i["part_masks"].shape[2]] = i["part_masks"]
        om = torch.zeros((i["object_mask"].shape[0],) + tuple(pad_hw), dtype=i["object_mask"].dtype)
        om[:, : i["object_mask"].shape[1], : i["object_mask"].shape[2]] = i["object_mask"]
        out.append({"labels": i["part_labels"], "masks": pm, "object_masks": om})
    return out


def unique_assignment(masks, scores, unique, min_ratio, min_score):
    obj_map = masks.topk(1, dim=0)[0] > 0.0
    if unique:
        pred = scores[:, None, None] * masks.sigmoid()
        scoremap = pred.topk(1, dim=0)[1]
        ids = scoremap.unique()
        new = torch.stack([(scoremap[0] == cid) & obj_map[0] for cid in ids]).float()
        scores = scores[ids]
        valid = new.flatten(1).sum(1) / obj_map.flatten(1).sum(1) > min_ratio
        if valid.any():
            new, scores = new[valid], scores[valid]
        valid = scores > min_score
        if valid.any():
            new, scores = new[valid], scores[valid]
        return new.bool(), scores
    valid = (masks > 0).flatten(1).sum(1) / obj_map.flatten(1).sum(1) > min_ratio
    if valid.any():
        masks, scores = masks[valid], scores[valid]
    valid = scores > min_score
    if valid.any():
        masks, scores = masks[valid], scores[valid]
    return masks > 0, scores


def instance_inference(mask_cls, mask_pred, target_masks, target_object_masks, target_labels, topk, unique, min_ratio, min_score,
                       apply_object_mask=True):
    scores = mask_cls.softmax(-1)[:, :-1].topk(1, dim=1)[0].flatten()
    scores, idx = scores.topk(topk, sorted=False)
    mask_pred = mask_pred[idx]
    if apply_object_mask:
        mask_pred = mask_pred * target_object_masks.sum(dim=0, keepdim=True).bool()
    masks, scores = unique_assignment(mask_pred, scores, unique, min_ratio, min_score)
    iou = mask_iou(masks, target_masks)
    top1, top1_idx = iou.topk(1, dim=1)
    fg = (top1 > 0.001).flatten()
    labels = target_labels[top1_idx.flatten()[fg]]
    masks, scores = masks[fg], scores[fg]
    if masks.shape[0] == 0:
        masks = torch.zeros((1,) + tuple(mask_pred.shape[1:]), dtype=torch.bool)
        scores, labels = scores.new_zeros(1), labels.new_zeros(1)
    return masks, scores, labels


def inference(outputs, inputs, pad_hw, topk, unique, min_ratio, min_score):
    """-> per image (pred masks bool [P,h,w], scores [P], matched labels [P], gt masks bool [n,h,w])"""
    targets = prepare_gt_targets(inputs, pad_hw)
    up = F.interpolate(outputs["pred_masks"], size=tuple(pad_hw), mode="bilinear", align_corners=False)
    res = []
    for cls, m, t, i in zip(outputs["pred_logits"], up, targets, inputs):
        size = tuple(i["image"].shape[-2:])
        h, w = i.get("height", size[0]), i.get("width", size[1])
        m = sem_seg_postprocess(m, size, h, w)
        tm = sem_seg_postprocess(t["masks"].float(), size, h, w).bool()
        to = sem_seg_postprocess(t["object_masks"].float(), size, h, w).bool()
        res.append(instance_inference(cls.to(m), m, tm, to, t["labels"], topk, unique, min_ratio, min_score) + (tm,))
    return res


# ----------------------------------------------------------------------------- PartDistillationModel (class-aware)
def unique_assignment_with_classes(masks, scores, class_labels, unique, min_ratio, min_score):
    """part_distillation_model.py:336-383, quirks included: in the plain branch the area filter REPLACES the logits by
    score * sigmoid(logit) (always positive), so the returned `> 0` masks are all-True whenever that filter keeps anything."""
    obj_map = masks.topk(1, dim=0)[0] > 0.0
    if unique:
        pred = scores[:, None, None] * masks.sigmoid()
        scoremap = pred.topk(1, dim=0)[1]
        ids = scoremap.unique()
        seg = torch.stack([(scoremap[0] == cid) & obj_map[0] for cid in ids]).float()
        scores, class_labels = scores[ids], class_labels[ids]
        new_labels = class_labels.unique()
        new = torch.stack([seg[class_labels == cid].sum(dim=0).bool() for cid in new_labels]).float()
        new_scores = torch.stack([scores[class_labels == cid].topk(1, dim=0)[0].flatten()[0] for cid in new_labels])
        valid = new.flatten(1).sum(1) / obj_map.flatten(1).sum(1) > min_ratio
        if valid.any():
            new, new_scores, new_labels = new[valid], new_scores[valid], new_labels[valid]
        valid = new_scores > min_score
        if valid.any():
            new, new_scores, new_labels = new[valid], new_scores[valid], new_labels[valid]
        return new.bool(), new_scores, new_labels
    pred = scores[:, None, None] * masks.sigmoid()
    valid = (pred > 0.5).flatten(1).sum(1) / obj_map.flatten(1).sum(1) > min_ratio
    if valid.any():
        masks, scores, class_labels = pred[valid], scores[valid], class_labels[valid]
    valid = scores > min_score
    if valid.any():
        masks, scores, class_labels = masks[valid], scores[valid], class_labels[valid]
    return masks > 0, scores, class_labels


def instance_inference_with_classification(mask_cls, mask_pred, target_mask, target_object_mask, target_labels, mapping, num_classes,
                                           num_queries, topk, unique, min_ratio, min_score, fg_thr, oracle_classifier):
    scores = mask_cls.softmax(-1)[:, :-1]
    labels = torch.arange(num_classes).unsqueeze(0).repeat(num_queries, 1).flatten(0, 1)
    scores, idx = scores.flatten(0, 1).topk(topk, sorted=False)
    labels = labels[idx]
    if mapping is not None:
        labels = mapping[labels]
    idx = torch.div(idx, num_classes, rounding_mode="floor")
    mask_pred = mask_pred[idx] * target_object_mask.sum(dim=0, keepdim=True).bool()
    masks, scores, labels = unique_assignment_with_classes(mask_pred, scores, labels, unique, min_ratio, min_score)
    iou = mask_iou(masks, target_mask)
    top1, top1_idx = iou.topk(1, dim=1)
    fg = (top1 > fg_thr).flatten()
    gt_labels = target_labels[top1_idx.flatten()[fg]]
    masks, scores, labels = masks[fg], scores[fg], labels[fg]
    if masks.shape[0] == 0:
        masks = torch.zeros((1,) + tuple(mask_pred.shape[1:]), dtype=torch.bool)
        scores = scores.new_zeros(1)
        labels = gt_labels = torch.full((1,), num_classes, dtype=torch.long)
    return masks, scores, (gt_labels if oracle_classifier else labels)


def inference_pd(outputs, inputs, object_classes, pad_hw, num_classes, topk, unique, min_ratio, min_score, mapping_by_class,
                 fg_thr=0.1, oracle_classifier=False):
    targets = prepare_gt_targets(inputs, pad_hw)
    up = F.interpolate(outputs["pred_masks"], size=tuple(pad_hw), mode="bilinear", align_corners=False)
    res = []
    for cls, m, t, i, oc in zip(outputs["pred_logits"], up, targets, inputs, object_classes):
        size = tuple(i["image"].shape[-2:])
        h, w = i.get("height", size[0]), i.get("width", size[1])
        m = sem_seg_postprocess(m, size, h, w)
        tm = sem_seg_postprocess(t["masks"].float(), size, h, w).bool()
        to = sem_seg_postprocess(t["object_masks"].float(), size, h, w).bool()
        mapping = None if mapping_by_class is None else mapping_by_class[oc]
        res.append(instance_inference_with_classification(cls.to(m), m, tm, to, t["labels"], mapping, num_classes, cls.shape[0], topk,
                                                          unique, min_ratio, min_score, fg_thr, oracle_classifier))
    return res
