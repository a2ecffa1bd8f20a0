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
