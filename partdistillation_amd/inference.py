"""Evaluation branch of ProposalModel on the device (reference proposal_model.py:220-302, 340-430).

The reference upsamples all Q mask-logit maps to the padded image size, crops / resizes them, gathers the top-k,
multiplies by the object mask, takes sigmoid, scales by the scores and arg-maxes per pixel — every step a full
[Q, H, W] fp32 tensor (0.4 GB per image at 1024^2, Q = 100) — and computes mask IoUs on the CPU through pycocotools.
Here (per-pixel-unique post-processing, the shipped default, no output resize) `pd_mask_assign` interpolates the
low-resolution logits inside the single pass that consumes them and writes an int16 arg-max map, the object map and the
per-query positive-pixel counts; areas, the validity filters and the IoUs with the ground-truth parts are histogram
counts over that map (exact integers, IoU in float64 like pycocotools).  Other settings take the dense route with torch
ops on the device."""
import torch
import torch.nn.functional as F

from . import lib as _lib
from .compat import Instances


def sem_seg_postprocess(result, img_size, output_height, output_width):
    """detectron2.modeling.postprocessing.sem_seg_postprocess: crop the padding, resize to the original resolution."""
    result = result[:, : img_size[0], : img_size[1]].expand(1, -1, -1, -1)
    return F.interpolate(result, size=(output_height, output_width), mode="bilinear", align_corners=False)[0]


def mask_iou(pr, gt):
    """COCO mask IoU (iscrowd = 0) of bool masks [P,H,W] x [n,H,W] -> float64 [P,n]; exact integer counts."""
    a, b = pr.flatten(1).float(), gt.flatten(1).float()
    inter = (a @ b.t()).double()                                          # 0/1 products, sums < 2^24: exact in fp32
    union = a.sum(1).double()[:, None] + b.sum(1).double()[None, :] - inter
    return torch.where(union > 0, inter / union.clamp_min(1), torch.zeros_like(inter))


def _filter(masks_area, obj_area, scores, min_ratio, min_score):
    """the two `if valid.any(): keep valid` filters of _unique_assignment (:277-285 / :289-297) as an index tensor"""
    keep = torch.arange(scores.shape[0], device=scores.device)
    valid = masks_area.double() / obj_area.double() > min_ratio
    if bool(valid.any()):
        keep = keep[valid]
    valid = scores[keep] > min_score
    if bool(valid.any()):
        keep = keep[valid]
    return keep


def _match(iou, labels):
    top1, top1_idx = iou.topk(1, dim=1)
    fg = (top1 > 0.001).flatten()
    return fg, labels[top1_idx.flatten()[fg]]


def instance_inference_fused(model, mask_cls, logits_low, pad_hw, out_hw, target_masks, target_object_masks, target_labels, topk):
    """unique-per-pixel post-processing without dense [Q,H,W] tensors; out_hw == un-padded image size."""
    H, W = out_hw
    dev = logits_low.device
    scores = mask_cls.float().softmax(-1)[:, :-1].topk(1, dim=1)[0].flatten()
    scores, idx = scores.topk(topk, sorted=False)
    sel = logits_low[idx].float().contiguous()                                        # [K, h, w]
    K, h, w = sel.shape
    obj_in = None
    if model.apply_masking_with_object_mask:
        obj_in = target_object_masks.sum(dim=0).bool().to(torch.uint8).contiguous()
    arg = torch.empty((H, W), dtype=torch.int16, device=dev)
    obj = torch.empty((H, W), dtype=torch.uint8, device=dev)
    positive = torch.zeros((K,), dtype=torch.int32, device=dev)
    _lib.check(_lib.load().pd_mask_assign(sel.data_ptr(), scores.contiguous().data_ptr(), obj_in.data_ptr() if obj_in is not None else None,
                                          arg.data_ptr(), obj.data_ptr(), positive.data_ptr(), K, h, w, pad_hw[0], pad_hw[1], H, W,
                                          _lib.current_stream()))
    argl, objb = arg.long(), obj.bool()
    ids = (torch.bincount(argl.flatten(), minlength=K) > 0).nonzero().flatten()       # scoremap.unique()
    area = torch.bincount(argl[objb], minlength=K)                                    # |(scoremap == k) & obj_map|
    keep = ids[_filter(area[ids], objb.sum(), scores[ids], model.minimum_pseudo_mask_ratio, model.minimum_pseudo_mask_score)]
    # IoU with the ground-truth parts from histogram counts over the arg-max map
    inter = torch.stack([torch.bincount(argl[objb & t], minlength=K) for t in target_masks]).t()[keep].double()    # [P, n]
    union = area[keep].double()[:, None] + target_masks.flatten(1).sum(1).double()[None, :] - inter
    iou = torch.where(union > 0, inter / union.clamp_min(1), torch.zeros_like(inter))
    fg, labels = _match(iou, target_labels)
    keep = keep[fg]
    masks = (argl[None] == keep[:, None, None]) & objb[None]
    return masks, scores[keep], labels


def instance_inference_dense(model, mask_cls, mask_pred, target_masks, target_object_masks, target_labels, topk):
    """the reference's sequence on dense [K,H,W] tensors (plain post-processing, or a resized output)"""
    scores = mask_cls.float().softmax(-1)[:, :-1].topk(1, dim=1)[0].flatten()
    scores, idx = scores.topk(topk, sorted=False)
    mask_pred = mask_pred[idx]
    if model.apply_masking_with_object_mask:
        mask_pred = mask_pred * target_object_masks.sum(dim=0, keepdim=True).bool()
    obj_map = mask_pred.max(dim=0)[0] > 0.0
    if model.use_unique_per_pixel_label:
        scoremap = (scores[:, None, None] * mask_pred.sigmoid()).argmax(0)
        ids = scoremap.unique()
        masks = (scoremap[None] == ids[:, None, None]) & obj_map[None]
        scores = scores[ids]
    else:
        masks = mask_pred > 0
    keep = _filter(masks.flatten(1).sum(1), obj_map.sum(), scores, model.minimum_pseudo_mask_ratio, model.minimum_pseudo_mask_score)
    masks, scores = masks[keep], scores[keep]
    fg, labels = _match(mask_iou(masks, target_masks), target_labels)
    return masks[fg], scores[fg], labels


def prepare_gt_targets(model, inputs, images):
    """reference :340-366: part masks / labels (`part_instances`) and object masks (`instances`) padded to the batch size"""
    h_pad, w_pad = images.tensor.shape[-2:]
    out = []
    for x in inputs:
        parts, objs = x["part_instances"].to(model.device), x["instances"].to(model.device)
        pm, om = parts.gt_masks.tensor, objs.gt_masks.tensor
        ppad = torch.zeros((pm.shape[0], h_pad, w_pad), dtype=pm.dtype, device=pm.device)
        ppad[:, : pm.shape[1], : pm.shape[2]] = pm
        opad = torch.zeros((om.shape[0], h_pad, w_pad), dtype=om.dtype, device=om.device)
        opad[:, : om.shape[1], : om.shape[2]] = om
        out.append({"labels": parts.gt_classes.to(model.device), "masks": ppad, "object_masks": opad})
    return out


@torch.no_grad()
def inference(model, batched_inputs, targets, images, outputs, vis=False):
    """reference :220-258 -> [{"proposals": Instances(pred_masks, pred_classes, scores), "gt_masks": Instances(...)}]"""
    logits_all = outputs["pred_masks"]
    if logits_all is None:                                   # decoder ran without dense masks
        from .modeling.transformer_decoder.mask2former_transformer_decoder import materialize_masks
        logits_all = materialize_masks(dict(outputs))["pred_masks"]
    pad_hw = tuple(images.tensor.shape[-2:])
    topk = model.wandb_vis_topk if vis and not model.use_unique_per_pixel_label else model.test_topk_per_image
    results = []
    for cls, low, tgt, inp, size in zip(outputs["pred_logits"], logits_all, targets, batched_inputs, images.image_sizes):
        height, width = inp.get("height", size[0]), inp.get("width", size[1])
        tm = sem_seg_postprocess(tgt["masks"].float(), size, height, width).bool()
        to = sem_seg_postprocess(tgt["object_masks"].float(), size, height, width).bool()
        if model.use_unique_per_pixel_label and (height, width) == tuple(size) and low.is_cuda:
            masks, scores, labels = instance_inference_fused(model, cls, low, pad_hw, (height, width), tm, to, tgt["labels"], topk)
        else:
            dense = F.interpolate(low[None].float(), size=pad_hw, mode="bilinear", align_corners=False)[0]
            dense = sem_seg_postprocess(dense, size, height, width)
            masks, scores, labels = instance_inference_dense(model, cls, dense, tm, to, tgt["labels"], topk)
        if masks.shape[0] == 0:                               # does not contribute to the evaluation (:398-402)
            masks = torch.zeros((1, height, width), dtype=torch.bool, device=low.device)
            scores, labels = scores.new_zeros(1), labels.new_zeros(1)
        r = Instances((height, width))
        r.pred_masks, r.pred_classes, r.scores = masks, labels, scores
        gt = Instances((height, width))
        gt.gt_masks, gt.gt_classes, gt.pred_masks, gt.pred_classes = tm, tgt["labels"], tm, tgt["labels"]
        results.append({"proposals": r, "gt_masks": gt})
    return results


# ================================================================================================ PartDistillationModel
def _unique_assignment_with_classes(model, masks, scores, class_labels):
    """reference part_distillation_model.py:336-383 on the device (its quirks included: see the oracle's docstring).
    Per-pixel-unique branch: queries that won no pixel drop out, the per-query segments are merged per predicted class
    (mask = union, score = best score of the class) - all through one arg-max map, no [K,H,W] stacks besides the result."""
    obj_map = masks.max(dim=0)[0] > 0.0
    if model.use_unique_per_pixel_label:
        scoremap = (scores[:, None, None] * masks.sigmoid()).argmax(0)
        K = masks.shape[0]
        ids = (torch.bincount(scoremap.flatten(), minlength=K) > 0).nonzero().flatten()            # scoremap.unique()
        cls_of_query = torch.full((K,), -1, dtype=torch.long, device=masks.device)
        cls_of_query[ids] = class_labels[ids]
        new_labels = class_labels[ids].unique()
        classmap = torch.where(obj_map, cls_of_query[scoremap], torch.full_like(scoremap, -1))   # class of every object pixel
        new = classmap[None] == new_labels[:, None, None]
        per_query = torch.where(cls_of_query[None, :] == new_labels[:, None], scores[None, :], scores.new_full((), -1.0))
        new_scores = per_query[:, ids].max(dim=1)[0]
        keep = _filter(new.flatten(1).sum(1), obj_map.sum(), new_scores, model.min_pseudo_mask_ratio, model.min_pseudo_mask_score)
        return new[keep], new_scores[keep], new_labels[keep]
    pred = scores[:, None, None] * masks.sigmoid()
    keep = torch.arange(scores.shape[0], device=scores.device)
    valid = (pred > 0.5).flatten(1).sum(1).double() / obj_map.sum().double() > model.min_pseudo_mask_ratio
    if bool(valid.any()):
        masks, keep = pred, keep[valid]                        # (sic) the logits are replaced by score * sigmoid
    valid = scores[keep] > model.min_pseudo_mask_score
    if bool(valid.any()):
        keep = keep[valid]
    return masks[keep] > 0, scores[keep], class_labels[keep]


def instance_inference_with_classification(model, mask_cls, mask_pred, target_mask, target_object_mask, target_labels,
                                           target_object_label, vis=False):
    """reference :459-501"""
    topk = model.wandb_vis_topk if vis and not model.use_unique_per_pixel_label else model.test_topk_per_image
    nc = model.num_part_classes
    scores = mask_cls.float().softmax(-1)[:, :-1]
    labels = torch.arange(nc, device=scores.device).unsqueeze(0).repeat(scores.shape[0], 1).flatten(0, 1)
    scores, idx = scores.flatten(0, 1).topk(topk, sorted=False)
    labels = labels[idx]
    if model.mode == "eval":
        labels = model.majority_vote_mapping[int(target_object_label)][labels]
    mask_pred = mask_pred[torch.div(idx, nc, rounding_mode="floor")]
    if model.apply_masking_with_object_mask:
        mask_pred = mask_pred * target_object_mask.sum(dim=0, keepdim=True).bool()
    masks, scores, labels = _unique_assignment_with_classes(model, mask_pred, scores, labels)
    iou = mask_iou(masks, target_mask)
    top1, top1_idx = iou.topk(1, dim=1)
    fg = (top1 > model.fg_score_threshold).flatten()
    gt_labels = target_labels[top1_idx.flatten()[fg]]
    masks, scores, labels = masks[fg], scores[fg], labels[fg]
    if masks.shape[0] == 0:                                   # does not contribute to the evaluation
        masks = torch.zeros((1,) + tuple(mask_pred.shape[1:]), dtype=torch.bool, device=mask_pred.device)
        scores = scores.new_zeros(1)
        labels = gt_labels = torch.full((1,), nc, dtype=torch.long, device=mask_pred.device)
    r = Instances(tuple(mask_pred.shape[-2:]))
    r.pred_masks, r.scores, r.pred_classes = masks, scores, (gt_labels if model.use_oracle_classifier else labels)
    return r


def prepare_pd_gt_targets(model, inputs, images):
    """reference :430-455: like prepare_gt_targets, plus the object class; key names as in the reference"""
    out = prepare_gt_targets(model, inputs, images)
    for t, x in zip(out, inputs):
        t["object_mask"] = t.pop("object_masks")
        t["gt_object_class"] = x["instances"].to(model.device).gt_classes.to(model.device)
    return out


@torch.no_grad()
def pd_inference(model, batched_inputs, targets, images, outputs, vis=False):
    """reference :239-288 -> [{"predictions": Instances, "gt_instances": Instances, "gt_object_label": ...}]"""
    logits_all = outputs["pred_masks"]
    if logits_all is None:
        from .modeling.transformer_decoder.mask2former_transformer_decoder import materialize_masks
        logits_all = materialize_masks(dict(outputs))["pred_masks"]
    pad_hw = tuple(images.tensor.shape[-2:])
    results = []
    for cls, low, tgt, inp, size in zip(outputs["pred_logits"], logits_all, targets, batched_inputs, images.image_sizes):
        height, width = inp.get("height", size[0]), inp.get("width", size[1])
        dense = F.interpolate(low[None].float(), size=pad_hw, mode="bilinear", align_corners=False)[0]
        dense = sem_seg_postprocess(dense, size, height, width)
        tm = sem_seg_postprocess(tgt["masks"].float(), size, height, width).bool()
        to = sem_seg_postprocess(tgt["object_mask"].float(), size, height, width).bool()
        r = instance_inference_with_classification(model, cls, dense, tm, to, tgt["labels"], tgt["gt_object_class"], vis=vis)
        if model.mode == "save" and not vis:
            save_part_segmentation(model, inp, r)
        gt = Instances((height, width))
        gt.gt_masks, gt.gt_classes, gt.pred_masks, gt.pred_classes = tm, tgt["labels"], tm, tgt["labels"]
        results.append({"predictions": r, "gt_instances": gt, "gt_object_label": tgt["gt_object_class"]})
    return results


# ================================================================================================ PartRankingModel
def _rank_unique_assignment(model, masks, scores, feats):
    """reference part_ranking_model.py:359-401 on the device: per-pixel-unique proposals (queries that won no pixel drop
    out) or plain thresholding, then the area-ratio / score filters (each applied only if something survives it); the
    proposal features follow the proposals."""
    obj_map = masks.max(dim=0)[0] > 0.0
    if model.use_unique_per_pixel_label_during_clustering:
        scoremap = (scores[:, None, None] * masks.sigmoid()).argmax(0)
        ids = (torch.bincount(scoremap.flatten(), minlength=masks.shape[0]) > 0).nonzero().flatten()      # scoremap.unique()
        new = (scoremap[None] == ids[:, None, None]) & obj_map[None]
        scores, feats = scores[ids], feats[ids]
    else:
        new = masks
    area = (new if new.dtype == torch.bool else new > 0).flatten(1).sum(1)
    keep = _filter(area, obj_map.sum(), scores, model.min_pseudo_mask_ratio_1, model.min_pseudo_mask_score_1)
    new = new[keep]
    return (new if new.dtype == torch.bool else new > 0), scores[keep], feats[keep]


def _rank_match_gt(model, masks, scores, extra, target_mask):
    """reference :288-299: proposals whose best IoU with a ground-truth part exceeds fg_score_threshold"""
    iou = mask_iou(masks, target_mask)
    fg = (iou.topk(1, dim=1)[0] > model.fg_score_threshold).flatten()
    return masks[fg], scores[fg], extra[fg]


def rank_use_classifier(model, features, cid):
    """reference :441-457: nearest-centroid scores, negative squared L2 up to the |x|^2 term ("l2") or dot product"""
    if cid not in model.classifier:
        raise ValueError("class ID {} not in classifier. ({})".format(cid, list(model.classifier.keys())))
    y = model.classifier[cid]
    xy = features @ y.t()
    if model.classifier_metric == "l2":
        return xy - (features * features).sum(dim=1)[:, None] - (y * y).sum(dim=1)[None, :]
    if model.classifier_metric == "dot":
        return xy
    raise ValueError(model.classifier_metric)


def rank_instance_inference_with_proposal_feats(model, feats, mask_cls, mask_pred, target_object_mask, vis=False):
    """reference :517-533 (mode "cluster")"""
    unique = model.use_unique_per_pixel_label_during_clustering
    topk = model.wandb_vis_topk if vis and not unique else model.test_topk_per_image
    scores = mask_cls.float().softmax(-1)[:, :-1].flatten()
    scores, idx = scores.topk(topk, sorted=False)
    masks, scores, feats = _rank_unique_assignment(model, mask_pred[idx], scores, feats[idx])
    if model.apply_masking_with_object_mask:
        masks = masks * target_object_mask.sum(dim=0, keepdim=True).bool()
    return masks, scores, feats


def rank_instance_inference_with_classification(model, feats, mask_cls, mask_pred, target_mask, target_object_mask, target_label,
                                                vis=False):
    """reference :460-513: ranking score x nearest-centroid class probability, top-k over (query, cluster) pairs,
    majority-vote mapping in mode "eval", per-class merging, ground-truth matching"""
    import types
    unique = model.use_unique_per_pixel_label_during_labeling
    cid = int(target_label)
    nc = model.classifier[cid].shape[0] if cid in model.classifier else 0
    object_scores = mask_cls.float().softmax(-1)[:, :1]
    scores = object_scores * rank_use_classifier(model, feats, cid).softmax(-1)
    topk = model.wandb_vis_topk if vis and not unique else model.test_topk_per_image
    labels = torch.arange(nc, device=scores.device).unsqueeze(0).repeat(model.num_queries, 1).flatten()
    scores, idx = scores.flatten().topk(topk, sorted=False)
    labels = labels[idx]
    if model.mode == "eval":
        if len(model.majority_vote_mapping) == 0:
            raise ValueError("Class mapping is not registered.")
        labels = model.majority_vote_mapping[cid][labels]
    mask_pred = mask_pred[torch.div(idx, nc, rounding_mode="floor")]
    if model.apply_masking_with_object_mask:
        mask_pred = mask_pred * target_object_mask.sum(dim=0, keepdim=True).bool()
    view = types.SimpleNamespace(use_unique_per_pixel_label=unique, min_pseudo_mask_ratio=model.min_pseudo_mask_ratio_2,
                                 min_pseudo_mask_score=model.min_pseudo_mask_score_2)
    masks, scores, labels = _unique_assignment_with_classes(view, mask_pred, scores, labels)
    masks, scores, labels = _rank_match_gt(model, masks, scores, labels, target_mask)
    if masks.shape[0] == 0:                                   # does not contribute to the evaluation
        masks = torch.zeros((1,) + tuple(mask_pred.shape[1:]), dtype=torch.bool, device=mask_pred.device)
        scores, labels = scores.new_zeros(1), torch.zeros(1, dtype=torch.long, device=mask_pred.device)
    r = Instances(tuple(mask_pred.shape[-2:]))
    r.pred_masks, r.scores, r.pred_classes = masks, scores, labels
    return r


def rank_prepare_targets(model, inputs, images):
    """reference :404-438: evaluation inputs carry part_instances + instances (object), labelling inputs only instances"""
    h_pad, w_pad = images.tensor.shape[-2:]

    def pad(m):
        out = torch.zeros((m.shape[0], h_pad, w_pad), dtype=m.dtype, device=m.device)
        out[:, : m.shape[1], : m.shape[2]] = m
        return out
    out = []
    for x in inputs:
        obj = x["instances"].to(model.device)
        if "part_instances" in x:
            part = x["part_instances"].to(model.device)
            out.append({"part_labels": part.gt_classes.to(model.device), "object_label": obj.gt_classes.to(model.device),
                        "masks": pad(part.gt_masks.tensor), "object_mask": pad(obj.gt_masks.tensor)})
        else:
            m = pad(obj.gt_masks.tensor)
            out.append({"object_label": obj.gt_classes.to(model.device), "masks": m, "object_mask": m})
    return out


@torch.no_grad()
def rank_inference(model, batched_inputs, targets, images, outputs, vis=False):
    """reference :186-258 -> per image {"predictions", "gt_instances", "gt_object_label", "gt_label"
    (+ "proposal_features" in mode "cluster")}"""
    logits_all = outputs["pred_masks"]
    if logits_all is None:
        from .modeling.transformer_decoder.mask2former_transformer_decoder import materialize_masks
        logits_all = materialize_masks(dict(outputs))["pred_masks"]
    feats_all = outputs[model.proposal_key].float()
    if model.proposal_features_norm:
        feats_all = F.normalize(feats_all, p=2, dim=-1)
    pad_hw = tuple(images.tensor.shape[-2:])
    results = []
    for cls, low, feats, tgt, inp, size in zip(outputs["pred_logits"], logits_all, feats_all, targets, batched_inputs, images.image_sizes):
        height, width = inp.get("height", size[0]), inp.get("width", size[1])
        dense = F.interpolate(low[None].float(), size=pad_hw, mode="bilinear", align_corners=False)[0]
        dense = sem_seg_postprocess(dense, size, height, width)
        tm = sem_seg_postprocess(tgt["masks"].float(), size, height, width).bool()
        to = sem_seg_postprocess(tgt["object_mask"].float(), size, height, width).bool()
        res = {}
        if model.mode == "cluster":
            masks, scores, pf = rank_instance_inference_with_proposal_feats(model, feats, cls, dense, to, vis=vis)
            masks, scores, pf = _rank_match_gt(model, masks, scores, pf, tm)
            r = Instances((height, width))
            r.pred_masks, r.scores = masks.bool(), scores
            res["predictions"], res["proposal_features"] = r, pf
            n_rows = pf.shape[0]
        else:
            res["predictions"] = rank_instance_inference_with_classification(model, feats, cls, dense, tm, to, tgt["object_label"], vis=vis)
            if model.mode == "save" and not vis:
                save_generated_part_labels(model, inp, tgt["object_label"], res["predictions"])
            n_rows = feats.shape[0]
        gt = Instances(tuple(tm.shape[-2:]))
        gt.gt_masks, gt.pred_masks = tm, tm
        if "part_labels" in tgt:
            gt.gt_classes = tgt["part_labels"]
        res["gt_instances"], res["gt_object_label"] = gt, tgt["object_label"]
        res["gt_label"] = tgt["object_label"].reshape(-1)[:1].repeat(n_rows)          # (:256) one object label per returned feature row
        results.append(res)
    return results


# ================================================================================================ label export ("save" modes)
def _save_dict(root, inp, res):
    import os
    d = os.path.join(root, str(inp["class_code"]))
    os.makedirs(d, exist_ok=True)
    torch.save(res, os.path.join(d, str(inp["image_id"])))


def save_generated_part_labels(model, inp, label, instance):
    """reference part_ranking_model.py:262-279: one torch.save-d dict per image under root_save_path/class_code/image_id —
    the on-disk pseudo-label format the part-distillation dataset reads (COCO RLE dicts with utf-8 counts)."""
    from .utils import rle
    masks = instance.pred_masks.cpu()
    H, W = masks.shape[1:]
    res = {"file_name": inp["file_name"], "image_id": inp["image_id"], "class_code": inp["class_code"], "height": H, "width": W,
           "part_masks": rle.masks_to_coco_json(masks), "part_labels": instance.pred_classes.cpu(),
           "object_ratio": masks.sum().long().item() / (H * W), "part_ratios": masks.flatten(1).sum(-1) / (H * W),
           "object_class_label": int(label), "part_scores": instance.scores.cpu().numpy()}
    _save_dict(model.root_save_path, inp, res)
    return res


def save_part_segmentation(model, inp, instance):
    """reference part_distillation_model.py:290-307"""
    from .utils import rle
    masks = instance.pred_masks.cpu()
    H, W = masks.shape[1:]
    object_area = masks.sum().long().item()
    res = {"file_name": inp["file_name"], "image_id": inp["image_id"], "class_code": inp["class_code"], "height": H, "width": W,
           "part_masks": rle.masks_to_coco_json(masks), "part_labels": instance.pred_classes.cpu(),
           "part_area_ratios": masks.flatten(1).sum(-1).long() / object_area, "object_ratio": object_area / (H * W),
           "part_scores": instance.scores.cpu().numpy()}
    _save_dict(model.root_save_path, inp, res)
    return res

