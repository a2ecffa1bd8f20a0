"""ORACLE (test infrastructure, never imported by the product): CPU restatement of the reference's part-ranking stage —
part_ranking_model.py:186-258 (inference), :282-285 (masking_with_object_mask), :288-299 (match_gt_masks), :303-356
(_unique_assignment_with_classes), :359-401 (_unique_assignment), :441-457 (use_classifier), :460-513
(instance_inference_with_classification), :517-533 (instance_inference_with_proposal_feats) and
evaluation/clustering_module.py:43-80 (per-class sklearn KMeans(n_clusters, random_state=0); scikit-learn is the pinned
third-party implementation and is called, not restated) — in the ORIGINAL dense form.
Pinned against tests/golden/infer_rank.pt (the real reference run, tests/golden/make_golden.py: gen_infer_rank)."""
import torch
import torch.nn.functional as F

from .inference_ref import mask_iou, sem_seg_postprocess, unique_assignment_with_classes


def unique_assignment(masks, scores, feats, unique, min_ratio, min_score):
    """:359-401"""
    obj_map = masks.topk(1, dim=0)[0] > 0.0
    if unique:
        pred = scores[:, None, None] * masks.sigmoid()
        scoremap = pred.topk(1, dim=0)[1]
        ids = scoremap.unique()
        new = torch.stack([(scoremap[0] == cid) & obj_map[0] for cid in ids]).float()
        scores, feats = scores[ids], feats[ids]
        valid = new.flatten(1).sum(1) / obj_map.flatten(1).sum(1) > min_ratio
        if valid.any():
            new, scores, feats = new[valid], scores[valid], feats[valid]
        valid = scores > min_score
        if valid.any():
            new, scores, feats = new[valid], scores[valid], feats[valid]
        return new.bool(), scores, feats
    valid = (masks > 0).flatten(1).sum(1) / obj_map.flatten(1).sum(1) > min_ratio
    if valid.any():
        masks, scores, feats = masks[valid], scores[valid], feats[valid]
    valid = scores > min_score
    if valid.any():
        masks, scores, feats = masks[valid], scores[valid], feats[valid]
    return masks > 0, scores, feats


def match_gt_masks(masks, scores, extra, target_masks, fg_thr):
    """:288-299"""
    top1 = mask_iou(masks, target_masks).topk(1, dim=1)[0]
    fg = (top1 > fg_thr).flatten()
    return masks[fg], scores[fg], extra[fg]


def use_classifier(features, centroids, metric):
    """:447-457"""
    xy = features @ centroids.t()
    if metric == "l2":
        return xy - (features * features).sum(dim=1)[:, None] - (centroids * centroids).sum(dim=1)[None, :]
    return xy


def inference(outputs, feats_all, inputs, object_classes, pad_hw, mode, metric, unique, min_ratio, min_score, topk, centroids,
              mapping_by_class, fg_thr=0.1, feature_norm=True):
    """-> per image (masks, scores, features) in mode "cluster", (masks, scores, classes) otherwise, plus gt_label"""
    up = F.interpolate(outputs["pred_masks"], size=tuple(pad_hw), mode="bilinear", align_corners=False)
    if feature_norm:
        feats_all = F.normalize(feats_all, p=2, dim=-1)
    res = []
    for cls, m, feats, i, oc in zip(outputs["pred_logits"], up, feats_all, inputs, object_classes):
        size = tuple(i["image"].shape[-2:])
        h, w = i.get("height", size[0]), i.get("width", size[1])

        def pad(t):
            out = torch.zeros((t.shape[0],) + tuple(pad_hw), dtype=t.dtype)
            out[:, : t.shape[1], : t.shape[2]] = t
            return out
        m = sem_seg_postprocess(m, size, h, w)
        tm = sem_seg_postprocess(pad(i["part_masks"]).float(), size, h, w).bool()
        to = sem_seg_postprocess(pad(i["object_mask"]).float(), size, h, w).bool()
        cls = cls.to(m)
        obj = to.sum(dim=0, keepdim=True).bool()
        if mode == "cluster":
            scores, idx = cls.softmax(-1)[:, :-1].flatten().topk(topk, sorted=False)
            masks, scores, pf = unique_assignment(m[idx], scores, feats[idx], unique, min_ratio, min_score)
            masks = masks * obj
            masks, scores, pf = match_gt_masks(masks, scores, pf, tm, fg_thr)
            res.append((masks.bool(), scores, pf, torch.full((pf.shape[0],), oc)))
            continue
        cent = centroids[oc]
        nc = cent.shape[0]
        scores = cls.softmax(-1)[:, :1] * use_classifier(feats, cent, metric).softmax(-1)
        labels = torch.arange(nc).unsqueeze(0).repeat(cls.shape[0], 1).flatten()
        scores, idx = scores.flatten().topk(topk, sorted=False)
        labels = labels[idx]
        if mode == "eval":
            labels = mapping_by_class[oc][labels]
        mp = m[torch.div(idx, nc, rounding_mode="floor")] * obj
        masks, scores, labels = unique_assignment_with_classes(mp, scores, labels, unique, min_ratio, min_score)
        masks, scores, labels = match_gt_masks(masks, scores, labels, tm, fg_thr)
        if masks.shape[0] == 0:
            masks, scores, labels = torch.zeros((1,) + tuple(mp.shape[1:]), dtype=torch.bool), scores.new_zeros(1), torch.zeros(1, dtype=torch.long)
        res.append((masks, scores, labels, torch.full((feats.shape[0],), oc)))
    return res


def cluster_centroids(features, labels, num_clusters):
    """clustering_module.py:43-80: one sklearn KMeans per object class that has more proposals than clusters (the others
    get torch.randn centroids: callers seed torch first)"""
    from sklearn.cluster import KMeans
    feats, labs = torch.cat(features, dim=0), torch.cat(labels, dim=0)
    out = {}
    for cid in labs.unique().long().numpy():
        x = feats[labs == cid]
        if x.shape[0] > num_clusters:
            km = KMeans(n_clusters=num_clusters, random_state=0).fit(x)
            out[int(cid)] = torch.tensor(km.cluster_centers_).float()
        else:
            out[int(cid)] = torch.randn(num_clusters, x.shape[1])
    return out
