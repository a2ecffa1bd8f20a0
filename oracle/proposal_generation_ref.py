"""ORACLE (test infrastructure, never imported by the product): CPU restatement of the reference's pixel-grouping
proposal generation — proposal_generation_model.py:117-127 (_prepare_features), :131-160 (per-image pipeline),
:202-237 (_get_superpixels, _measure_distance, generate_pseudo_labels) and detectron2's sem_seg_postprocess — in its
ORIGINAL dense form (C-channel features upsampled to full resolution, gathered, multiplied by the centroids).

K-means: the reference calls scikit-learn (`KMeans(n_clusters=K, random_state=0)`, pinned 1.7.x in this image).
`kmeans_lloyd_np` restates sklearn's Lloyd iteration (centred data, variance-scaled tol, strict / tol convergence, final
re-assignment) for GIVEN initial centres; tests/test_oracle_propgen.py pins it against sklearn itself with the same
`init=` array.  Pinned against the goldens captured from the real reference (tests/golden/propgen.pt)."""
import numpy as np
import torch
import torch.nn.functional as F


def sem_seg_postprocess(result, img_size, output_height, output_width):
    result = result[:, : img_size[0], : img_size[1]].expand(1, -1, -1, -1)
    return F.interpolate(result, size=(output_height, output_width), mode="bilinear", align_corners=False)[0]


def prepare_features(features, keys, normalize):
    H, W = features[keys[0]].shape[-2:]
    out = torch.cat([F.interpolate(features[k], size=(H, W), mode="bilinear", align_corners=False) for k in keys], dim=1)
    return F.normalize(out, dim=1, p=2) if normalize else out


def measure_distance(A, B, metric):
    if metric == "dot":
        return A @ B.T
    return 2 * A @ B.T - (A * A).sum(dim=1)[:, None] - (B * B).sum(1, keepdim=True).t()


def kmeans_lloyd_np(X, init, max_iter=300, tol=1e-4):
    """sklearn.cluster._kmeans lloyd for dense float32 X [N,C] and given initial centres [K,C] ->
    (centres [K,C], labels [N], n_iter).  Empty clusters keep their centre (sklearn relocates; not exercised)."""
    X = np.asarray(X, dtype=np.float32)
    mean = X.mean(axis=0)
    Xc = X - mean
    scaled_tol = np.mean(np.var(Xc, axis=0)) * tol
    centers = np.asarray(init, dtype=np.float32) - mean
    K = centers.shape[0]
    labels_old = np.full(X.shape[0], -1)
    strict, it = False, 0

    def assign(c):
        return ((c * c).sum(1)[None, :] - 2.0 * Xc @ c.T).argmin(1)

    for it in range(1, max_iter + 1):
        labels = assign(centers)
        new = centers.copy()
        for k in range(K):
            sel = labels == k
            if sel.any():
                new[k] = Xc[sel].mean(axis=0)
        shift = ((new - centers) ** 2).sum()
        centers = new
        if np.array_equal(labels, labels_old):
            strict = True
            break
        if shift <= scaled_tol:
            break
        labels_old = labels
    if not strict:
        labels = assign(centers)
    return centers + mean, labels, it


def generate_pseudo_labels(feature, feature_resized, object_mask, object_mask_resized, centroids, metric):
    """reference :224-237 with the centroids given -> (binary masks [P,H,W] bool, label map [H,W] long)."""
    feature_prop = feature_resized[:, object_mask_resized].transpose(0, 1).contiguous()
    pred_labels = measure_distance(feature_prop, centroids, metric).topk(1, dim=1)[1].flatten() + 1
    mask = torch.zeros(feature_resized.shape[-2:], dtype=torch.long)
    mask[torch.where(object_mask_resized)] = pred_labels
    uniq = pred_labels.unique()
    return torch.stack([mask == l for l in uniq]) if len(uniq) else torch.zeros((0,) + tuple(mask.shape), dtype=torch.bool), mask


def proposal_generation(features, inputs, keys, metric, normalize, size_div, K, centroids_fn):
    """the per-image loop of the reference's forward (:131-160) on given backbone features.
    inputs: [{"mask" float [1,H,W], "height", "width"}]; centroids_fn(image index, data [N,C]) -> [K,C].
    -> list of (binary masks, label map, object_mask_resized) (None when the object has <= K feature pixels)."""
    sizes = [tuple(i["mask"].shape[-2:]) for i in inputs]
    Hp, Wp = max(s[0] for s in sizes), max(s[1] for s in sizes)
    if size_div > 1:
        Hp, Wp = (Hp + size_div - 1) // size_div * size_div, (Wp + size_div - 1) // size_div * size_div
    feats = prepare_features(features, keys, normalize)
    feats_resized = F.interpolate(feats, size=(Hp, Wp), mode="bilinear", align_corners=False)
    out = []
    for i, (inp, f, fr, size) in enumerate(zip(inputs, feats, feats_resized, sizes)):
        masks = torch.zeros((inp["mask"].shape[0], Hp, Wp), dtype=inp["mask"].dtype)
        masks[:, : size[0], : size[1]] = inp["mask"]
        h, w = inp.get("height", size[0]), inp.get("width", size[1])
        fr_i = sem_seg_postprocess(fr, size, h, w)
        mask_resized = sem_seg_postprocess(masks, size, h, w)[0].bool()
        mask_low = F.interpolate(masks[None].float(), size=f.shape[-2:], mode="nearest")[0, 0].bool()
        data = f[:, mask_low].transpose(0, 1).contiguous()
        if len(data) <= K:
            out.append(None)
            continue
        cent = centroids_fn(i, data)
        binary, label_map = generate_pseudo_labels(f, fr_i, mask_low, mask_resized, cent, metric)
        out.append((binary, label_map, mask_resized))
    return out
