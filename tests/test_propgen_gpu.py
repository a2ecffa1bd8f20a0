"""GPU parity of the pixel-grouping proposal generation (BASELINE config 4): label-map kernel, device
K-means and the whole ProposalGenerationModel against the CPU oracle and the goldens captured from the reference."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import common as C
from oracle import proposal_generation_ref as P

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("K,h,w,Hp,Wp,H,W", [(4, 16, 16, 128, 128, 128, 128), (4, 16, 16, 128, 128, 112, 120), (7, 5, 9, 40, 72, 33, 70)])
def test_scores_argmax_kernel_vs_torch(K, h, w, Hp, Wp, H, W):
    from partdistillation_amd import lib
    scores = C.seeded((K, h, w), 1).to(DEV)
    mask = (C.seeded((H, W), 2) > -0.3).to(DEV)
    labels = torch.empty((H, W), dtype=torch.uint8, device=DEV)
    lib.check(lib.load().pd_scores_argmax_u8(scores.data_ptr(), mask.to(torch.uint8).data_ptr(), labels.data_ptr(), K, h, w, Hp, Wp, H, W,
                                             lib.current_stream()))
    up = F.interpolate(scores[None], size=(Hp, Wp), mode="bilinear", align_corners=False)[0, :, :H, :W]
    want = torch.where(mask, up.argmax(0) + 1, torch.zeros((), dtype=torch.long, device=DEV))
    top2 = up.topk(2, dim=0)[0]
    clear = (top2[0] - top2[1]) > 1e-5                       # leave out numerical near-ties
    assert torch.equal(labels.long()[clear], want[clear]) and (~clear).float().mean() < 0.01
    assert (labels[~mask] == 0).all()


def test_device_lloyd_matches_oracle_and_sklearn_semantics():
    from partdistillation_amd.functions.kmeans import kmeans_lloyd
    rng = np.random.default_rng(5)
    for N, Cc, K in [(83, 40, 4), (5700, 64, 4), (40, 8, 3)]:
        blobs = rng.normal(size=(K, Cc)).astype(np.float32) * 2
        X = (blobs[rng.integers(K, size=N)] + rng.normal(size=(N, Cc)).astype(np.float32)).astype(np.float32)
        init = X[rng.choice(N, K, replace=False)].copy()
        c_ref, l_ref, it_ref = P.kmeans_lloyd_np(X, init)
        c, l, it = kmeans_lloyd(torch.from_numpy(X).to(DEV), K, init=torch.from_numpy(init).to(DEV))
        assert (l.cpu().numpy() != l_ref).mean() < 2e-3, (N, Cc, K)
        np.testing.assert_allclose(c.cpu().numpy(), c_ref, rtol=1e-3, atol=1e-3)
        assert abs(it - it_ref) <= 1


def test_batched_hip_lloyd_matches_oracle():
    """pd_kmeans_assign / pd_kmeans_update, three images of different sizes advancing together, against the sklearn-pinned
    restatement with the same initial centres: same iteration counts, same centres"""
    from partdistillation_amd.functions.kmeans import kmeans_lloyd_batched
    rng = np.random.default_rng(11)
    datas, inits, refs = [], [], []
    for N, K in [(83, 4), (5431, 4), (700, 4)]:
        Cc = 1152
        blobs = rng.normal(size=(K, Cc)).astype(np.float32)
        X = (blobs[rng.integers(K, size=N)] + 1.5 * rng.normal(size=(N, Cc)).astype(np.float32)).astype(np.float32)
        init = X[rng.choice(N, K, replace=False)].copy()
        datas.append(torch.from_numpy(X).to(DEV)), inits.append(torch.from_numpy(init).to(DEV))
        refs.append(P.kmeans_lloyd_np(X, init))
    centers, n_iters = kmeans_lloyd_batched(datas, 4, inits=inits)
    for b, (c_ref, l_ref, it_ref) in enumerate(refs):
        np.testing.assert_allclose(centers[b].cpu().numpy(), c_ref, rtol=2e-3, atol=2e-3)
        assert abs(n_iters[b] - it_ref) <= 1, (b, n_iters[b], it_ref)
    c3, n3 = kmeans_lloyd_batched([d[:, :40].contiguous() for d in datas[:1]], 3, inits=[inits[0][:3, :40].contiguous()])
    c_ref, _, it_ref = P.kmeans_lloyd_np(datas[0][:, :40].cpu().numpy(), inits[0][:3, :40].cpu().numpy())
    np.testing.assert_allclose(c3[0].cpu().numpy(), c_ref, rtol=2e-3, atol=2e-3)


def test_bounded_e_step_is_bit_identical_to_the_exhaustive_one(monkeypatch):
    """pd_kmeans_assign_bounded (distance bounds: most points are not read after the first iterations; slabs without a changed label keep their
    partial sums) against pd_kmeans_assign_partial: the SAME centres bit for bit and the same iteration counts — well separated blobs, heavily
    overlapping ones (long runs, many points near a boundary), L2-normalised part-ranking features with K = 8, one cluster fewer than blobs."""
    from partdistillation_amd.functions import kmeans as km
    rng = np.random.default_rng(5)
    cases = []
    for N, K, Cc, spread, norm in [(5431, 4, 1536, 1.5, False), (3000, 4, 1536, 6.0, False), (1300, 8, 256, 1.0, True), (900, 3, 64, 2.5, False),
                                   (70, 4, 1536, 1.0, False)]:
        blobs = rng.normal(size=(K + 1, Cc)).astype(np.float32)
        X = (blobs[rng.integers(K + 1, size=N)] + spread * rng.normal(size=(N, Cc)).astype(np.float32)).astype(np.float32)
        if norm:
            X /= np.linalg.norm(X, axis=1, keepdims=True)
        cases.append((K, torch.from_numpy(X).to(DEV), torch.from_numpy(X[rng.choice(N, K, replace=False)].copy()).to(DEV)))
    for K in (4, 8, 3):
        datas, inits = [c[1] for c in cases if c[0] == K], [c[2] for c in cases if c[0] == K]
        out = {}
        for on in (False, True):
            monkeypatch.setattr(km, "BOUNDED", on)
            out[on] = km.kmeans_lloyd_batched(datas, K, inits=inits)
        assert out[True][1] == out[False][1], (K, out[True][1], out[False][1])
        assert torch.equal(out[True][0], out[False][0]), K
        assert max(out[True][1]) >= 5


def test_batched_hip_lloyd_k8_part_ranking_geometry():
    """the KMAX = 8 instantiation (part ranking: 8 clusters per object class on 256-d query features,
    evaluation/clustering_module.py:43-70): several classes of different sizes advancing together against the sklearn-pinned
    restatement with the same initial centres; K = 5 and 7 go through the same instantiation"""
    from partdistillation_amd.functions.kmeans import kmeans_lloyd_batched
    rng = np.random.default_rng(23)
    for K in (8, 5, 7):
        datas, inits, refs = [], [], []
        for N in (400, 57, 1300, 9):
            Cc = 256
            blobs = rng.normal(size=(K, Cc)).astype(np.float32) * 0.6
            X = (blobs[rng.integers(K, size=N)] + rng.normal(size=(N, Cc)).astype(np.float32)).astype(np.float32)
            X /= np.linalg.norm(X, axis=1, keepdims=True)                      # L2-normalised query features
            init = X[rng.choice(N, K, replace=False)].copy()
            datas.append(torch.from_numpy(X).to(DEV)), inits.append(torch.from_numpy(init).to(DEV))
            refs.append(P.kmeans_lloyd_np(X, init))
        centers, n_iters = kmeans_lloyd_batched(datas, K, inits=inits)
        for b, (c_ref, l_ref, it_ref) in enumerate(refs):
            np.testing.assert_allclose(centers[b].cpu().numpy(), c_ref, rtol=2e-3, atol=2e-4, err_msg=f"K={K} class {b}")
            assert abs(n_iters[b] - it_ref) <= 1, (K, b, n_iters[b], it_ref)


def test_kmeans_plusplus_seeding_gives_a_comparable_partition():
    """own RNG, so not sklearn's partition - but the objective must be in the same league"""
    from sklearn.cluster import KMeans
    from partdistillation_amd.functions.kmeans import kmeans_lloyd
    rng = np.random.default_rng(9)
    blobs = rng.normal(size=(4, 32)).astype(np.float32) * 3
    X = (blobs[rng.integers(4, size=2000)] + rng.normal(size=(2000, 32)).astype(np.float32)).astype(np.float32)
    sk = KMeans(n_clusters=4, random_state=0).fit(X)
    g = torch.Generator(device=DEV).manual_seed(0)
    c, l, _ = kmeans_lloyd(torch.from_numpy(X).to(DEV), 4, generator=g)
    inertia = ((torch.from_numpy(X).to(DEV) - c[l]) ** 2).sum().item()
    assert inertia < 1.1 * sk.inertia_


def _model(metric, norm, feats):
    from partdistillation_amd.proposal_generation_model import ProposalGenerationModel

    class Stub(torch.nn.Module):
        size_divisibility = 32

        def forward(self, x):
            return {k: v.to(x.device) for k, v in feats.items()}
    m = ProposalGenerationModel(backbone=Stub(), size_divisibility=C.PROPGEN["size_div"], dataset_name="synthetic",
                                pixel_mean=[123.675, 116.28, 103.53], pixel_std=[58.395, 57.12, 57.375], distance_metric=metric,
                                backbone_feature_key_list=["res3", "res4"], num_superpixel_clusters=C.PROPGEN["K"],
                                feature_normalize=norm)
    return m.to(DEV).eval()


@pytest.mark.parametrize("tag,metric,norm", [("dot_0", "dot", False), ("l2_1", "l2", True)])
def test_proposal_generation_model_vs_reference_golden(golden, tag, metric, norm):
    """whole model (stub backbone) with the reference's final centroids as the K-means start: the label map from the
    low-resolution score maps must reproduce the reference's dense full-resolution labelling"""
    from partdistillation_amd.compat import BitMasks, Instances
    from partdistillation_amd.utils import rle
    g = golden("propgen")[tag]
    feats, inputs = C.make_propgen_inputs()
    model = _model(metric, norm, feats)
    model.init_centroids = lambda i: g[i]["centroids"].to(DEV)
    batched = []
    for i in inputs:
        inst = Instances(tuple(i["mask"].shape[-2:]))
        inst.gt_masks = BitMasks(i["mask"])
        batched.append({"image": i["image"], "instances": inst, "height": i["height"], "width": i["width"], "file_name": i["file_name"],
                        "class_code": i["class_code"], "gt_object_class": i["gt_object_class"]})
    res = model(batched)
    for r, want in zip(res, g):
        torch.testing.assert_close(r["centroids"].cpu(), want["centroids"], rtol=1e-3, atol=1e-4)
        assert r["kmeans_iterations"] <= 2
        got = model.binary_masks(r).cpu()
        assert got.shape == want["pseudo_label"].shape
        mismatch = (got != want["pseudo_label"]).any(0).float().mean().item()
        assert mismatch < 2e-3, mismatch                     # fp32 re-association at near-ties only
        assert r["height"] == want["pseudo_label"].shape[1] and r["class_index"] == 7
        assert abs(r["object_ratio"] - want["object_mask_resized"].float().mean().item()) < 1e-6
        for j, l in zip(r["part_mask"], r["present_labels"]):
            seg = j["segmentation"]
            assert (rle.decode({"size": seg["size"], "counts": seg["counts"]}) == (r["labels"].cpu().numpy() == l)).all()


# ----------------------------------------------------------------------------- evaluation branch of ProposalModel (§8f-2)
@pytest.mark.parametrize("metric", ["dot", "l2"])
def test_config4_full_size_labels_equal_dense_reference_labelling(metric, monkeypatch):
    """BASELINE config 4 at FULL size (SURVEY §8f-1 measurement spec): 4 x 1024 x 1024 synthetic images, one elliptical object each
    (~35 % of the area), R50 backbone, res3 + res4 (C = 1536), K = 4.  The product labels the image from K score maps formed at
    feature resolution; the reference (proposal_generation_model.py:141-146, 226-227) upsamples the C-channel features to full
    resolution (6.4 GB per image) and takes the arg-max of `features . centroids` on the object's pixels.  Bilinear interpolation
    is linear, so the two are the same function in real arithmetic: restate the reference's dense route here with torch ops on the
    SAME features and the product's own centroids ("identical centroids injected") and count the pixels that differ — only fp32
    re-association near-ties may (stated bound: 1e-4 of the object's pixels; SURVEY expects < 1e-5 on real features).  Also: label 0
    exactly outside the object, labels 1..K inside, the COCO RLE of each label decodes to `labels == l`, and the device
    Lloyd's fixed point is a Lloyd fixed point of the reference's clustering input (one more sklearn-semantics assignment step on
    the CPU restatement leaves the labels of the 1/8-resolution object pixels unchanged up to the same near-tie bound)."""
    import os
    from oracle import proposal_generation_ref as P
    from partdistillation_amd.compat import BitMasks, Instances, build_model
    from partdistillation_amd.config import setup_cfg
    from partdistillation_amd.utils import rle
    import partdistillation_amd.modeling, partdistillation_amd.proposal_generation_model  # noqa: F401,E401
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = setup_cfg(os.path.join(root, "partdistillation_amd", "configs", "proposal_generation", "r50.yaml"),
                    ["PROPOSAL_GENERATION.DISTANCE_METRIC", metric])
    assert cfg.PROPOSAL_GENERATION.NUM_SUPERPIXEL_CLUSTERS == 4 and list(cfg.PROPOSAL_GENERATION.BACKBONE_FEATURE_KEY_LIST) == ["res3", "res4"]
    torch.manual_seed(0)
    model = build_model(cfg).to(DEV).eval()
    S, Bn = 1024, 4
    ys, xs = torch.meshgrid(torch.arange(S) / S, torch.arange(S) / S, indexing="ij")
    g = torch.Generator().manual_seed(44)
    batch, masks = [], []
    for b in range(Bn):
        cy, cx = 0.5 + 0.03 * (b - 1.5), 0.5 - 0.02 * (b - 1.5)                      # four different ellipses, ~35 % of the image
        m = (((ys - cy) ** 2 / (0.13 - 0.005 * b) + (xs - cx) ** 2 / (0.085 + 0.004 * b)) < 1.0)
        inst = Instances((S, S))
        inst.gt_masks = BitMasks(m[None].to(DEV))
        # smooth random images: the features of white noise have no cluster structure at all
        img = F.interpolate(torch.rand(1, 3, S // 16, S // 16, generator=g), size=(S, S), mode="bicubic", align_corners=False)[0].clamp(0, 1) * 255
        batch.append({"image": img.to(DEV), "instances": inst, "file_name": f"{b}.pth", "class_code": "n0"})
        masks.append(m.to(DEV))
        assert 0.3 < m.float().mean() < 0.4
    feats = {}
    f0 = model._prepare_features
    monkeypatch.setattr(model, "_prepare_features", lambda fo: feats.setdefault("f", f0(fo)))
    model.kmeans_generator = torch.Generator(device=DEV).manual_seed(0)
    res = model(batch)
    f = feats["f"]
    assert f.shape == (Bn, 1536, S // 8, S // 8) and f.dtype == torch.float32
    worst = 0.0
    for b, r in enumerate(res):
        labels, cen, m = r["labels"], r["centroids"].float(), masks[b]
        assert labels.shape == (S, S) and labels.dtype == torch.uint8
        assert not labels[~m].any() and bool((labels[m] >= 1).all()) and bool((labels <= 4).all())
        # ("dot" on post-ReLU features may hand every pixel to the longest centroid — the reference's metric does the same; the
        # nearest-centroid metric uses all four)
        assert set(r["present_labels"]) <= {1, 2, 3, 4} and (metric == "dot" or len(r["present_labels"]) >= 2)
        # the reference's dense route: [C, H, W] features -> object pixels -> distance to the centroids -> top-1
        dense = F.interpolate(f[b:b + 1], size=(S, S), mode="bilinear", align_corners=False)[0]           # 6.4 GB
        obj = dense[:, m]                                                                                  # [C, N]
        del dense
        sc = cen @ obj
        if metric == "l2":
            sc = 2.0 * sc - (cen * cen).sum(1)[:, None] - (obj * obj).sum(0)[None]                         # -(|x - c|^2), as :214-218
        want = sc.argmax(0).to(torch.uint8) + 1
        top2 = sc.topk(2, dim=0)[0]
        del obj, sc
        diff = (labels[m] != want)
        worst = max(worst, diff.float().mean().item())
        # every differing pixel is a numerical near-tie of the two best centroids
        margin = (top2[0] - top2[1])[diff]
        assert diff.float().mean().item() < 1e-4, (b, diff.float().mean().item())
        assert margin.numel() == 0 or float(margin.max()) <= 1e-3 * float(top2[0].abs().max()), float(margin.max())
        # COCO RLE per label == the label map (column-major runs; utils/rle.decode is pinned to the reference encoder's goldens)
        for l, entry in zip(r["present_labels"], r["part_mask"]):
            seg = dict(entry["segmentation"])
            seg["counts"] = seg["counts"].encode("utf-8") if isinstance(seg["counts"], str) else seg["counts"]
            assert np.array_equal(rle.decode(seg).astype(bool), (labels == l).cpu().numpy())
        # the converged centroids are a Lloyd fixed point of the reference's clustering input (masked 1/8-resolution vectors)
        m_low = F.interpolate(m[None, None].float(), size=f.shape[-2:], mode="nearest")[0, 0].bool()
        X = f[b][:, m_low].t().contiguous()
        d2 = (X * X).sum(1, keepdim=True) - 2.0 * X @ cen.t() + (cen * cen).sum(1)[None]
        assign = d2.argmin(1)
        newc = torch.stack([X[assign == k].mean(0) for k in range(4)])
        shift = ((newc - cen) ** 2).sum().item()
        tol = 1e-4 * X.var(0).mean().item()                                                              # sklearn's tol * mean variance
        # a point whose two nearest centres tie to fp32 re-association noise may sit in the other cluster here than on the device: each
        # such point moves two centres by at most |x - c| / (cluster size) — (2 |x|_max / n_min)^2 of squared shift per near-tie point
        top2 = (-d2).topk(2, dim=1)[0]
        near = int(((top2[:, 0] - top2[:, 1]) <= 1e-4 * d2.abs().max()).sum())
        n_min = min(int((assign == k).sum()) for k in range(4))
        slack = near * (2.0 * float(X.norm(dim=1).max()) / max(n_min, 1)) ** 2
        assert r["kmeans_iterations"] >= 1 and (shift <= tol * 4 + slack or r["kmeans_iterations"] >= 300), (shift, tol, slack, near, r["kmeans_iterations"])
    print(f"config 4 full size ({metric}): worst label mismatch {worst:.2e} of the object's pixels; Lloyd iterations "
          f"{[int(r['kmeans_iterations']) for r in res]}")


@pytest.mark.parametrize("tag,unique,min_score", [("unique_1", True, -1.0), ("unique_0", False, 0.3)])
def test_proposal_model_inference_vs_reference_golden(golden, tag, unique, min_score):
    """predicted part masks / scores / matched labels of the device evaluation branch against the real reference run:
    image 0 takes the fused pd_mask_assign route (no output resize, unique labels), image 1 the dense route"""
    import types
    from partdistillation_amd import inference as I
    from partdistillation_amd.compat import BitMasks, ImageList, Instances
    g = golden("infer")[tag]
    outputs, inputs = C.make_infer_inputs()
    outputs = {k: v.to(DEV) for k, v in outputs.items()}
    model = types.SimpleNamespace(device=torch.device(DEV), test_topk_per_image=C.INFER["topk"], wandb_vis_topk=C.INFER["topk"],
                                  use_unique_per_pixel_label=unique, minimum_pseudo_mask_ratio=0.02, minimum_pseudo_mask_score=min_score,
                                  apply_masking_with_object_mask=True)
    batched = []
    for i in inputs:
        parts, objs = Instances(tuple(i["image"].shape[-2:])), Instances(tuple(i["image"].shape[-2:]))
        parts.gt_masks, parts.gt_classes = BitMasks(i["part_masks"]), i["part_labels"]
        objs.gt_masks = BitMasks(i["object_mask"])
        batched.append({"image": i["image"], "part_instances": parts, "instances": objs, "height": i["height"], "width": i["width"]})
    images = ImageList.from_tensors([i["image"].to(DEV) for i in inputs], C.INFER["size_div"])
    targets = I.prepare_gt_targets(model, batched, images)
    res = I.inference(model, batched, targets, images, outputs)
    for r, want in zip(res, g):
        p = r["proposals"]
        assert p.pred_masks.shape == want["pred_masks"].shape, (p.pred_masks.shape, want["pred_masks"].shape)
        # `scores.topk(k, sorted=False)` leaves the order of the proposals to the implementation (it differs between the
        # CPU and GPU kernels of torch itself): compare as sets, paired by their (distinct) scores
        o1, o2 = p.scores.cpu().argsort(), want["scores"].argsort()
        assert torch.equal(p.pred_classes.cpu()[o1], want["pred_classes"][o2])
        torch.testing.assert_close(p.scores.cpu()[o1], want["scores"][o2], rtol=1e-5, atol=1e-6)
        diff = (p.pred_masks.cpu()[o1] != want["pred_masks"][o2]).float().mean().item()
        assert diff < 2e-3, diff                                       # interpolation / sigmoid rounding at near-ties
        assert torch.equal(r["gt_masks"].gt_masks.cpu(), want["gt_masks"])


def test_proposal_model_eval_end_to_end():
    """the registered ProposalModel in eval mode: backbone -> head (dense masks at inference) -> device evaluation branch"""
    import os
    from partdistillation_amd.compat import BitMasks, Instances, build_model
    from partdistillation_amd.config import setup_cfg
    import partdistillation_amd.modeling, partdistillation_amd.proposal_model  # noqa: F401,E401
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = setup_cfg(os.path.join(root, "partdistillation_amd", "configs", "proposal_learning", "r50_mask2former.yaml"),
                    ["MODEL.MASK_FORMER.NUM_OBJECT_QUERIES", "20", "MODEL.MASK_FORMER.DEC_LAYERS", "3",
                     "MODEL.SEM_SEG_HEAD.TRANSFORMER_ENC_LAYERS", "1", "TEST.DETECTIONS_PER_IMAGE", "10"])
    model = build_model(cfg).eval()
    _, inputs = C.make_infer_inputs()
    batched = []
    for i in inputs:
        parts, objs = Instances(tuple(i["image"].shape[-2:])), Instances(tuple(i["image"].shape[-2:]))
        parts.gt_masks, parts.gt_classes = BitMasks(i["part_masks"]), i["part_labels"]
        objs.gt_masks = BitMasks(i["object_mask"])
        batched.append({"image": i["image"], "part_instances": parts, "instances": objs})
    with torch.no_grad():
        res = model(batched)
    assert len(res) == 2
    for r, i in zip(res, inputs):
        p = r["proposals"]
        assert p.pred_masks.dtype == torch.bool and tuple(p.pred_masks.shape[-2:]) == tuple(i["image"].shape[-2:])
        assert p.pred_masks.shape[0] == p.scores.shape[0] == p.pred_classes.shape[0] >= 1
        assert not (p.pred_masks & ~i["object_mask"].to(DEV)).any()            # proposals stay inside the object
        assert r["gt_masks"].gt_masks.shape[0] == 3


@pytest.mark.parametrize("tag,mode,unique,min_score,oracle_cls", [("raw_1", "", True, -1.0, False), ("eval_1", "eval", True, -1.0, False),
                                                                  ("eval_0", "eval", False, 0.05, True)])
def test_part_distillation_inference_vs_reference_golden(golden, tag, mode, unique, min_score, oracle_cls):
    """class-aware evaluation branch of PartDistillationModel on the device against the real reference run"""
    import types
    from partdistillation_amd import inference as I
    from partdistillation_amd.compat import BitMasks, ImageList, Instances
    g = golden("infer_pd")[tag]
    outputs, inputs = C.make_infer_inputs()
    K = C.INFER_PD_CLASSES
    outputs = {"pred_masks": outputs["pred_masks"].to(DEV), "pred_logits": (C.seeded((len(inputs), C.INFER["Q"], K + 1), 5300) * 2).to(DEV)}
    model = types.SimpleNamespace(device=torch.device(DEV), test_topk_per_image=C.INFER["topk"] * 2, wandb_vis_topk=C.INFER["topk"] * 2,
                                  use_unique_per_pixel_label=unique, min_pseudo_mask_ratio=0.02, min_pseudo_mask_score=min_score,
                                  apply_masking_with_object_mask=True, num_part_classes=K, mode=mode, fg_score_threshold=0.1,
                                  use_oracle_classifier=oracle_cls,
                                  majority_vote_mapping={3: torch.tensor(C.INFER_PD_MAPPING[0], device=DEV), 4: torch.tensor(C.INFER_PD_MAPPING[1], device=DEV)})
    batched = []
    for b, i in enumerate(inputs):
        parts, objs = Instances(tuple(i["image"].shape[-2:])), Instances(tuple(i["image"].shape[-2:]))
        parts.gt_masks, parts.gt_classes = BitMasks(i["part_masks"]), i["part_labels"]
        objs.gt_masks, objs.gt_classes = BitMasks(i["object_mask"]), torch.tensor([3 + b])
        batched.append({"image": i["image"], "part_instances": parts, "instances": objs, "height": i["height"], "width": i["width"]})
    images = ImageList.from_tensors([i["image"].to(DEV) for i in inputs], C.INFER["size_div"])
    targets = I.prepare_pd_gt_targets(model, batched, images)
    res = I.pd_inference(model, batched, targets, images, outputs)
    for r, want in zip(res, g):
        p = r["predictions"]
        assert p.pred_masks.shape == want["pred_masks"].shape, (p.pred_masks.shape, want["pred_masks"].shape)
        key = lambda s, c: torch.argsort(s.double() + c.double() * 1e-9)          # pair proposals by (distinct) score
        o1, o2 = key(p.scores.cpu(), p.pred_classes.cpu()), key(want["scores"], want["pred_classes"])
        assert torch.equal(p.pred_classes.cpu()[o1], want["pred_classes"][o2])
        torch.testing.assert_close(p.scores.cpu()[o1], want["scores"][o2], rtol=1e-5, atol=1e-6)
        assert (p.pred_masks.cpu()[o1] != want["pred_masks"][o2]).float().mean().item() < 2e-3
        assert int(r["gt_object_label"]) == int(want["gt_object_label"])
