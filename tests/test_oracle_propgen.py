"""CPU tests of the proposal-generation oracle (oracle/proposal_generation_ref.py): against the goldens captured from
the real reference pipeline (tests/golden/propgen.pt, real scikit-learn K-means inside) and against scikit-learn itself;
plus the COCO RLE writer's round trip."""
import numpy as np
import pytest
import torch

import common as C
from oracle import proposal_generation_ref as P


@pytest.mark.parametrize("tag,metric,norm", [("dot_0", "dot", False), ("l2_1", "l2", True)])
def test_dense_labelling_matches_reference_golden(golden, tag, metric, norm):
    g = golden("propgen")[tag]
    feats, inputs = C.make_propgen_inputs()
    cfg = C.PROPGEN
    res = P.proposal_generation(feats, inputs, ["res3", "res4"], metric, norm, cfg["size_div"], cfg["K"],
                                lambda i, data: g[i]["centroids"])
    for r, want in zip(res, g):
        binary, label_map, mask_resized = r
        assert torch.equal(mask_resized, want["object_mask_resized"])
        assert torch.equal(binary, want["pseudo_label"])


def test_lloyd_restatement_matches_sklearn():
    """same data, same initial centres -> same partition and centres as scikit-learn's KMeans(algorithm='lloyd')"""
    from sklearn.cluster import KMeans
    rng = np.random.default_rng(3)
    for N, Cc, K in [(83, 40, 4), (500, 16, 4), (40, 8, 3)]:
        blobs = rng.normal(size=(K, Cc)).astype(np.float32) * 2
        X = (blobs[rng.integers(K, size=N)] + rng.normal(size=(N, Cc)).astype(np.float32)).astype(np.float32)
        init = X[rng.choice(N, K, replace=False)].copy()
        sk = KMeans(n_clusters=K, init=init, n_init=1, random_state=0).fit(X)
        centers, labels, n_iter = P.kmeans_lloyd_np(X, init)
        assert np.array_equal(labels, sk.labels_), (N, Cc, K)
        np.testing.assert_allclose(centers, sk.cluster_centers_, rtol=1e-4, atol=1e-5)
        assert n_iter == sk.n_iter_


def test_reference_centroids_are_a_lloyd_fixed_point(golden):
    """the goldens' centroids came out of sklearn: one more Lloyd pass from them must not move the partition"""
    g = golden("propgen")["dot_0"]
    feats, inputs = C.make_propgen_inputs()
    f = P.prepare_features(feats, ["res3", "res4"], False)
    for i, want in enumerate(g):
        m = torch.nn.functional.interpolate(torch.nn.functional.pad(inputs[i]["mask"], (0, 128 - inputs[i]["mask"].shape[-1], 0, 128 - inputs[i]["mask"].shape[-2]))[None],
                                            size=f.shape[-2:], mode="nearest")[0, 0].bool()
        data = f[i][:, m].t().contiguous().numpy()
        assert data.shape[0] == want["n_points"]
        centers, labels, n_iter = P.kmeans_lloyd_np(data, want["centroids"].numpy())
        np.testing.assert_allclose(centers, want["centroids"].numpy(), rtol=1e-4, atol=1e-5)
        assert n_iter <= 2


def test_coco_rle_round_trip_and_format():
    from partdistillation_amd.utils import rle
    rng = np.random.default_rng(0)
    for shape in [(5, 7), (64, 48), (1, 1), (3, 1000)]:
        for p in (0.0, 0.3, 0.97, 1.0):
            m = rng.random(shape) < p
            r = rle.encode(m)
            assert r["size"] == list(shape) and (rle.decode(r) == m).all()
    # column-major runs, first run counts zeros: [[0,1],[1,1]] -> 1 zero, 3 ones
    assert rle.mask_to_counts(np.array([[0, 1], [1, 1]])).tolist() == [1, 3]
    assert rle.encode(np.ones((2, 2), bool))["counts"] == b"04"
    big = rle.counts_to_string([0, 70000, 5, 69990])          # multi-group counts and a negative difference
    assert rle.string_to_counts(big).tolist() == [0, 70000, 5, 69990]
    labels = np.array([[0, 1, 1], [2, 2, 0]], dtype=np.uint8)
    js = rle.labels_to_coco_json(labels, [1, 2])
    assert [(rle.decode({"size": j["segmentation"]["size"], "counts": j["segmentation"]["counts"]}) == (labels == l)).all()
            for j, l in zip(js, [1, 2])] == [True, True]


@pytest.mark.parametrize("tag,unique,min_score", [("unique_1", True, -1.0), ("unique_0", False, 0.3)])
def test_inference_oracle_matches_reference_golden(golden, tag, unique, min_score):
    from oracle import inference_ref as I
    g = golden("infer")[tag]
    outputs, inputs = C.make_infer_inputs()
    res = I.inference(outputs, inputs, (128, 128), C.INFER["topk"], unique, 0.02, min_score)
    for (masks, scores, labels, gt), want in zip(res, g):
        assert torch.equal(masks, want["pred_masks"]) and torch.equal(labels, want["pred_classes"])
        torch.testing.assert_close(scores, want["scores"], rtol=0, atol=0)
        assert torch.equal(gt, want["gt_masks"])


@pytest.mark.parametrize("tag,mode,unique,min_score,oracle_cls", [("raw_1", "", True, -1.0, False), ("eval_1", "eval", True, -1.0, False),
                                                                  ("eval_0", "eval", False, 0.05, True)])
def test_part_distillation_inference_oracle_matches_reference_golden(golden, tag, mode, unique, min_score, oracle_cls):
    from oracle import inference_ref as I
    g = golden("infer_pd")[tag]
    outputs, inputs = C.make_infer_inputs()
    K = C.INFER_PD_CLASSES
    outputs = dict(outputs, pred_logits=C.seeded((len(inputs), C.INFER["Q"], K + 1), 5300) * 2)
    mapping = {3: torch.tensor(C.INFER_PD_MAPPING[0]), 4: torch.tensor(C.INFER_PD_MAPPING[1])} if mode == "eval" else None
    res = I.inference_pd(outputs, inputs, [3, 4], (128, 128), K, C.INFER["topk"] * 2, unique, 0.02, min_score, mapping,
                         oracle_classifier=oracle_cls)
    for (masks, scores, labels), want in zip(res, g):
        assert torch.equal(masks, want["pred_masks"]) and torch.equal(labels, want["pred_classes"])
        torch.testing.assert_close(scores, want["scores"], rtol=0, atol=0)
