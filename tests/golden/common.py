"""Shared by tests/golden/make_golden.py (build container, imports the
reference) and the tests (both boxes, never touch the reference): deterministic
weights, replayable random draws, tensor digests and the small case configs.

Weights are NOT stored in the fixtures: they are regenerated from a seed and
the (name -> shape, dtype) table stored in the fixture, with the CPU torch
generator (bit-stable across the two boxes: same image, same torch).
"""
import zlib

import torch

_TORCH_RAND = torch.rand  # bound early: make_golden.py monkey-patches torch.rand


def _scale_for(name, shape):
    last = name.rsplit(".", 1)[-1]
    if "running_var" in name:
        return "var"
    if "running_mean" in name:
        return 0.1
    if "relative_position_index" in name or "num_batches_tracked" in name:
        return "keep"
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        if "sampling_offsets" in name:
            return 0.3 / fan_in ** 0.5
        if "embed.weight" in name or "query_feat" in name or "level_embed" in name or "bias_table" in name:
            return 0.5
        return 1.0 / fan_in ** 0.5
    if last == "weight":
        return "norm"
    if "sampling_offsets.bias" in name:
        return 1.5
    return 0.05


def seeded_weights(table, seed):
    """table: {name: (shape tuple, dtype str)} -> {name: tensor}."""
    out = {}
    for name in sorted(table):
        shape, dt = table[name]
        dtype = getattr(torch, dt)
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
        sc = _scale_for(name, shape)
        if sc == "keep":
            continue
        r = torch.randn(tuple(shape), generator=g, dtype=torch.float64)
        if sc == "var":
            t = r.abs() * 0.5 + 0.5
        elif sc == "norm":
            t = 1.0 + 0.1 * r
        else:
            t = r * sc
        out[name] = t.to(dtype)
    return out


def table_of(state_dict):
    return {k: (tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in state_dict.items()}


class ReplayRand:
    """rand(shape) -> uniform [0,1) float32; call k uses generator seed+k."""

    def __init__(self, seed):
        self.seed = seed
        self.calls = 0

    def __call__(self, *shape, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        g = torch.Generator().manual_seed(self.seed + self.calls)
        self.calls += 1
        return _TORCH_RAND(shape, generator=g, dtype=torch.float32)


def seeded(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(tuple(shape), generator=g, dtype=torch.float64) * scale).to(dtype)


FULL_LIMIT = 1 << 20      # bytes: tensors up to this size are pinned element by element ("full"), larger ones by the strided sample only


def digest(t, n=2048):
    """Pin of a tensor: shape, sums, a strided sample and — up to FULL_LIMIT bytes — every element."""
    t = t.detach().cpu()
    if t.dtype == torch.bool:
        t = t.to(torch.uint8)
    f = t.reshape(-1)
    stride = max(1, f.numel() // n)
    d = {"shape": torch.tensor(list(t.shape), dtype=torch.int64),
         "sum": f.double().sum().reshape(1), "abssum": f.double().abs().sum().reshape(1),
         "sample": f[::stride][:n].clone()}
    if 0 < f.numel() * f.element_size() <= FULL_LIMIT:
        d["full"] = f.clone()
    return d


def check_digest(t, d, rtol, atol, what=""):
    got = digest(t)
    assert got["shape"].tolist() == d["shape"].tolist(), f"{what}: shape {got['shape'].tolist()} != {d['shape'].tolist()}"
    if "full" in d:                          # every element (fixtures regenerated in round 6)
        torch.testing.assert_close(t.detach().cpu().reshape(-1).double(), d["full"].double(), rtol=rtol, atol=atol, msg=lambda m: f"{what} (element-wise): {m}")
    torch.testing.assert_close(got["sample"].double(), d["sample"].double(), rtol=rtol, atol=atol, msg=lambda m: f"{what} sample: {m}")
    n = max(1, t.numel())
    torch.testing.assert_close(got["abssum"], d["abssum"], rtol=max(rtol, 1e-6), atol=atol * n, msg=lambda m: f"{what} abssum: {m}")


def check_digest_scaled(t, d, frac, what=""):
    """error measured against the tensor's own scale: max|got - want| <= frac * max|want| (deep gradient chains)."""
    got = digest(t)
    assert got["shape"].tolist() == d["shape"].tolist(), f"{what}: shape"
    full = "full" in d
    want = (d["full"] if full else d["sample"]).double()
    have = (t.detach().cpu().reshape(-1).to(torch.uint8 if t.dtype == torch.bool else t.dtype) if full else got["sample"]).double()
    err = (have - want).abs().max().item()
    scale = max(want.abs().max().item(), 1e-30)
    assert err <= frac * scale, f"{what}: max abs err {err:.3e} > {frac} * scale {scale:.3e}" + (" (element-wise)" if full else "")
    assert abs(got["abssum"].item() - d["abssum"].item()) <= frac * d["abssum"].item() + 1e-30, f"{what}: abssum"


# ----------------------------------------------------------------------------- case configs
TINY = dict(conv_dim=64, mask_dim=64, nheads=8, enc_layers=2, enc_ffn=1024, channels=(16, 32, 64, 128),
            image=128, batch=1, queries=12, dec_layers=3, dec_ffn=128, num_classes=1,
            num_points=96, oversample=3.0, importance=0.75, n_targets=3)

C1 = dict(conv_dim=256, mask_dim=256, nheads=8, enc_layers=6, enc_ffn=1024, channels=(256, 512, 1024, 2048),
          image=256, batch=1, queries=100, dec_layers=9, dec_ffn=2048, num_classes=1,
          num_points=12544, oversample=3.0, importance=0.75, n_targets=4)

SWIN_TINY = dict(pretrain_img_size=96, patch_size=4, embed_dim=24, depths=(2, 2, 2, 2), num_heads=(2, 2, 4, 4),
                 window_size=4, image=(72, 88), batch=2)

# the geometry BASELINE configs 3 / 5 run (window 12, head_dim 32 in every stage -> pd_window_attn_*_w12 and the fused Swin
# stage apply): 100 x 132 image -> 25 x 33 tokens, so every stage pads to the window, shifted blocks straddle region borders,
# PatchMerging pads odd maps
SWIN_W12 = dict(pretrain_img_size=96, patch_size=4, embed_dim=64, depths=(2, 2, 2, 2), num_heads=(2, 4, 8, 16),
                window_size=12, image=(100, 132), batch=2)


def make_features(cfg, seed):
    s = cfg["image"]
    return {f"res{i + 2}": seeded((cfg["batch"], c, s // st, s // st), seed + i)
            for i, (c, st) in enumerate(zip(cfg["channels"], (4, 8, 16, 32)))}


def make_targets(cfg, seed, size=None):
    """n disjoint blob masks per image (bool [n,S,S]) + zero labels."""
    s = size or cfg["image"]
    n = cfg["n_targets"]
    out = []
    for b in range(cfg["batch"]):
        g = torch.Generator().manual_seed(seed + 17 * b)
        centers = torch.rand((n, 2), generator=g) * 0.5 + 0.25
        ys, xs = torch.meshgrid(torch.arange(s) / s, torch.arange(s) / s, indexing="ij")
        d = torch.stack([(ys - c[0]) ** 2 + (xs - c[1]) ** 2 for c in centers])
        inside = ((ys - 0.5) ** 2 / 0.16 + (xs - 0.5) ** 2 / 0.1) < 1.0
        lab = d.argmin(0)
        masks = torch.stack([(lab == k) & inside for k in range(n)])
        out.append({"labels": torch.zeros(n, dtype=torch.int64), "masks": masks})
    return out


# ----------------------------------------------------------------------------- proposal generation (BASELINE config 4)
PROPGEN = dict(C3=16, C4=24, K=4, size_div=32, images=[(128, 128, 128, 128), (112, 128, 96, 110)])   # (H, W, out_h, out_w)


def make_propgen_inputs(cfg=PROPGEN, seed=4100):
    """synthetic backbone features (smooth fields + noise), one elliptical object mask per image, image / output sizes"""
    import torch.nn.functional as F
    Hp = max(i[0] for i in cfg["images"])
    Wp = max(i[1] for i in cfg["images"])
    d = cfg["size_div"]
    Hp, Wp = (Hp + d - 1) // d * d, (Wp + d - 1) // d * d
    B = len(cfg["images"])
    feats = {}
    for key, ch, stride, s0 in (("res3", cfg["C3"], 8, seed), ("res4", cfg["C4"], 16, seed + 1)):
        h, w = Hp // stride, Wp // stride
        base = F.interpolate(seeded((B, ch, 3, 3), s0), size=(h, w), mode="bilinear", align_corners=False)
        feats[key] = base + 0.15 * seeded((B, ch, h, w), s0 + 10)
    inputs = []
    for b, (H, W, oh, ow) in enumerate(cfg["images"]):
        ys, xs = torch.meshgrid(torch.arange(H) / H, torch.arange(W) / W, indexing="ij")
        m = ((ys - 0.5) ** 2 / 0.12 + (xs - 0.45) ** 2 / 0.09) < 1.0
        inputs.append({"image": seeded((3, H, W), seed + 20 + b) * 50 + 100, "mask": m[None].float(),   # float 0/1: the reference resizes it bilinearly
                       "height": oh, "width": ow,
                       "file_name": f"img{b}.pth", "file_path": f"/nowhere/img{b}.JPEG", "class_code": "n000", "class_name": "thing",
                       "gt_object_class": 7})
    return feats, inputs


# ----------------------------------------------------------------------------- inference branch
INFER = dict(Q=12, topk=8, low=32, size_div=32, images=[(96, 128, 96, 128), (128, 112, 100, 90)])   # (H, W, out_h, out_w)


def make_infer_inputs(cfg=INFER, seed=5200):
    """decoder outputs (class logits [B,Q,2], low-resolution mask logits [B,Q,low,low] made of smooth blobs) and per-image
    ground truth: 3 part masks + labels (`part_instances`) and the object mask (`instances`)"""
    import torch.nn.functional as F
    B, Q, low = len(cfg["images"]), cfg["Q"], cfg["low"]
    logits = seeded((B, Q, 2), seed) * 2
    base = F.interpolate(seeded((B, Q, 5, 5), seed + 1) * 3, size=(low, low), mode="bilinear", align_corners=False)
    masks = base + 0.3 * seeded((B, Q, low, low), seed + 2) - 0.8
    inputs = []
    for b, (H, W, oh, ow) in enumerate(cfg["images"]):
        ys, xs = torch.meshgrid(torch.arange(H) / H, torch.arange(W) / W, indexing="ij")
        inside = ((ys - 0.5) ** 2 / 0.17 + (xs - 0.5) ** 2 / 0.12) < 1.0
        g = torch.Generator().manual_seed(seed + 30 + b)
        centers = torch.rand((3, 2), generator=g) * 0.5 + 0.25
        lab = torch.stack([(ys - c[0]) ** 2 + (xs - c[1]) ** 2 for c in centers]).argmin(0)
        parts = torch.stack([(lab == k) & inside for k in range(3)])
        inputs.append({"image": seeded((3, H, W), seed + 40 + b) * 50 + 100, "part_masks": parts, "part_labels": torch.tensor([2, 0, 1]),
                       "object_mask": inside[None], "height": oh, "width": ow})
    return {"pred_logits": logits, "pred_masks": masks}, inputs


INFER_PD_CLASSES = 4                                                         # part classes of the class-aware evaluation goldens
INFER_PD_MAPPING = ([2, 0, 0, 1], [1, 1, 3, 0])                               # majority-vote class mapping of object classes 3 and 4


# ----------------------------------------------------------------------------- part ranking
RANK = dict(C=16, clusters=3, object_classes=(3, 4), n_feats=(40, 2, 25))       # pooled proposals of object classes 3, 4, 5: class 4 has too few (2 <= clusters)


def make_rank_features(cfg=RANK, seed=5400):
    """decoder_output stand-in [B, Q, C] for the INFER images, and a pool of per-proposal features + object labels for the
    clustering module: classes 3 and 5 are mixtures of `clusters` well separated blobs, class 4 has only 2 proposals"""
    feats = seeded((len(INFER["images"]), INFER["Q"], cfg["C"]), seed)
    pool, labels = [], []
    for i, (cid, n) in enumerate(zip((3, 4, 5), cfg["n_feats"])):
        centers = seeded((cfg["clusters"], cfg["C"]), seed + 10 + i) * 4
        which = torch.arange(n) % cfg["clusters"]
        pool.append(centers[which] + 0.2 * seeded((n, cfg["C"]), seed + 20 + i))
        labels.append(torch.full((n,), cid))
    return feats, pool, labels


def rank_centroids(cfg=RANK, seed=5450):
    """classifier centroids registered for object classes 3 and 4 ([clusters, C])"""
    import torch.nn.functional as F
    return {cid: F.normalize(seeded((cfg["clusters"], cfg["C"]), seed + cid), dim=-1) * 3 for cid in cfg["object_classes"]}


RANK_MAPPING = ([2, 0, 1], [1, 1, 0])                                          # majority-vote mapping of object classes 3 / 4


# ----------------------------------------------------------------------------- meta-architecture train branch
META = dict(TINY, batch=2, images=[(128, 128), (96, 128)], n_parts=(3, 2), part=(5, 4))   # ragged sizes, ragged #masks; (N_obj, K)


def stub_backbone_weights(cfg, seed=6100):
    return [seeded((c, 3, 1, 1), seed + i, 0.6) for i, c in enumerate(cfg["channels"])]


def stub_backbone(x, weights):
    """stand-in backbone shared by the reference run (make_golden.gen_meta), the oracle and the product tests: res{2..5} =
    1x1 conv (seeded) of the stride-s average pool of the NORMALISED, PADDED image batch — so the features carry the
    meta-architecture's normalisation and ImageList padding."""
    import torch.nn.functional as F
    return {f"res{i + 2}": F.conv2d(F.avg_pool2d(x, s), w.to(x)) for i, (s, w) in enumerate(zip((4, 8, 16, 32), weights))}


def make_meta_inputs(cfg=META, seed=6200):
    """per image: uint8 image [3,H,W], n disjoint bool part masks [n,H,W] inside an ellipse, part labels (distinct, < K),
    the object class (< N_obj)"""
    out = []
    for b, ((H, W), n) in enumerate(zip(cfg["images"], cfg["n_parts"])):
        g = torch.Generator().manual_seed(seed + b)
        img = torch.randint(0, 256, (3, H, W), generator=g, dtype=torch.uint8)
        ys, xs = torch.meshgrid(torch.arange(H) / H, torch.arange(W) / W, indexing="ij")
        inside = ((ys - 0.5) ** 2 / 0.16 + (xs - 0.5) ** 2 / 0.1) < 1.0
        centers = torch.rand((n, 2), generator=g) * 0.5 + 0.25
        lab = torch.stack([(ys - c[0]) ** 2 + (xs - c[1]) ** 2 for c in centers]).argmin(0)
        masks = torch.stack([(lab == k) & inside for k in range(n)])
        out.append({"image": img, "masks": masks, "gt_classes": torch.randperm(cfg["part"][1], generator=g)[:n],
                    "gt_object_class": (1, 3)[b % 2], "height": H, "width": W})
    return out
