"""ORACLE — test infrastructure only (also bench.py's ``cpu_baseline`` leg).

Plain-PyTorch, CPU, fp32 *functional* restatement of the reference's
Mask2Former part-proposal / part-distillation training step.
Everything takes a flat ``sd`` (state_dict with the reference's key names,
SURVEY Appendix B) so the same weights drive the oracle and the HIP product
path.  Random draws (matcher / criterion point coordinates) come from an
injected ``rand(shape) -> Tensor in [0,1)`` so tests can replay them.

Pinned against goldens captured from the real reference modules
(tests/golden/make_golden.py) in tests/test_oracle_*.py.  "Parity unpinned by
the reference's own tests" applies to: the ResNet-50 arithmetic (detectron2
0.6, un-vendored), ImageList padding, point_sample, LSA tie-breaking — those
follow SURVEY Appendix D and are pinned by our own goldens only.

Each function cites the reference file:line it follows (paths relative to
/root/reference/part_distillation/).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import clib
from .msda import msda_torch


# ----------------------------------------------------------------------------- helpers
def linear(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def layer_norm(sd, p, x, eps=1e-5):
    return F.layer_norm(x, x.shape[-1:], sd[p + ".weight"], sd[p + ".bias"], eps)


def group_norm(sd, p, x, groups=32, eps=1e-5):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def sine_pos_embed(b, h, w, num_pos_feats, temperature=10000.0, scale=2 * math.pi, dtype=torch.float32):
    """modeling/transformer_decoder/position_encoding.py:33-56 with mask=None,
    normalize=True.  Returns [b, 2*num_pos_feats, h, w]."""
    eps = 1e-6
    y = torch.arange(1, h + 1, dtype=torch.float32).view(h, 1).expand(h, w)
    x = torch.arange(1, w + 1, dtype=torch.float32).view(1, w).expand(h, w)
    y = y / (float(h) + eps) * scale
    x = x / (float(w) + eps) * scale
    i = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / num_pos_feats)
    px = x[:, :, None] / dim_t
    py = y[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    pos = torch.cat((py, px), dim=2).permute(2, 0, 1)
    return pos.unsqueeze(0).expand(b, -1, -1, -1).to(dtype)


def point_sample(inp, coords):
    """detectron2 point_sample: coords [N,P,2] in [0,1] (x,y),
    bilinear, zero padding, align_corners=False. -> [N,C,P]"""
    return F.grid_sample(inp, 2.0 * coords.unsqueeze(2) - 1.0, mode="bilinear",
                         padding_mode="zeros", align_corners=False).squeeze(3)


def mha(sd, p, query, key, value, nheads, attn_mask=None):
    """nn.MultiheadAttention forward (seq-first [L,B,C]), dropout 0, bool
    attn_mask [B*h, Lq, Lk] with True = blocked.  Used by
    mask2former_transformer_decoder.py:49-50,107-110."""
    Lq, B, C = query.shape
    Lk = key.shape[0]
    hd = C // nheads
    w, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q = F.linear(query, w[:C], b[:C])
    k = F.linear(key, w[C:2 * C], b[C:2 * C])
    v = F.linear(value, w[2 * C:], b[2 * C:])
    q = q.reshape(Lq, B * nheads, hd).transpose(0, 1) * (hd ** -0.5)
    k = k.reshape(Lk, B * nheads, hd).transpose(0, 1)
    v = v.reshape(Lk, B * nheads, hd).transpose(0, 1)
    s = torch.bmm(q, k.transpose(1, 2))
    if attn_mask is not None:
        s = s.masked_fill(attn_mask, float("-inf"))
    a = torch.softmax(s, dim=-1)
    o = torch.bmm(a, v).transpose(0, 1).reshape(Lq, B, C)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


# ----------------------------------------------------------------------------- pixel decoder
def msdeform_attn_layer(sd, p, query, ref_pts, src, shapes, nheads, npoints):
    """pixel_decoder/ops/modules/ms_deform_attn.py:86-131 (2-d reference points)."""
    N, Lq, C = query.shape
    L = shapes.shape[0]
    S = src.shape[1]
    value = linear(sd, p + ".value_proj", src).view(N, S, nheads, C // nheads)
    off = linear(sd, p + ".sampling_offsets", query).view(N, Lq, nheads, L, npoints, 2)
    aw = linear(sd, p + ".attention_weights", query).view(N, Lq, nheads, L * npoints)
    aw = torch.softmax(aw, -1).view(N, Lq, nheads, L, npoints)
    normalizer = torch.stack([shapes[:, 1], shapes[:, 0]], -1).to(off.dtype)
    loc = ref_pts[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    out = msda_torch(value, shapes, loc, aw)
    return linear(sd, p + ".output_proj", out)


def encoder_reference_points(shapes, b):
    """msdeformattn.py:145-157 with valid_ratios == 1: pixel centres / (W,H)."""
    refs = []
    for h, w in shapes.tolist():
        ys = torch.linspace(0.5, h - 0.5, h, dtype=torch.float32)
        xs = torch.linspace(0.5, w - 0.5, w, dtype=torch.float32)
        ry, rx = torch.meshgrid(ys, xs, indexing="ij")
        refs.append(torch.stack((rx.reshape(-1) / w, ry.reshape(-1) / h), -1))
    ref = torch.cat(refs, 0)[None].expand(b, -1, -1)
    return ref[:, :, None, :].expand(-1, -1, len(shapes), -1)


def pixel_decoder_forward(sd, p, features, *, nheads=8, npoints=4, enc_layers=6,
                          transformer_in=("res3", "res4", "res5"), common_stride=4, strides=None):
    """MSDeformAttnPixelDecoder.forward_features, pixel_decoder/msdeformattn.py:318-362.
    features: dict res2..res5 (NCHW).  Returns (mask_features, enc_out0, multi_scale[3])."""
    strides = strides or {"res2": 4, "res3": 8, "res4": 16, "res5": 32}
    tin = sorted(transformer_in, key=lambda k: strides[k])
    conv_dim = sd[p + ".input_proj.0.0.weight"].shape[0]
    srcs, poss = [], []
    for idx, f in enumerate(tin[::-1]):                                   # low-res first (:323)
        x = features[f].float()
        y = F.conv2d(x, sd[f"{p}.input_proj.{idx}.0.weight"], sd[f"{p}.input_proj.{idx}.0.bias"])
        srcs.append(group_norm(sd, f"{p}.input_proj.{idx}.1", y))
        poss.append(sine_pos_embed(x.shape[0], x.shape[2], x.shape[3], conv_dim // 2))
    # MSDeformAttnTransformerEncoderOnly.forward :65-93
    b = srcs[0].shape[0]
    shapes = torch.tensor([[s.shape[2], s.shape[3]] for s in srcs], dtype=torch.long)
    lvl_embed = sd[p + ".transformer.level_embed"]
    src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
    pos = torch.cat([q.flatten(2).transpose(1, 2) + lvl_embed[i].view(1, 1, -1) for i, q in enumerate(poss)], 1)
    ref = encoder_reference_points(shapes, b)
    out = src
    for i in range(enc_layers):                                           # EncoderLayer.forward :126-135
        lp = f"{p}.transformer.encoder.layers.{i}"
        a = msdeform_attn_layer(sd, lp + ".self_attn", out + pos, ref, out, shapes, nheads, npoints)
        out = layer_norm(sd, lp + ".norm1", out + a)
        f2 = linear(sd, lp + ".linear2", F.relu(linear(sd, lp + ".linear1", out)))
        out = layer_norm(sd, lp + ".norm2", out + f2)
    sizes = [int(h * w) for h, w in shapes.tolist()]
    outs = [z.transpose(1, 2).reshape(b, -1, int(shapes[i, 0]), int(shapes[i, 1]))
            for i, z in enumerate(out.split(sizes, dim=1))]
    # extra FPN levels (:347-355)
    all_feats = sorted(features.keys(), key=lambda k: strides[k])
    n_fpn = int(np.log2(min(strides[k] for k in tin)) - np.log2(common_stride))
    for idx, f in enumerate(all_feats[:n_fpn][::-1]):
        k = n_fpn - idx                                                   # adapter_{k} / layer_{k}
        x = features[f].float()
        lat = group_norm(sd, f"{p}.adapter_{k}.norm", F.conv2d(x, sd[f"{p}.adapter_{k}.weight"]))
        y = lat + F.interpolate(outs[-1], size=lat.shape[-2:], mode="bilinear", align_corners=False)
        y = F.relu(group_norm(sd, f"{p}.layer_{k}.norm", F.conv2d(y, sd[f"{p}.layer_{k}.weight"], padding=1)))
        outs.append(y)
    mask_features = F.conv2d(outs[-1], sd[p + ".mask_features.weight"], sd[p + ".mask_features.bias"])
    return mask_features, outs[0], outs[:3]


def base_pixel_decoder_forward(sd, p, features, strides=None):
    """BasePixelDecoder.forward_features, pixel_decoder/fpn.py:140-158 (norm "GN": no conv bias, GroupNorm(32); nearest
    top-down upsampling :153; 3 x 3 mask_features :119-127).  -> (mask_features, None, multi_scale[3])"""
    strides = strides or {"res2": 4, "res3": 8, "res4": 16, "res5": 32}
    names = sorted(features.keys(), key=lambda k: strides[k])
    n = len(names)
    y, ms = None, []
    for idx, f in enumerate(names[::-1]):
        k = n - idx                                                        # adapter_{k} / layer_{k}
        x = features[f]
        if idx == 0:
            y = F.relu(group_norm(sd, f"{p}.layer_{k}.norm", F.conv2d(x, sd[f"{p}.layer_{k}.weight"], padding=1)))
        else:
            cur = group_norm(sd, f"{p}.adapter_{k}.norm", F.conv2d(x, sd[f"{p}.adapter_{k}.weight"]))
            y = cur + F.interpolate(y, size=cur.shape[-2:], mode="nearest")
            y = F.relu(group_norm(sd, f"{p}.layer_{k}.norm", F.conv2d(y, sd[f"{p}.layer_{k}.weight"], padding=1)))
        if len(ms) < 3:
            ms.append(y)
    return F.conv2d(y, sd[p + ".mask_features.weight"], sd[p + ".mask_features.bias"], padding=1), None, ms


# ----------------------------------------------------------------------------- transformer decoder
def mask_embed_mlp(sd, p, x):
    """MLP, mask2former_transformer_decoder.py:196-208 (3 layers)."""
    x = F.relu(linear(sd, p + ".layers.0", x))
    x = F.relu(linear(sd, p + ".layers.1", x))
    return linear(sd, p + ".layers.2", x)


def prediction_heads(sd, p, output, mask_features, target_size, nheads, part=None):
    """forward_prediction_heads, mask2former_transformer_decoder.py:441-459;
    part-distillation variant part_distillation_transformer_decoder.py:215-254
    when ``part=(targets, num_part_classes)``."""
    dec = layer_norm(sd, p + ".decoder_norm", output).transpose(0, 1)          # [B,Q,C]
    if part is None:
        cls = linear(sd, p + ".class_embed", dec)
    else:
        targets, K = part
        full = F.linear(dec.double(), sd[p + ".class_embed.weight"], sd[p + ".class_embed.bias"])
        rows = [full[i][:, int(t["gt_object_class"]) * K:(int(t["gt_object_class"]) + 1) * K]
                for i, t in enumerate(targets)]
        cls = torch.cat([torch.stack(rows, 0), full[:, :, -1:]], dim=-1) + full.sum() * 0
    me = mask_embed_mlp(sd, p + ".mask_embed", dec)
    masks = torch.einsum("bqc,bchw->bqhw", me, mask_features)
    am = F.interpolate(masks, size=target_size, mode="bilinear", align_corners=False)
    am = (am.sigmoid().flatten(2).unsqueeze(1).repeat(1, nheads, 1, 1).flatten(0, 1) < 0.5).bool().detach()
    return cls, masks, am, dec


def decoder_forward(sd, p, multi_scale, mask_features, *, nheads=8, dec_layers=9, part=None):
    """MultiScaleMaskedTransformerDecoder.forward, mask2former_transformer_decoder.py:370-439
    (input_proj = identity: in_channels == hidden_dim, :331-335)."""
    C = sd[p + ".query_feat.weight"].shape[1]
    src, pos, sizes = [], [], []
    for i, x in enumerate(multi_scale):
        sizes.append(x.shape[-2:])
        pos.append(sine_pos_embed(x.shape[0], x.shape[2], x.shape[3], C // 2).flatten(2).permute(2, 0, 1))
        s = x.flatten(2) + sd[p + ".level_embed.weight"][i][None, :, None]
        src.append(s.permute(2, 0, 1))
    bs = src[0].shape[1]
    qpos = sd[p + ".query_embed.weight"].unsqueeze(1).repeat(1, bs, 1)
    out = sd[p + ".query_feat.weight"].unsqueeze(1).repeat(1, bs, 1)
    classes, masks = [], []
    cls, msk, am, dec = prediction_heads(sd, p, out, mask_features, sizes[0], nheads, part)
    classes.append(cls), masks.append(msk)
    for i in range(dec_layers):
        lvl = i % 3
        am = am.clone()
        am[torch.where(am.sum(-1) == am.shape[-1])] = False                 # :405
        cp = f"{p}.transformer_cross_attention_layers.{i}"
        t2 = mha(sd, cp + ".multihead_attn", out + qpos, src[lvl] + pos[lvl], src[lvl], nheads, am)
        out = layer_norm(sd, cp + ".norm", out + t2)
        sp = f"{p}.transformer_self_attention_layers.{i}"
        t2 = mha(sd, sp + ".self_attn", out + qpos, out + qpos, out, nheads)
        out = layer_norm(sd, sp + ".norm", out + t2)
        fp = f"{p}.transformer_ffn_layers.{i}"
        t2 = linear(sd, fp + ".linear2", F.relu(linear(sd, fp + ".linear1", out)))
        out = layer_norm(sd, fp + ".norm", out + t2)
        cls, msk, am, dec = prediction_heads(sd, p, out, mask_features, sizes[(i + 1) % 3], nheads, part)
        classes.append(cls), masks.append(msk)
    res = {"pred_logits": classes[-1], "pred_masks": masks[-1], "decoder_output": dec,
           "aux_outputs": [{"pred_logits": a, "pred_masks": b} for a, b in zip(classes[:-1], masks[:-1])]}
    if part is not None:
        res["query_feats"] = out.permute(1, 0, 2)
    return res


# ----------------------------------------------------------------------------- matcher / criterion
def lsa(cost):
    """scipy.optimize.linear_sum_assignment restated in C (oracle/lsa_ref.c)."""
    import ctypes
    c = np.ascontiguousarray(cost.detach().cpu().double().numpy())
    nr, nc = c.shape
    k = min(nr, nc)
    r, cc = np.zeros(k, np.int64), np.zeros(k, np.int64)
    st = clib.lib().pd_oracle_lsa(ctypes.c_int(nr), ctypes.c_int(nc), c.ctypes.data_as(ctypes.c_void_p),
                                  r.ctypes.data_as(ctypes.c_void_p), cc.ctypes.data_as(ctypes.c_void_p))
    if st != 0:
        raise ValueError("cost matrix is infeasible")
    return r, cc


def matcher_cost(logits, pred_masks, labels, tgt_masks, coords, w_class, w_mask, w_dice):
    """Cost matrix of one image, modeling/matcher.py:108-158.  coords [1,P,2]."""
    prob = logits.sigmoid() if logits.shape[-1] == 1 else logits.softmax(-1)
    cost_class = -prob[:, labels]
    tgt = point_sample(tgt_masks[:, None].to(pred_masks), coords.repeat(tgt_masks.shape[0], 1, 1)).squeeze(1).float()
    out = point_sample(pred_masks[:, None], coords.repeat(pred_masks.shape[0], 1, 1)).squeeze(1).float()
    P = out.shape[1]
    pos = F.binary_cross_entropy_with_logits(out, torch.ones_like(out), reduction="none")
    neg = F.binary_cross_entropy_with_logits(out, torch.zeros_like(out), reduction="none")
    cost_mask = (torch.einsum("nc,mc->nm", pos, tgt) + torch.einsum("nc,mc->nm", neg, 1 - tgt)) / P
    sg = out.sigmoid()
    num = 2 * torch.einsum("nc,mc->nm", sg, tgt)
    den = sg.sum(-1)[:, None] + tgt.sum(-1)[None, :]
    cost_dice = 1 - (num + 1) / (den + 1)
    return w_mask * cost_mask + w_class * cost_class + w_dice * cost_dice


@torch.no_grad()
def hungarian_match(outputs, targets, rand, *, w_class, w_mask, w_dice, num_points, costs=None):
    """HungarianMatcher.memory_efficient_forward, modeling/matcher.py:100-168.  ``costs``: optional list that receives
    every image's [Q, n] cost matrix (for the tests' "is this assignment optimal under the oracle's costs" check)."""
    res = []
    for b in range(outputs["pred_logits"].shape[0]):
        coords = rand((1, num_points, 2))
        C = matcher_cost(outputs["pred_logits"][b], outputs["pred_masks"][b], targets[b]["labels"],
                         targets[b]["masks"], coords, w_class, w_mask, w_dice)
        C = C.reshape(C.shape[0], -1).cpu()
        if costs is not None:
            costs.append(C.clone())
        row, col = lsa(C)
        order = C[row, col].topk(len(row), largest=False)[1]                 # :162
        res.append((torch.as_tensor(row[order.numpy()], dtype=torch.int64).reshape(-1),
                    torch.as_tensor(col[order.numpy()], dtype=torch.int64).reshape(-1)))
    return res


def uncertain_points(logits, rand, num_points, oversample, importance):
    """detectron2 get_uncertain_point_coords_with_randomness
    with uncertainty = -|logit| (criterion.py:77-91)."""
    n = logits.shape[0]
    ns = int(num_points * oversample)
    coords = rand((n, ns, 2))
    unc = -point_sample(logits, coords).abs()
    k = int(importance * num_points)
    idx = torch.topk(unc[:, 0, :], k=k, dim=1)[1]
    idx = idx + ns * torch.arange(n, dtype=torch.long)[:, None]
    picked = coords.view(-1, 2)[idx.view(-1)].view(n, k, 2)
    if num_points - k > 0:
        picked = torch.cat([picked, rand((n, num_points - k, 2))], dim=1)
    return picked


def loss_labels(logits, targets, indices, num_classes, empty_weight):
    """criterion.py:126-145."""
    logits = logits.float()
    if sum(len(s) for s, _ in indices) == 0 and len(indices) == 0:
        return logits.sum() * 0.0
    bidx = torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(indices)])
    sidx = torch.cat([s for s, _ in indices])
    tcls = torch.full(logits.shape[:2], num_classes, dtype=torch.int64)
    tcls[bidx, sidx] = torch.cat([t["labels"][j] for t, (_, j) in zip(targets, indices)])
    return F.cross_entropy(logits.transpose(1, 2), tcls, empty_weight)


def loss_masks(pred_masks, targets, indices, num_masks, rand, num_points, oversample, importance, points_override=None, points_out=None):
    """criterion.py:147-207 (+ nested_tensor padding utils/misc.py:52-74).  Test hooks: ``points_out`` (list) receives the sample
    points this head chose; ``points_override`` ([N, P, 2]) replaces them for the losses — the sampler still runs (same random draws)."""
    bidx = torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(indices)])
    sidx = torch.cat([s for s, _ in indices])
    tidx = torch.cat([j for _, j in indices])
    src = pred_masks[bidx, sidx][:, None]
    nmax = max(t["masks"].shape[0] for t in targets)
    hmax = max(t["masks"].shape[1] for t in targets)
    wmax = max(t["masks"].shape[2] for t in targets)
    padded = torch.zeros((len(targets), nmax, hmax, wmax), dtype=targets[0]["masks"].dtype)
    for i, t in enumerate(targets):
        m = t["masks"]
        padded[i, :m.shape[0], :m.shape[1], :m.shape[2]] = m
    tgt = padded.to(src)[bidx, tidx][:, None]
    with torch.no_grad():
        coords = uncertain_points(src, rand, num_points, oversample, importance)
        if points_out is not None:
            points_out.append(coords)
        if points_override is not None:
            coords = points_override.to(coords)
        labels = point_sample(tgt, coords).squeeze(1)
    logits = point_sample(src, coords).squeeze(1)
    bce = F.binary_cross_entropy_with_logits(logits, labels, reduction="none").mean(1).sum() / num_masks
    sg = logits.sigmoid()
    dice = (1 - (2 * (sg * labels).sum(-1) + 1) / (sg.sum(-1) + labels.sum(-1) + 1)).sum() / num_masks
    return bce, dice


def set_criterion(outputs, targets, rand, *, num_classes, eos_coef=0.1, w_class=2.0, w_mask=5.0, w_dice=5.0,
                  num_points=12544, oversample=3.0, importance=0.75, world_size=1, num_masks_total=None,
                  match_points=None, return_indices=False, indices_override=None, costs=None, points_override=None, points_out=None):
    """SetCriterion.forward, criterion.py:235-270: match + CE + point BCE/dice
    for the final output and each aux output.  Unweighted losses (30 keys).
    Test hooks: ``costs`` (list) receives the per-(head, image) cost matrices; ``indices_override`` (list per head of
    per-image (rows, cols)) replaces the assignment the losses are computed with — the matcher still runs (same random
    draws, its own optimum is what ``return_indices`` hands back); ``points_override`` / ``points_out`` (lists per head) do the
    same for the importance-sampled loss points (see loss_masks)."""
    empty_weight = torch.ones(num_classes + 1)
    empty_weight[-1] = eos_coef
    nm = float(sum(len(t["labels"]) for t in targets)) if num_masks_total is None else float(num_masks_total)
    num_masks = max(nm / world_size, 1.0)
    losses, all_idx = {}, []
    layers = [("", outputs)] + [(f"_{i}", a) for i, a in enumerate(outputs.get("aux_outputs", []))]
    for li, (suffix, out) in enumerate(layers):
        hc = [] if costs is not None else None
        idx = hungarian_match(out, targets, rand, w_class=w_class, w_mask=w_mask, w_dice=w_dice,
                              num_points=match_points or num_points, costs=hc)
        all_idx.append(idx)
        if costs is not None:
            costs.append(hc)
        if indices_override is not None:
            idx = indices_override[li]
        losses["loss_ce" + suffix] = loss_labels(out["pred_logits"], targets, idx, num_classes, empty_weight)
        bce, dice = loss_masks(out["pred_masks"], targets, idx, num_masks, rand, num_points, oversample, importance,
                               points_override=None if points_override is None else points_override[li], points_out=points_out)
        losses["loss_mask" + suffix], losses["loss_dice" + suffix] = bce, dice
    return (losses, all_idx) if return_indices else losses


def weight_dict(dec_layers=10, w_class=2.0, w_mask=5.0, w_dice=5.0):
    """proposal_model.py:127-134."""
    wd = {"loss_ce": w_class, "loss_mask": w_mask, "loss_dice": w_dice}
    for i in range(dec_layers - 1):
        wd.update({f"loss_ce_{i}": w_class, f"loss_mask_{i}": w_mask, f"loss_dice_{i}": w_dice})
    return wd


# ----------------------------------------------------------------------------- backbones
def frozen_bn(sd, p, x, eps=1e-5):
    scale = sd[p + ".weight"] * (sd[p + ".running_var"] + eps).rsqrt()
    bias = sd[p + ".bias"] - sd[p + ".running_mean"] * scale
    return x * scale.view(1, -1, 1, 1).to(x.dtype) + bias.view(1, -1, 1, 1).to(x.dtype)


def resnet50_forward(sd, p, x, depths=(3, 4, 6, 3)):
    """detectron2 0.6 build_resnet_backbone, cfg of
    configs/mask2former/coco/instance-segmentation/Base-COCO-InstanceSegmentation.yaml:2-15
    (STEM basic, STRIDE_IN_1X1 False, FrozenBN, res2..res5).  SURVEY Appendix D."""
    def cbn(name, t, stride=1, padding=0):
        return frozen_bn(sd, name + ".norm", F.conv2d(t, sd[name + ".weight"], None, stride, padding))
    x = F.relu(cbn(p + ".stem.conv1", x, 2, 3))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = {}
    for si, nblocks in enumerate(depths):
        stage = f"res{si + 2}"
        for bi in range(nblocks):
            bp = f"{p}.{stage}.{bi}"
            stride = 2 if (bi == 0 and si > 0) else 1
            sc = cbn(bp + ".shortcut", x, stride) if (bp + ".shortcut.weight") in sd else x
            y = F.relu(cbn(bp + ".conv1", x, 1))
            y = F.relu(cbn(bp + ".conv2", y, stride, 1))
            y = cbn(bp + ".conv3", y)
            x = F.relu(y + sc)
        outs[stage] = x
    return outs


# ----------------------------------------------------------------------------- meta-arch
def image_list(images, divisibility, pad_value=0.0):
    """detectron2 ImageList.from_tensors."""
    h = max(i.shape[-2] for i in images)
    w = max(i.shape[-1] for i in images)
    if divisibility > 1:
        h = (h + divisibility - 1) // divisibility * divisibility
        w = (w + divisibility - 1) // divisibility * divisibility
    out = images[0].new_full((len(images), images[0].shape[0], h, w), pad_value)
    for k, im in enumerate(images):
        out[k, :, :im.shape[-2], :im.shape[-1]] = im
    return out


def prepare_pseudo_targets(batched_inputs, h_pad, w_pad, with_object_class=False):
    """proposal_model.py:313-338 / part_distillation_model.py:405-428."""
    tg = []
    for x in batched_inputs:
        m = x["instances"]["gt_masks"]
        pad = torch.zeros((m.shape[0], h_pad, w_pad), dtype=m.dtype)
        pad[:, :m.shape[1], :m.shape[2]] = m
        t = {"masks": pad, "object_masks": pad.sum(0, keepdim=True)}
        if with_object_class:
            t["labels"] = x["instances"]["gt_classes"].long()
            t["gt_object_class"] = int(x["gt_object_class"])
        else:
            t["labels"] = torch.zeros(m.shape[0]).long()
        tg.append(t)
    return tg


PIXEL_MEAN = (123.675, 116.280, 103.530)
PIXEL_STD = (58.395, 57.120, 57.375)


def proposal_model_losses(sd, batched_inputs, rand, *, backbone="r50", num_classes=1, dec_layers=10, nheads=8,
                          enc_layers=6, num_points=12544, oversample=3.0, importance=0.75, size_div=32,
                          world_size=1, part=None, backbone_fn=None, match_points=None, return_indices=False,
                          indices_override=None, costs=None, points_override=None, points_out=None):
    """ProposalModel.forward train branch, proposal_model.py:177-204 (and
    PartDistillationModel.forward :197-226 when ``part=num_part_classes``):
    normalise -> pad -> backbone -> head -> criterion -> weight."""
    mean = torch.tensor(PIXEL_MEAN).view(-1, 1, 1)
    std = torch.tensor(PIXEL_STD).view(-1, 1, 1)
    imgs = image_list([(x["image"].float() - mean) / std for x in batched_inputs], size_div)
    if backbone_fn is not None:
        feats = backbone_fn(sd, "backbone", imgs)
    elif backbone == "r50":
        feats = resnet50_forward(sd, "backbone", imgs)
    else:
        raise ValueError(backbone)
    targets = prepare_pseudo_targets(batched_inputs, imgs.shape[-2], imgs.shape[-1], with_object_class=part is not None)
    mf, _, ms = pixel_decoder_forward(sd, "sem_seg_head.pixel_decoder", feats, nheads=nheads, enc_layers=enc_layers)
    out = decoder_forward(sd, "sem_seg_head.predictor", ms, mf, nheads=nheads, dec_layers=dec_layers - 1,
                          part=(targets, part) if part is not None else None)
    losses, idx = set_criterion(out, targets, rand, num_classes=num_classes if part is None else part,
                                num_points=num_points, oversample=oversample, importance=importance,
                                world_size=world_size, match_points=match_points, return_indices=True,
                                indices_override=indices_override, costs=costs, points_override=points_override, points_out=points_out)
    wd = weight_dict(dec_layers)
    losses = {k: v * wd[k] for k, v in losses.items() if k in wd}
    return (losses, idx) if return_indices else losses


# ----------------------------------------------------------------------------- optimizer
def clipped_adamw_step(params, grads, state, *, lrs, wds, clip=0.01, betas=(0.9, 0.999), eps=1e-8, step):
    """FullModelGradientClippingOptimizer(AdamW).step, base_trainer.py:118-133:
    global-norm clip (norm type 2, torch.nn.utils.clip_grad_norm_: coef =
    clip/(total+1e-6) clamped to 1) then torch.optim.AdamW (decoupled WD).  A gradient of None = a parameter
    whose .grad is None: skipped by clip_grad_norm_ and by AdamW alike (weights and moments untouched)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads if g is not None)).float()
    coef = torch.clamp(clip / (total + 1e-6), max=1.0)
    b1, b2 = betas
    for i, (p, g) in enumerate(zip(params, grads)):
        if g is None:
            continue
        g = g * coef
        m, v = state[i]
        p.mul_(1 - lrs[i] * wds[i])
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-lrs[i] / bc1)
    return total
