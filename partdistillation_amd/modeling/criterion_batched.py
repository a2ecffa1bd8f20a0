"""All prediction heads of the set criterion in ONE pass.

The reference (criterion.py:235-270, matcher.py:100-168) loops over the final + 9 auxiliary outputs and, inside the
matcher, over the images: ~150 tiny launches and one host round trip per (head, image).  Nothing in that loop is
sequentially dependent — every head's matching depends only on that head's outputs — so here the H heads x B images
are stacked and processed together with the same arithmetic:

  * point sampling uses the CHANNEL dimension for the batch of masks that share coordinates: the Q query masks of a
    (head, image) are Q channels of one grid_sample input, the n target masks of an image are n channels, so no mask
    is repeated or re-cast per call (the reference repeats the coordinates per mask, matcher.py:130-139);
  * the H*B assignment problems are one launch of pd_lsa_batched;
  * matched pairs are addressed through index tensors built from host-known sizes (n_b per image), so no
    boolean-mask indexing / .nonzero() synchronises the host;
  * importance sampling runs one top-k over all matched masks of all heads.

Random draws follow the reference's order (per head: B matcher draws, then the oversampled and the random point sets
of loss_masks) when a replay hook is installed, so parity tests can replay them; otherwise three device-side draws.
Loss values are identical to the per-head loop up to fp32 summation order."""
import torch
import torch.nn.functional as F

from ..functions import criterion_ops as cops
from ..functions import lsa as lsa_op
from ..functions import rowwise as rw
from ..functions import smallgemm as _sg


class LossDict(dict):
    """dict of per-head losses that also carries their stacked vectors (one autograd node per loss type)."""
    vectors = None
    total = None
    indices = None
    points = None
    stacked = None


class _TokensTimesRows(torch.autograd.Function):
    """tok [hw, C] @ rows[n, C]^T -> [hw, n] (fp32): the mask logits of the n matched queries of an image.  The gradient of `rows`
    contracts over the hw = 65 536 tokens into an [n, C] result — the shape the library serves with a 260 us kernel (2 x per step);
    here it is the transpose-read split GEMM of the encoder's weight gradients (functions/gemm.py: 30 us)."""

    @staticmethod
    def forward(ctx, tok, rows):
        ctx.save_for_backward(tok, rows)
        return tok @ rows.t()

    @staticmethod
    def backward(ctx, g):
        from ..functions.gemm import gemm_wgrad
        tok, rows = ctx.saved_tensors
        g = g.contiguous()
        d_tok = g @ rows if ctx.needs_input_grad[0] else None
        d_rows = gemm_wgrad(g, tok) if ctx.needs_input_grad[1] else None        # g^T tok
        return d_tok, d_rows


class _TokensTimesRowsBatched(torch.autograd.Function):
    """the same for every image of the batch in one autograd node: tok [B, hw, C], rows_b [n_b, C] -> ([hw, n_b])_b.  The gradient of
    `tok` is written per image straight into one [B, hw, C] tensor (separate nodes made autograd zero-fill that tensor and add each
    image's slice into it: 36 + 63 + 49 us of fills / adds / copies per step)."""

    @staticmethod
    def forward(ctx, tok, *rows):
        ctx.save_for_backward(tok, *rows)
        return tuple(tok[b] @ r.t() for b, r in enumerate(rows))

    @staticmethod
    def backward(ctx, *gs):
        from ..functions.gemm import gemm_wgrad
        tok, rows = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        d_tok = torch.empty_like(tok) if ctx.needs_input_grad[0] else None
        d_rows = []
        for b, (g, r) in enumerate(zip(gs, rows)):
            if g is None or r.shape[0] == 0:
                if d_tok is not None:
                    d_tok[b].zero_()
                d_rows.append(None if g is None else torch.zeros_like(r))
                continue
            g = g.contiguous()
            if d_tok is not None:
                torch.mm(g, r, out=d_tok[b])
            if r.shape[0] % 4 == 0 and tok.shape[2] % 4 == 0:
                d_rows.append(gemm_wgrad(g, tok[b]))                             # g^T tok: contraction over the hw tokens
            else:
                d_rows.append(g.t() @ tok[b])
        return (d_tok, *d_rows)


def _tokens_times_rows_batched(tok, rows):
    if tok.is_cuda and tok.dtype == torch.float32 and all(r.dtype == torch.float32 for r in rows) and tok.is_contiguous() \
            and torch.is_grad_enabled():
        return _TokensTimesRowsBatched.apply(tok, *rows)
    return [tok[b] @ r.t() for b, r in enumerate(rows)]


def _tokens_times_rows(tok, rows):
    if tok.is_cuda and tok.dtype == torch.float32 and rows.dtype == torch.float32 and rows.shape[0] % 4 == 0 and tok.shape[1] % 4 == 0 \
            and tok.stride(1) == 1 and torch.is_grad_enabled():
        return _TokensTimesRows.apply(tok, rows)
    return tok @ rows.t()


_SEL_CACHE = {}


def _pair_selectors(H, B, npair, device):
    key = (H, B, tuple(npair), str(device))
    if key not in _SEL_CACHE:
        if len(_SEL_CACHE) >= _CACHE_CAP:                                  # target counts vary per batch on real data: bounded
            _SEL_CACHE.clear(), _DERIVED.clear()
        sel_h, sel_b, sel_k = [], [], []
        for h in range(H):
            for b in range(B):
                for k in range(npair[b]):
                    sel_h.append(h), sel_b.append(b), sel_k.append(k)
        mk = lambda x: torch.tensor(x, dtype=torch.long, device=device)
        per_image_l = [[i for i, bb in enumerate(sel_b) if bb == b] for b in range(B)]
        per_image = [mk(l) for l in per_image_l]
        flat = [i for l in per_image_l for i in l]                         # pairs in image-major order
        inv = mk(sorted(range(len(flat)), key=flat.__getitem__))           # image-major position of pair i
        _SEL_CACHE[key] = (mk(sel_h), mk(sel_b), mk(sel_k), per_image, inv, mk(flat), [len(l) for l in per_image_l])
    return _SEL_CACHE[key]


_CONST_CACHE = {}
_DERIVED = {}
_CACHE_CAP = 512


def _const(values, dtype, device):
    key = (tuple(values), dtype, str(device))
    if key not in _CONST_CACHE:
        _CONST_CACHE[key] = torch.tensor(list(values), dtype=dtype, device=device)
    return _CONST_CACHE[key]


def _gs(inp, coords):
    """grid_sample of [N,C,H,W] at coords [N,P,2] in [0,1] (x,y) -> [N,C,P]; bilinear, zeros, align_corners=False."""
    if rw.point_sample_planar_supported(inp, coords):
        return rw.point_sample_planar(inp, coords)          # HIP kernel: 10 us where the generic grid_sampler takes 50 (200 backward)
    return F.grid_sample(inp, 2.0 * coords.unsqueeze(2) - 1.0, mode="bilinear", padding_mode="zeros", align_corners=False).squeeze(3)


def batched_set_criterion(crit, outputs, targets, padded_masks):
    """-> LossDict with the reference's 3*H keys.  `padded_masks` bool [B, n_max, Hm, Wm].

    Index conventions: h = criterion order (0 = final output, 1.. = aux_outputs[h-1]; the order of the random draws);
    d = decoder order (d = aux index, last = final); problems are laid out p = b*H + d so that the stacked mask tensor
    [B, D, Q, h, w] the decoder produces is consumed without any re-ordering copy."""
    m = crit.matcher
    dev = outputs["pred_logits"].device
    aux = outputs["aux_outputs"]
    H, (B, Q, K1) = len(aux) + 1, outputs["pred_logits"].shape
    # sparse: the decoder did not form the [B, D*Q, h, w] masks (mask = mask_embed . mask_features).  Bilinear point
    # sampling is linear in the map, so the matcher's samples of ALL Q masks at its shared points are
    # mask_embed . point_sample(mask_features), and only the N matched masks are evaluated densely for the loss.
    sparse = outputs.get("pred_masks") is None and outputs.get("mask_embeds") is not None
    masks_bd = None
    if sparse:
        emb_bd, mfeat = outputs["mask_embeds"], outputs["mask_features"]                         # [B,D,Q,C], [B,C,h,w]
        assert emb_bd.shape[1] == H
    elif outputs.get("all_masks") is not None and outputs["all_masks"].shape[1] == H:
        masks_bd = outputs["all_masks"]                                                          # [B,D,Q,h,w]
    else:
        masks_bd = torch.stack([a["pred_masks"] for a in aux] + [outputs["pred_masks"]], dim=1)
    if "all_logits" in outputs and outputs["all_logits"].shape[0] == H:
        logits_bd = outputs["all_logits"].transpose(0, 1)                                        # [B,D,Q,K1]
    else:
        logits_bd = torch.stack([a["pred_logits"] for a in aux] + [outputs["pred_logits"]], dim=1)
    dec_of = [H - 1] + list(range(H - 1))                                                        # h -> d
    h_of = [dec_of.index(d) for d in range(H)]                                                   # d -> h
    d_of_h, h_of_d = _const(dec_of, torch.long, dev), _const(h_of, torch.long, dev)
    ns = [int(t["labels"].shape[0]) for t in targets]
    nmax = max(ns)
    npair = [min(Q, n) for n in ns]
    N_h = sum(npair)
    Pm, P = m.num_points, crit.num_points
    kover, kimp = int(P * crit.oversample_ratio), int(crit.importance_sample_ratio * P)
    krand = P - kimp

    # ---- random draws (reference order when replayed): mcoords [H,B,Pm,2]; o/r coords [H*N_h, *, 2] (h-major)
    if crit.rand is not None:
        mc, oc, rc = [], [], []
        for _ in range(H):
            mc.append(torch.stack([crit.rand((1, Pm, 2))[0] for _ in range(B)]))
            oc.append(crit.rand((N_h, kover, 2)))
            if krand > 0:
                rc.append(crit.rand((N_h, krand, 2)))
        mcoords = torch.stack(mc).to(dev)
        ocoords = torch.cat(oc).to(dev)
        rcoords = torch.cat(rc).to(dev) if krand > 0 else None
    else:
        mcoords = torch.rand((H, B, Pm, 2), device=dev)
        ocoords = torch.rand((H * N_h, kover, 2), device=dev)
        rcoords = torch.rand((H * N_h, krand, 2), device=dev) if krand > 0 else None

    # the target masks are sampled as the bytes they are stored in (pd_point_sample_u8) on the GPU; the fp32 copy is the torch path's
    byte_masks = cops.point_sample_masks_supported(padded_masks, ocoords) and nmax > 0
    tmask = None if byte_masks else padded_masks.float()                                         # [B,nmax,Hm,Wm], once
    labels_pad = torch.zeros((B, nmax), dtype=torch.long, device=dev)
    for b, t in enumerate(targets):
        labels_pad[b, : ns[b]] = t["labels"]
    ncols = _const([ns[b] for b in range(B) for _ in range(H)], torch.int32, dev)                # per problem p = b*H+d

    with torch.no_grad(), torch.autocast(device_type=dev.type, enabled=False):
        # ---- matcher costs for all (image, head) problems (matcher.py:108-158)
        mc_bd = mcoords[h_of_d].transpose(0, 1)                                                  # [B,D,Pm,2]
        if sparse:
            e = emb_bd.detach().reshape(B * H, Q, -1)
            mfd, mcf = mfeat.detach(), mc_bd.reshape(B, H * Pm, 2)
        if sparse and cops.match_point_logits_supported(mfd, mcf, e):
            pm = cops.match_point_logits(mfd, mcf, e)                    # sampler + product in one kernel: the sampled features stay in LDS
        elif sparse:
            fm = rw.point_sample_nhwc(mfeat.detach(), mc_bd.reshape(B, H * Pm, 2),                # [B, D*Pm, C], already in e's dtype
                                      out_dtype=e.dtype if e.dtype == torch.bfloat16 else torch.float32)
            # bf16 mask embeddings (autocast): the reference's mask logits are a bf16 product too (einsum under AMP, :449)
            fmv = fm.view(B * H, Pm, -1).to(e.dtype)
            if _sg.bmm_tn_supported(e, fmv):
                pm = _sg.bmm_tn(e, fmv)                                  # all (image, head) problems: one launch of the skinny bf16 kernel
            else:
                pm = torch.bmm(e, fmv.transpose(1, 2))
        else:
            pm = _gs(masks_bd.detach().reshape(B * H, Q, *masks_bd.shape[-2:]).float(), mc_bd.reshape(B * H, Pm, 2))   # [BD,Q,Pm]
        if byte_masks:                                                                           # row (b, j): map b * nmax + j at image b's points
            tg4 = cops.point_sample_masks(padded_masks, mc_bd.reshape(B, H * Pm, 2), None, nmax).view(B, nmax, H, Pm)
        else:
            tg4 = _gs(tmask, mc_bd.reshape(B, H * Pm, 2)).reshape(B, nmax, H, Pm)                # the sampler's layout: targets x heads
        lf = logits_bd.detach().float()
        prob = lf.sigmoid() if K1 == 1 else lf.softmax(-1)
        if cops.matcher_costs_supported(pm, tg4, prob, labels_pad) and nmax > 0:
            # softplus / sigmoid sums, the two products with the targets, the class term and the weighted sum: one pass over the point
            # logits (pd_matcher_costs); neither fp32 copy of the logits is formed
            C = cops.matcher_costs(pm, tg4, prob.reshape(B * H, Q, K1), labels_pad, H, m.cost_mask, m.cost_class, m.cost_dice)
        else:
            tg = tg4.transpose(1, 2).reshape(B * H, nmax, Pm)
            tgt = tg.transpose(1, 2)
            pm = pm.float()
            sg = pm.sigmoid()
            sp_sum, sg_sum = F.softplus(pm).sum(-1), sg.sum(-1)
            cost_mask = (sp_sum[:, :, None] - torch.bmm(pm, tgt)) / Pm
            cost_dice = 1 - (2 * torch.bmm(sg, tgt) + 1) / (sg_sum[:, :, None] + tg.sum(-1)[:, None, :] + 1)
            cost_class = -torch.gather(prob, 3, labels_pad[:, None, None, :].expand(B, H, Q, nmax)).reshape(B * H, Q, nmax)
            C = m.cost_mask * cost_mask + m.cost_class * cost_class + m.cost_dice * cost_dice
        rows, cols = lsa_op.solve_batched(C, ncols)                                              # [BD,nmax]
        sel_h, sel_b, sel_k, per_image, inv_img, img_major, img_counts = _pair_selectors(H, B, npair, dev)   # pairs, h-major
        # everything about the pair list that does not depend on the matcher's output is computed once per (heads, batch, target counts)
        # (was ~10 launches of index arithmetic on 80-element tensors per step)
        dk = (H, B, tuple(npair), Q, nmax, str(dev))
        der = _DERIVED.get(dk)
        if der is None:
            if len(_DERIVED) >= _CACHE_CAP:
                _DERIVED.clear()
            sd = d_of_h[sel_h]
            sp = sel_b * H + sd
            der = _DERIVED[dk] = (sd, sp, sp * rows.shape[1] + sel_k, (sd * B + sel_b) * Q, (sel_b * H + sd) * Q, sel_b * nmax)
        sel_d, sel_p, lin_pk, base_db, base_bd, base_tgt = der
        q_idx, j_idx = rows.reshape(-1).index_select(0, lin_pk), cols.reshape(-1).index_select(0, lin_pk)
        # ---- classification targets (criterion.py:126-145)
        tclass = torch.full((B, H, Q), crit.num_classes, dtype=torch.long, device=dev)
        tclass.view(-1).index_copy_(0, base_bd + q_idx, labels_pad.view(-1).index_select(0, base_tgt + j_idx))     # [b, d, q] <- labels[b, j]
    num_masks = crit.num_masks(targets, dev)

    with torch.autocast(device_type=dev.type, enabled=False):
        w = crit.empty_weight
        logits_f = logits_bd.float()
        fused_vectors = None                             # set below when pd_loss_vectors forms all three vectors (it needs bce / dice)
        lv_ok = cops.loss_vectors_supported(logits_f, tclass, w) and N_h > 0
        if not lv_ok:
            nll = F.cross_entropy(logits_f.reshape(B * H, Q, K1).transpose(1, 2), tclass.reshape(B * H, Q), w, reduction="none")
            ce_d = nll.reshape(B, H, Q).sum((0, 2)) / w[tclass].sum((0, 2))
            loss_ce = ce_d.index_select(0, d_of_h)       # (index_select: its backward is one index_add; x[idx] sorts the indices: 6 launches)
        # ---- mask losses on the matched pairs (criterion.py:147-207)
        if sparse:
            # the matched queries' embeddings, gathered ONCE and already in image-major order: rows of the flat [B heads Q, C] matrix
            emb_db = emb_bd.transpose(0, 1)                                                      # the decoder's own [heads, B, Q, C] layout
            if emb_db.is_contiguous():
                emb_rows, flat_idx = emb_db.reshape(H * B * Q, -1), base_db + q_idx      # a view: no copy of the embeddings
            else:
                emb_rows, flat_idx = emb_bd.reshape(B * H * Q, -1), base_bd + q_idx
            e_img = emb_rows.index_select(0, flat_idx.index_select(0, img_major)).float()        # [N,C]
            # [hw, C] @ [C, n_b]: with channels-last mask features the operand is read in place and its gradient comes
            # back channels-last, the layout the 1x1 mask_features convolution's backward wants
            mf_tok = mfeat.float().permute(0, 2, 3, 1).reshape(B, -1, mfeat.shape[1])            # [B, hw, C]
            if cops.pair_logits_supported(mf_tok, e_img):
                # one launch: the N matched pairs' logits written in pair order (pd_pair_logits_fwd; fp32 matrix cores)
                src = cops.pair_logits(mf_tok, e_img, img_major, img_counts).view(-1, 1, *mfeat.shape[-2:])
            else:
                parts = [p.t() for p in _tokens_times_rows_batched(mf_tok, list(e_img.split(img_counts))) if p.numel()]
                src = torch.cat(parts).index_select(0, inv_img).view(-1, 1, *mfeat.shape[-2:])   # [N,1,h,w], pair (h-major) order
        else:
            src = masks_bd[sel_b, sel_d, q_idx][:, None].float()                                 # [N,1,h,w]
        with torch.no_grad():
            samp = _gs(src, ocoords).squeeze(1)                               # [N, kover]
            if cops.uncertain_points_supported(samp, ocoords, kimp):
                # the kimp oversampled points with the smallest |logit|, then the random ones: radix select + compaction, one launch
                # (pd_uncertain_points; was abs, neg, a ~20-launch top-k, gather, cat).  The losses are sums over the chosen points.
                coords = cops.uncertain_points(samp, ocoords, kimp, rcoords if krand > 0 else None)   # [N,P,2]
            else:
                idx = torch.topk(-samp.abs(), k=kimp, dim=1, sorted=False)[1]
                coords = torch.gather(ocoords, 1, idx[:, :, None].expand(-1, -1, 2))
                if krand > 0:
                    coords = torch.cat([coords, rcoords], dim=1)                                 # [N,P,2]
            if byte_masks:
                # pair n reads ITS target (image sel_b[n], target j_idx[n]) at ITS points: one launch, no n_targets-fold oversampling
                labels = cops.point_sample_masks(padded_masks, coords, base_tgt + j_idx)     # [N,P]
            else:
                labels = torch.empty((coords.shape[0], P), dtype=torch.float32, device=dev)
                for b in range(B):                                                               # targets as channels
                    pi = per_image[b]
                    if pi.numel() == 0:
                        continue
                    s = _gs(tmask[b:b + 1], coords[pi].reshape(1, -1, 2)).reshape(nmax, pi.numel(), P)
                    labels[pi] = s[j_idx[pi], _arange(pi.numel(), dev)]
        pl = _gs(src, coords).squeeze(1)                                                         # [N,P] (a view both ways: select's backward fills and copies)
        if cops.mask_point_losses_supported(pl, labels):
            bce, dice = cops.mask_point_losses(pl, labels)                    # per mask, forward and backward one launch each
        else:
            bce = F.binary_cross_entropy_with_logits(pl, labels, reduction="none").mean(1)
            ps = pl.sigmoid()
            dice = 1 - (2 * (ps * labels).sum(-1) + 1) / (ps.sum(-1) + labels.sum(-1) + 1)
        if lv_ok:
            # class-weighted cross entropy of every head and the per-head sums of the mask terms: one launch, one for all their gradients
            fused_vectors = cops.loss_vectors(logits_f, tclass, w, d_of_h, bce, dice, num_masks)        # [3, H]
            loss_ce, loss_mask, loss_dice = fused_vectors.unbind(0)
        else:
            loss_mask = bce.reshape(H, N_h).sum(1) / num_masks
            loss_dice = dice.reshape(H, N_h).sum(1) / num_masks

    out = LossDict()
    out.vectors = {"loss_ce": loss_ce, "loss_mask": loss_mask, "loss_dice": loss_dice}
    out.stacked = fused_vectors                            # [3, H] in the order of `vectors` when one node produced them
    for name, vec in out.vectors.items():
        parts = vec.unbind(0)
        out[name] = parts[0]
        for i in range(H - 1):
            out[f"{name}_{i}"] = parts[i + 1]
    out.indices = (rows, cols)
    out.points = coords.detach()          # [H * N_h, P, 2], h-major pair order: the loss points this step chose (parity tests replay them in the oracle)
    return out


_ARANGE = {}


def _arange(n, device):
    key = (n, str(device))
    if key not in _ARANGE:
        _ARANGE[key] = torch.arange(n, device=device)
    return _ARANGE[key]
