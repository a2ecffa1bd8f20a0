"""GPU parity of the set criterion's kernels (include/pd_criterion.h, csrc/criterion.hip) against the plain torch expressions of the
reference's matcher / criterion (matcher.py:13-62, 108-158; criterion.py:25-88, 181-189), evaluated in fp64 where a sum is involved.  The
full-size step tests (tests/test_product_gpu.py) cover the same kernels inside the step against the CPU oracle."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,H,Q,n,nt,K1", [(2, 10, 100, 12544, 4, 2), (1, 3, 5, 257, 1, 1), (2, 2, 7, 1000, 9, 5), (3, 1, 4, 64, 17, 3),
                                            (2, 4, 100, 112 * 112, 2, 2)])
def test_matcher_costs_match_the_torch_expressions(dtype, B, H, Q, n, nt, K1):
    """cost = w_mask * sigmoid-CE + w_class * (-prob[label]) + w_dice * dice over all (image, head) problems, the target samples read in
    the sampler's [B, targets, heads * points] layout; logits over the range where softplus switches branches.  Tolerance: 2e-5 relative
    to the cost's magnitude (fp32 sums over up to 12 544 points in a different order than the library GEMM)."""
    from partdistillation_amd.functions import criterion_ops as cops
    torch.manual_seed(B * 1000 + n + nt)
    x = (torch.randn(B * H, Q, n, device=DEV) * 8).to(dtype)
    x.view(-1)[:4] = torch.tensor([0.0, 20.0, 20.5, -45.0], device=DEV).to(dtype)
    t_flat = (torch.rand(B, nt, H * n, device=DEV) > 0.6).float() * torch.rand(B, nt, H * n, device=DEV)     # bilinear samples of 0/1 masks
    t_flat[:, -1] = 0                                                                                          # a padded (all-zero) target
    t4 = t_flat.view(B, nt, H, n)
    logits = torch.randn(B, H, Q, K1, device=DEV)
    prob = logits.sigmoid() if K1 == 1 else logits.softmax(-1)
    labels = torch.randint(0, K1, (B, nt), device=DEV)
    wm, wc, wd = 5.0, 2.0, 5.0
    assert cops.matcher_costs_supported(x, t4, prob, labels)
    got = cops.matcher_costs(x, t4, prob.reshape(B * H, Q, K1), labels, H, wm, wc, wd)
    xd = x.double()
    tg = t4.transpose(1, 2).reshape(B * H, nt, n).double()
    sp, sg = F.softplus(xd).sum(-1), xd.sigmoid()
    cost_mask = (sp[:, :, None] - torch.bmm(xd, tg.transpose(1, 2))) / n
    cost_dice = 1 - (2 * torch.bmm(sg, tg.transpose(1, 2)) + 1) / (sg.sum(-1)[:, :, None] + tg.sum(-1)[:, None, :] + 1)
    cost_class = -torch.gather(prob.double(), 3, labels[:, None, None, :].expand(B, H, Q, nt)).reshape(B * H, Q, nt)
    ref = wm * cost_mask + wc * cost_class + wd * cost_dice
    assert got.shape == ref.shape and torch.isfinite(got).all()
    err = float((got.double() - ref).abs().max() / ref.abs().max())
    print(f"matcher costs {dtype} B={B} H={H} Q={Q} n={n} targets={nt}: max |err| / max |cost| = {err:.2e}")
    assert err <= 2e-5


@pytest.mark.parametrize("B,heads,Q,Pm,H,W", [(2, 10, 100, 12544, 256, 256), (1, 1, 1, 4, 5, 7), (2, 3, 100, 132, 64, 48), (3, 2, 37, 1000, 33, 40),
                                               (1, 4, 128, 64, 16, 16)])
def test_match_point_logits_equal_the_sampler_followed_by_the_batched_product(B, heads, Q, Pm, H, W):
    """pd_match_point_logits (sampler + product in one kernel, the sampled features in LDS only) against the two launches it replaces —
    pd_point_sample_nhwc_f32_bf16, then pd_sgemm_tn_batched_bf16 —: BIT-identical (same fp32 bilinear sum, same roundings, same MFMA and k order),
    and against torch fp64 on the same operands (grid_sample of the fp32 map, bf16 product) within bf16 rounding.  Points include the map's
    border and positions outside [0, 1] (zero padding), the last point tile is ragged, Q is not a multiple of 32."""
    from partdistillation_amd.functions import criterion_ops as cops, rowwise as rw, smallgemm as sg
    torch.manual_seed(B * 100 + Pm)
    mf = torch.randn(B, 256, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    co = torch.rand(B, heads * Pm, 2, device=DEV) * 1.1 - 0.05
    co.view(-1)[:6] = torch.tensor([0.0, 0.0, 1.0, 1.0, 0.5, -0.2], device=DEV)[: min(6, co.numel())]
    e = (torch.randn(B * heads, Q, 256, device=DEV) * 0.3).bfloat16()
    assert cops.match_point_logits_supported(mf, co, e)
    got = cops.match_point_logits(mf, co, e)
    fm = rw.point_sample_nhwc(mf, co, out_dtype=torch.bfloat16).view(B * heads, Pm, 256)
    assert got.shape == (B * heads, Q, Pm) and got.dtype == torch.bfloat16
    if sg.bmm_tn_supported(e, fm) and Pm % 8 == 0:                   # (the batched product wants rows of 16 bytes)
        two = sg.bmm_tn(e, fm)
        assert torch.equal(got, two), float((got.float() - two.float()).abs().max())
    ref = F.grid_sample(mf.double(), (2 * co.double() - 1).view(B, heads * Pm, 1, 2), mode="bilinear", padding_mode="zeros", align_corners=False)
    ref = ref[..., 0].transpose(1, 2).reshape(B * heads, Pm, 256).float().bfloat16().double()              # [B heads, Pm, C], rounded as the operand is
    ref = torch.bmm(e.double(), ref.transpose(1, 2))
    err = float((got.double() - ref).abs().max() / ref.abs().max())
    print(f"match point logits B={B} heads={heads} Q={Q} points={Pm} map {H}x{W}: max |err| / max |logit| = {err:.2e}")
    assert err <= 1.2e-2            # one bf16 rounding of the result (2^-8) + operand roundings that differ by an ulp of the fp32 bilinear sum


@pytest.mark.parametrize("N,P", [(80, 12544), (1, 1), (7, 300), (3, 4099)])
def test_mask_point_losses_match_torch(N, P):
    """per-mask BCE-with-logits mean and dice at the sampled points, and their gradient (criterion.py:25-69), against fp64 autograd"""
    from partdistillation_amd.functions import criterion_ops as cops
    torch.manual_seed(N + P)
    x = (torch.randn(N, P, device=DEV) * 6).requires_grad_(True)
    y = ((torch.rand(N, P, device=DEV) > 0.5).float() * torch.rand(N, P, device=DEV)).contiguous()
    assert cops.mask_point_losses_supported(x, y)
    bce, dice = cops.mask_point_losses(x, y)
    gb, gd = torch.randn(N, device=DEV), torch.randn(N, device=DEV)
    (gx,) = torch.autograd.grad((bce * gb).sum() + (dice * gd).sum(), x)
    xd = x.detach().double().requires_grad_(True)
    yd = y.double()
    rb = F.binary_cross_entropy_with_logits(xd, yd, reduction="none").mean(1)
    ps = xd.sigmoid()
    rd = 1 - (2 * (ps * yd).sum(-1) + 1) / (ps.sum(-1) + yd.sum(-1) + 1)
    (rx,) = torch.autograd.grad((rb * gb.double()).sum() + (rd * gd.double()).sum(), xd)
    torch.testing.assert_close(bce.double(), rb.detach(), rtol=3e-6, atol=1e-7)
    torch.testing.assert_close(dice.double(), rd.detach(), rtol=3e-6, atol=3e-7)
    assert float((gx.double() - rx).abs().max()) <= 3e-6 * float(rx.abs().max()) + 1e-12
    # one of the two upstream gradients absent
    bce2, dice2 = cops.mask_point_losses(x, y)
    (g1,) = torch.autograd.grad((bce2 * gb).sum(), x)
    (r1,) = torch.autograd.grad((F.binary_cross_entropy_with_logits(xd, yd, reduction="none").mean(1) * gb.double()).sum(), xd)
    assert float((g1.double() - r1).abs().max()) <= 3e-6 * float(r1.abs().max()) + 1e-12


@pytest.mark.parametrize("N,K,k,R", [(80, 37632, 9408, 3136), (3, 40960, 40960, 0), (5, 1000, 1, 7), (2, 1025, 512, 0), (4, 64, 63, 2)])
def test_uncertain_points_select_what_topk_selects(N, K, k, R):
    """the k oversampled points with the smallest |logit| (= top-k of -|logit|, criterion.py:72-88, 181-189) followed by the random ones;
    the chosen SET equals torch.topk's (no ties in continuous draws)"""
    from partdistillation_amd.functions import criterion_ops as cops
    torch.manual_seed(K + k)
    v = torch.randn(N, K, device=DEV) * 5
    coords = torch.rand(N, K, 2, device=DEV)
    coords[:, :, 0] = torch.arange(K, device=DEV, dtype=torch.float32)        # x = the point's index (exact in fp32), y random
    rnd = torch.rand(N, R, 2, device=DEV) if R else None
    assert cops.uncertain_points_supported(v, coords, k)
    out = cops.uncertain_points(v, coords, k, rnd)
    assert out.shape == (N, k + R, 2)
    got = out[:, :k, 0].long()
    assert torch.equal(torch.gather(coords[:, :, 1], 1, got), out[:, :k, 1])  # each output pair is one input pair
    idx = torch.topk(-v.abs(), k=k, dim=1, sorted=False)[1].sort(1)[0]
    assert torch.equal(got.sort(1)[0], idx)
    if R:
        assert torch.equal(out[:, k:], rnd)


def test_uncertain_points_with_ties_at_the_threshold():
    """equal |logit| at the threshold (exact zeros, sign-symmetric values, a row of one repeated value): everything below the threshold is
    taken, the rest are points AT the threshold, no point twice — and the choice is the same on every call"""
    from partdistillation_amd.functions import criterion_ops as cops
    K, k = 3000, 1200
    v = torch.randint(-3, 4, (4, K), device=DEV).float()                  # seven distinct |values| only: the threshold bin is hundreds wide
    v[3] = 2.5
    coords = torch.arange(K, device=DEV, dtype=torch.float32)[None, :, None].expand(4, K, 2).contiguous()
    out = cops.uncertain_points(v, coords, k)
    assert torch.equal(out, cops.uncertain_points(v, coords, k))
    for r in range(4):
        a = v[r].abs()
        T = torch.sort(a)[0][k - 1]
        got = out[r, :, 0].long()
        assert got.unique().numel() == k
        less = torch.nonzero(a < T)[:, 0]
        assert torch.isin(less, got).all()                                 # every point below the threshold
        assert (a[got] <= T).all()                                         # and nothing above it


def test_criterion_ops_refuse_cpu_tensors():
    from partdistillation_amd.functions import criterion_ops as cops
    with pytest.raises(RuntimeError, match="GPU only"):
        cops.uncertain_points(torch.zeros(1, 4), torch.zeros(1, 4, 2), 2)
    with pytest.raises(RuntimeError, match="GPU only"):
        cops.MaskPointLosses.apply(torch.zeros(1, 4), torch.zeros(1, 4))
    with pytest.raises(RuntimeError, match="GPU only"):
        cops.matcher_costs(torch.zeros(1, 1, 4), torch.zeros(1, 1, 1, 4), torch.zeros(1, 1, 1), torch.zeros(1, 1, dtype=torch.long), 1, 1, 1, 1)


@pytest.mark.parametrize("rows_shape", [(10, 2, 100), (3, 64), (1, 200)])
def test_own_mlp_node_matches_the_module_path_under_autocast(rows_shape):
    """functions/mlp_own.py: the mask-embedding MLP as one node on pd_igemm_bf16 / pd_wgrad_bf16 against nn.Linear + ReLU under bf16 autocast
    (reference mask2former_transformer_decoder.py:198-204) — the same bf16 roundings between the layers, so outputs and gradients agree to
    bf16 resolution of their magnitudes (2^-7 relative to the tensor maximum; measured ~4e-3)."""
    from partdistillation_amd.functions import mlp_own
    from partdistillation_amd.modeling.transformer_decoder.mask2former_transformer_decoder import MLP
    torch.manual_seed(len(rows_shape))
    ref = MLP(256, 256, 256, 3).to(DEV)
    own = MLP(256, 256, 256, 3).to(DEV)
    own.load_state_dict(ref.state_dict())
    for p in own.parameters():
        p.data = p.data.to(torch.bfloat16)                       # the shadowed (bf16) parameters of the training step
    x = torch.randn(*rows_shape, 256, device=DEV)
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    go = torch.randn(*rows_shape, 256, device=DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert mlp_own.supported(x2, own.layers)
        y_ref = x1
        for i, l in enumerate(ref.layers):                       # the module path (MLP.forward would take the own node here too)
            y_ref = l(y_ref)
            y_ref = F.relu(y_ref) if i < 2 else y_ref
        y_own = own(x2)
    assert y_own.dtype == torch.bfloat16 and y_own.shape == y_ref.shape
    (y_ref.float() * go).sum().backward()
    (y_own.float() * go).sum().backward()

    def close(a, b, what):
        err = float((a.float() - b.float()).abs().max() / b.float().abs().max())
        assert err <= 2 ** -7, (what, err)
    close(y_own, y_ref, "y")
    close(x2.grad, x1.grad, "dx")
    for (n, p), q in zip(own.named_parameters(), ref.parameters()):
        assert p.grad.dtype == p.dtype
        close(p.grad, q.grad, n)


@pytest.mark.parametrize("rows_shape,K", [((10, 2, 100), 2), ((3, 70), 8), ((1, 203), 1)])
def test_own_heads_node_class_head_and_mlp_vs_fp32_module_path(rows_shape, K):
    """functions/mlp_own.py HeadsOwn: class head (pd_skinny_linear_*: bf16 input, fp32 products, fp32 logits) + the mask-embedding MLP as one
    node, against nn.Linear in fp32 on the SAME bf16-rounded input and weights (class head: 1e-5 of the magnitudes — only the summation order
    differs; the input gradient carries the MLP's bf16 gradient: 2^-7) — a transposed (non-contiguous) input as the decoder hands over."""
    from partdistillation_amd.functions import mlp_own
    from partdistillation_amd.modeling.transformer_decoder.mask2former_transformer_decoder import MLP
    import torch.nn as nn
    torch.manual_seed(K)
    mlp_ref, mlp_o = MLP(256, 256, 256, 3).to(DEV), MLP(256, 256, 256, 3).to(DEV)
    cls_ref, cls_o = nn.Linear(256, K).to(DEV), nn.Linear(256, K).to(DEV)
    mlp_o.load_state_dict(mlp_ref.state_dict()), cls_o.load_state_dict(cls_ref.state_dict())
    for p in list(mlp_o.parameters()) + list(cls_o.parameters()):
        p.data = p.data.to(torch.bfloat16)
    for p, q in zip(cls_ref.parameters(), cls_o.parameters()):
        p.data = q.data.float()                                       # the reference head holds the bf16-rounded values in fp32
    x = torch.randn(*rows_shape[::-1], 256, device=DEV).to(torch.bfloat16).float()
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    tr = (lambda t: t.transpose(0, 1)) if len(rows_shape) == 2 else (lambda t: t.permute(2, 1, 0, 3))
    gl, ge = torch.randn(*rows_shape, K, device=DEV), torch.randn(*rows_shape, 256, device=DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert mlp_own.heads_supported(tr(x2), cls_o, mlp_o.layers)
        logits, emb = mlp_own.heads(tr(x2), cls_o, mlp_o.layers)
        emb_ref = mlp_own.mlp(tr(x1), mlp_o.layers)                  # the MLP node alone (tested above against the module path)
    log_ref = cls_ref(tr(x1))
    assert logits.dtype == torch.float32 and logits.shape == (*rows_shape, K) and emb.dtype == torch.bfloat16
    ((logits * gl).sum() + (emb.float() * ge).sum()).backward()
    ((log_ref * gl).sum() + (emb_ref.float() * ge).sum()).backward()
    rel = lambda a, b: float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-30))
    assert rel(logits, log_ref) <= 1e-5
    assert float((emb.float() - emb_ref.float()).abs().max()) == 0.0
    assert rel(x2.grad, x1.grad) <= 2 ** -7
    for p, q in zip(cls_o.parameters(), cls_ref.parameters()):
        assert p.grad.dtype == torch.bfloat16
        assert rel(p.grad, q.grad) <= 2 ** -8                          # fp32 sums, rounded once to the bf16 parameter gradient
    for p, q in zip(mlp_o.parameters(), mlp_o.parameters()):
        assert p.grad is not None


@pytest.mark.parametrize("B,H,Q,K1,Nh", [(2, 10, 100, 2, 8), (3, 4, 37, 9, 1), (1, 1, 300, 1, 5), (2, 3, 5, 130, 300)])
def test_loss_vectors_equal_the_torch_expressions(B, H, Q, K1, Nh):
    """pd_loss_vectors_*: class-weighted cross entropy of every head (logits read through the decoder's [heads, B, Q, K] strides), the heads'
    BCE / dice sums over num_masks, and their gradients, against F.cross_entropy / sum / division in fp64."""
    from partdistillation_amd.functions import criterion_ops as cops
    torch.manual_seed(B + H + Q)
    raw = (torch.randn(H, B, Q, K1, device=DEV) * 4).requires_grad_(True)
    logits = raw.transpose(0, 1)                                          # [B, H, Q, K1] view
    tclass = torch.randint(0, K1, (B, H, Q), device=DEV)
    w = torch.rand(K1, device=DEV) + 0.1
    d_of_h = torch.tensor([H - 1] + list(range(H - 1)), device=DEV)
    bce, dice = torch.rand(H * Nh, device=DEV, requires_grad=True), torch.rand(H * Nh, device=DEV, requires_grad=True)
    nm = torch.tensor(7.0, device=DEV)
    assert cops.loss_vectors_supported(logits, tclass, w)
    vec = cops.loss_vectors(logits, tclass, w, d_of_h, bce, dice, nm)
    gv = torch.randn(3, H, device=DEV)
    (vec * gv).sum().backward()
    r64, b64, d64 = raw.detach().double().requires_grad_(True), bce.detach().double().requires_grad_(True), dice.detach().double().requires_grad_(True)
    l64 = r64.transpose(0, 1)
    nll = F.cross_entropy(l64.reshape(B * H, Q, K1).transpose(1, 2), tclass.reshape(B * H, Q), w.double(), reduction="none").reshape(B, H, Q)
    ce = (nll.sum((0, 2)) / w.double()[tclass].sum((0, 2)))[d_of_h]
    ref = torch.stack([ce, b64.reshape(H, Nh).sum(1) / 7.0, d64.reshape(H, Nh).sum(1) / 7.0])
    (ref * gv.double()).sum().backward()
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert rel(vec, ref) <= 2e-6
    assert rel(raw.grad, r64.grad) <= 5e-6 and rel(bce.grad, b64.grad) <= 1e-6 and rel(dice.grad, d64.grad) <= 1e-6


@pytest.mark.parametrize("counts,T", [([40, 40], 65536), ([40, 0, 7], 1000), ([100, 5], 1027), ([0, 3], 64), ([1], 4), ([48, 49, 16], 2050)])
def test_pair_logits_forward_and_gradients_vs_fp64(counts, T):
    """pd_pair_logits_*: the matched pairs' mask logits out[row(i)] = tok[b(i)] e[i] and both gradients against the same products in fp64:
    images without pairs, more pairs than one pass holds (> 48), token counts that are not multiples of the tiles (and of 4: scalar
    edges), a permuted output order.  fp32 products and sums: 2e-6 of the result's scale for K = 256, 1e-5 for the sums over T."""
    from partdistillation_amd.functions import criterion_ops as cops
    torch.manual_seed(T + len(counts))
    B, N, C = len(counts), sum(counts), 256
    tok = torch.randn(B, T, C, device=DEV, requires_grad=True)
    e = torch.randn(N, C, device=DEV, requires_grad=True)
    out_row = torch.randperm(N, device=DEV)
    assert cops.pair_logits_supported(tok, e)
    out = cops.pair_logits(tok, e, out_row, counts)
    g = torch.randn(N, T, device=DEV)
    out.backward(g)
    img = torch.repeat_interleave(torch.arange(B, device=DEV), torch.tensor(counts, device=DEV))
    t64, e64 = tok.detach().double().requires_grad_(True), e.detach().double().requires_grad_(True)
    ref = torch.empty(N, T, dtype=torch.float64, device=DEV)
    ref[out_row] = torch.einsum("ntc,nc->nt", t64[img], e64)
    ref.backward(g.double())
    scale = lambda x: float(x.abs().max().clamp_min(1e-30))
    assert float((out.double() - ref).abs().max()) <= 2e-6 * scale(ref)
    assert float((tok.grad.double() - t64.grad).abs().max()) <= 2e-6 * scale(t64.grad) + 1e-30
    assert float((e.grad.double() - e64.grad).abs().max()) <= 1e-5 * scale(e64.grad)
    for b, c in enumerate(counts):
        if c == 0:
            assert float(tok.grad[b].abs().max()) == 0.0


@pytest.mark.parametrize("shape,P", [((2, 4, 64, 48), 500), ((1, 1, 7, 9), 33), ((3, 2, 128, 128), 4096)])
def test_point_sample_masks_equals_grid_sample_of_the_float_copy(shape, P):
    """pd_point_sample_u8: the padded target masks sampled as stored (bool bytes) — bit-identical to the planar fp32 sampler on
    `.float()` copies and equal to F.grid_sample (bilinear, zeros, align_corners=False; matcher.py:130-139, criterion.py:196-199) within
    fp32 rounding, in both forms: every map of an image at the image's points, and one chosen map per row at the row's points"""
    from partdistillation_amd.functions import criterion_ops as cops
    from partdistillation_amd.functions import rowwise as rw
    B, nmax, H, W = shape
    torch.manual_seed(H + P)
    masks = torch.rand(shape, device=DEV) > 0.5
    coords = torch.rand(B, P, 2, device=DEV) * 1.1 - 0.05                    # some points outside [0, 1]: zeros padding
    assert cops.point_sample_masks_supported(masks, coords)
    got = cops.point_sample_masks(masks, coords, None, nmax).view(B, nmax, P)
    ref = F.grid_sample(masks.float(), 2.0 * coords.unsqueeze(2) - 1.0, mode="bilinear", padding_mode="zeros", align_corners=False).squeeze(3)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-6)
    assert torch.equal(got, rw.point_sample_planar(masks.float(), coords))
    N = 11
    idx = torch.randint(0, B * nmax, (N,), device=DEV)
    c2 = torch.rand(N, P, 2, device=DEV)
    got2 = cops.point_sample_masks(masks, c2, idx)
    ref2 = F.grid_sample(masks.float().view(B * nmax, 1, H, W)[idx], 2.0 * c2.unsqueeze(2) - 1.0, mode="bilinear", padding_mode="zeros",
                         align_corners=False)[:, 0, :, 0]
    torch.testing.assert_close(got2, ref2, rtol=1e-5, atol=1e-6)


def test_batched_criterion_with_ragged_target_counts_equals_the_torch_expressions(monkeypatch):
    """Images with DIFFERENT numbers of target masks (3 and 5: padded target columns, padded labels, per-problem column counts): the training
    losses and gradients of the proposal model with the criterion kernels (pd_matcher_costs, pd_point_sample_u8, pd_uncertain_points,
    pd_mask_point_losses_*) against the same model with the plain torch expressions (PD_CRITERION_KERNELS=0 — the path the reference goldens
    pin), identical weights and replayed random draws.  fp32, so what differs is summation order: losses to 1e-4, gradients to 2e-3 of
    their maximum (a flipped near-tie in the importance sampling moves single points)."""
    import os, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import common as C
    from partdistillation_amd.compat import build_model
    from partdistillation_amd.config import setup_cfg
    from partdistillation_amd.engine.synthetic import make_batch
    from partdistillation_amd.functions import criterion_ops as cops
    import partdistillation_amd.modeling, partdistillation_amd.proposal_model  # noqa: F401,E401
    cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd", "configs", "proposal_learning", "r50_mask2former.yaml"),
                    ["MODEL.MASK_FORMER.NUM_OBJECT_QUERIES", "20", "MODEL.MASK_FORMER.DEC_LAYERS", "3", "MODEL.SEM_SEG_HEAD.TRANSFORMER_ENC_LAYERS", "1",
                     "MODEL.MASK_FORMER.TRAIN_NUM_POINTS", "512", "SOLVER.AMP.ENABLED", "False"])
    torch.manual_seed(0)
    model = build_model(cfg).to(DEV).train()
    batch = make_batch(1, 128, n_parts=3, seed=5, device=DEV) + make_batch(1, 128, n_parts=5, seed=6, device=DEV)
    assert [len(b["instances"]) for b in batch] == [3, 5]
    res = {}
    for on in (True, False):
        monkeypatch.setattr(cops, "ENABLED", on)
        model.zero_grad(set_to_none=True)
        model.criterion.rand = C.ReplayRand(777)
        losses = model(batch)
        total = sum(losses.values())
        total.backward()
        res[on] = ({k: float(v) for k, v in losses.items()},
                   {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None and ("predictor" in n or "mask_features" in n)})
    la, lb = res[True][0], res[False][0]
    assert set(la) == set(lb) and len(la) == 9                      # three heads x (ce, mask, dice)
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-4 * max(abs(lb[k]), 1.0), (k, la[k], lb[k])
    ga, gb = res[True][1], res[False][1]
    assert set(ga) == set(gb) and len(ga) > 50
    worst = max(float((ga[n] - gb[n]).abs().max() / gb[n].abs().max().clamp_min(1e-12)) for n in gb)
    print(f"ragged targets, criterion kernels vs torch expressions: worst gradient deviation {worst:.2e} of the tensor maximum")
    assert worst <= 2e-3
