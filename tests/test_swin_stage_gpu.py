"""GPU tests of the fused Swin stage: the row kernel (include/pd_swin.h) against torch autograd on the same formula,
and the whole stage (modeling/backbone/swin_core.py) against the module-by-module path of the same BasicLayer (which is
pinned to the reference by the swin golden) — same weights, same input, same DropPath masks, bf16 autocast."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(got, want, frac, what):
    err = (got.double() - want.double()).abs().max().item()
    scale = max(want.double().abs().max().item(), 1e-30)
    assert err <= frac * scale, f"{what}: max abs err {err:.3e} > {frac} * {scale:.3e}"


@pytest.mark.parametrize("C", [64, 128, 192, 384, 512, 768, 1024, 1536])
@pytest.mark.parametrize("mode", ["plain", "ln1", "ln2"])
def test_row_kernel_matches_torch(C, mode):
    """plain: y = LN(x).  ln1: pending residual r (identity rows, DropPath scale), y window-major with zero rows.
    ln2: r window-major through the row map, y token-major."""
    from partdistillation_amd.functions import swin_rows as rows
    g = torch.Generator(device="cuda").manual_seed(C + len(mode))
    B, L, S = 3, 50, 64
    perm = torch.randperm(S, device="cuda", generator=g)
    tmap = perm[:L].to(torch.int32).contiguous()                      # token -> slot
    zero = perm[L:].to(torch.int32).contiguous()                      # padded slots
    x = torch.randn(B * L, C, device="cuda", generator=g)
    gamma = torch.randn(C, device="cuda", generator=g)
    beta = torch.randn(C, device="cuda", generator=g)
    rscale = torch.tensor([0.0, 1.0 / 0.7, 1.0 / 0.7], device="cuda")
    r_rows = {"plain": 0, "ln1": L, "ln2": S}[mode]
    y_rows = S if mode == "ln1" else L
    r = (torch.randn(B * r_rows, C, device="cuda", generator=g) * 2).to(torch.bfloat16) if mode != "plain" else None
    dy = torch.randn(B * y_rows, C, device="cuda", generator=g).to(torch.bfloat16)
    dsup = torch.randn(B * L, C, device="cuda", generator=g)
    rmap, ymap = (tmap if mode == "ln2" else None), (tmap if mode == "ln1" else None)
    zy, zr = (zero if mode == "ln1" else None), (zero if mode == "ln2" else None)

    s, y, st = rows.ln_fwd(x, r, rmap, r_rows, rscale if r is not None else None, gamma, beta, 1e-5, ymap, y_rows, zy, B, L)
    dgm, dbt = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    ds, dr = rows.ln_bwd(dy, ymap, y_rows, dsup, s, st, gamma, r is not None, rmap, r_rows, rscale if r is not None else None,
                         zr, dgm, dbt, B, L)

    xr, gr, br = x.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    rr = r.float().requires_grad_() if r is not None else None
    tok = tmap.long()
    sr = xr.view(B, L, C)
    if rr is not None:
        rv = rr.view(B, r_rows, C)
        rv = rv[:, tok] if mode == "ln2" else rv
        sr = sr + rscale.view(B, 1, 1) * rv
    yt = torch.nn.functional.layer_norm(sr, (C,), gr, br, 1e-5)
    if mode == "ln1":
        yr = torch.zeros(B, S, C, device="cuda").index_copy(1, tok, yt)
    else:
        yr = yt
    torch.testing.assert_close(s.view(B, L, C), sr.detach(), rtol=1e-6, atol=1e-6)
    _close(y.view(B, y_rows, C).float(), yr.detach(), 6e-3, "y (bf16 rounding)")
    if mode == "ln1":
        assert float(y.view(B, S, C)[:, zero.long()].abs().max()) == 0.0
    loss = (yr * dy.view(B, y_rows, C).float()).sum() + (sr * dsup.view(B, L, C)).sum()
    gs = torch.autograd.grad(loss, [xr, gr, br] + ([rr] if rr is not None else []))
    _close(ds, gs[0], 1e-5, "ds")
    _close(dgm, gs[1], 1e-4, "dgamma")
    _close(dbt, gs[2], 1e-4, "dbeta")
    if rr is not None:
        _close(dr.float(), gs[3], 6e-3, "dr (bf16 rounding)")
        if mode == "ln2":
            assert float(dr.view(B, S, C)[:, zero.long()].abs().max()) == 0.0
    # the MX-fp8 copies (include/pd_mx8.h) of y and dr: the same bf16 rows, quantised bit-exactly as oracle/mx8_ref.py does (zero rows included)
    from oracle import mx8_ref as MX
    for fmt in (0, 1):
        s2, y2, st2, (yq, ys) = rows.ln_fwd(x, r, rmap, r_rows, rscale if r is not None else None, gamma, beta, 1e-5, ymap, y_rows, zy, B, L, mx=fmt)
        # (C <= 192 without the MX copy runs the one-channel-per-lane kernels: another summation order, so "close", not "equal")
        _close(y2.float(), y.float(), 1e-2, "y next to its MX copy"); torch.testing.assert_close(s2, s, rtol=1e-6, atol=1e-6)
        rq, rs = MX.quantize(y2.cpu(), fmt)
        assert torch.equal(yq.cpu(), rq)
        live = (y2.float().view(-1, C // 32, 32).abs().amax(-1) > 0).cpu()         # an all-zero block's exponent byte is free (the kernel writes 0 for zero rows)
        assert torch.equal(ys.cpu()[live], rs[live])
        if r is not None:
            g8 = torch.zeros(8, 2, C, device="cuda")                                 # the column sums spread over 8 copies
            d2 = rows.ln_bwd(dy, ymap, y_rows, dsup, s, st, gamma, True, rmap, r_rows, rscale, zr, g8[0, 0], g8[0, 1], B, L, mx=fmt, n_rep=8, rep_stride=2 * C)
            _close(d2[0], ds, 1e-5, "ds next to the MX copy"); _close(d2[1].float(), dr.float(), 1e-2, "dr next to its MX copy")
            _close(g8.sum(0)[0], dgm, 1e-4, "dgamma from 8 copies"); _close(g8.sum(0)[1], dbt, 1e-4, "dbeta from 8 copies")
            rq, rs = MX.quantize(d2[1].cpu(), fmt)
            live = (d2[1].float().view(-1, C // 32, 32).abs().amax(-1) > 0).cpu()
            assert torch.equal(d2[2][0].cpu(), rq) and torch.equal(d2[2][1].cpu()[live], rs[live])


def test_row_kernel_rejects_unsupported_width():
    from partdistillation_amd import lib
    from partdistillation_amd.functions import swin_rows as rows
    x = torch.zeros(4, 96, device="cuda")
    with pytest.raises(lib.PdHipError, match="multiple of 64"):
        rows.ln_fwd(x, None, None, 0, None, torch.ones(96, device="cuda"), torch.zeros(96, device="cuda"), 1e-5, None, 4, None, 1, 4)
    with pytest.raises(RuntimeError, match="GPU only"):
        rows.ln_fwd(torch.zeros(4, 64), None, None, 0, None, torch.ones(64), torch.zeros(64), 1e-5, None, 4, None, 1, 4)


@pytest.mark.parametrize("dim,heads,H,W,depth,drop", [(64, 2, 30, 26, 2, 0.0), (128, 4, 24, 24, 2, 0.0), (64, 2, 30, 26, 4, 0.4)])
def test_fused_stage_matches_module_path(dim, heads, H, W, depth, drop):
    from partdistillation_amd.modeling.backbone import swin, swin_core
    torch.manual_seed(dim + H)
    rates = [drop * i / max(depth - 1, 1) for i in range(depth)]
    layer = swin.BasicLayer(dim=dim, depth=depth, num_heads=heads, window_size=12, drop_path=rates).cuda().train()
    for blk in layer.blocks:
        torch.nn.init.normal_(blk.attn.relative_position_bias_table, std=0.5)
        torch.nn.init.normal_(blk.norm1.weight, 1.0, 0.2)
        torch.nn.init.normal_(blk.norm2.bias, 0.0, 0.2)
    B = 2
    x0 = torch.randn(B, H * W, dim, device="cuda")
    go = torch.randn(B, H * W, dim, device="cuda")
    scales = None
    if drop > 0:
        keep = torch.tensor([1.0 - r for r in rates], device="cuda").view(depth, 1, 1)
        scales = (torch.rand(depth, 2, B, device="cuda") + keep).floor() / keep
        scales[1, 0, 0] = 0.0                                                     # at least one dropped branch
    res = {}
    orig_rand, orig_dp = torch.rand, swin.DropPath.forward
    for fused in (True, False):
        calls = {"n": 0}

        def dp_forward(self, t):                                                   # module path: the same masks, in call order
            if self.drop_prob == 0.0:
                return t
            k, j = divmod(calls["n"], 2)
            while rates[k] == 0.0:                                                 # blocks with rate 0 use nn.Identity
                k += 1
                calls["n"] += 2
            calls["n"] += 1
            return t * scales[k, j].view(B, 1, 1)
        try:
            swin.FUSED_STAGE = fused
            if scales is not None:
                if fused:
                    torch.rand = lambda *a, **kw: (scales * torch.tensor([1.0 - r for r in rates], device="cuda").view(depth, 1, 1)
                                                   + 1.0 - torch.tensor([1.0 - r for r in rates], device="cuda").view(depth, 1, 1)
                                                   - 1e-3).clamp_min(0.0)
                else:
                    swin.DropPath.forward = dp_forward
            layer.zero_grad()
            x = x0.clone().requires_grad_()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                assert swin_core.supported(layer, x)
                y = layer(x, H, W)[0]
            y.backward(go)
            res[fused] = (y.detach().float(), x.grad.clone(), {k: p.grad.clone() for k, p in layer.named_parameters()})
        finally:
            swin.FUSED_STAGE, torch.rand, swin.DropPath.forward = True, orig_rand, orig_dp
    _close(res[True][0], res[False][0], 2e-2, "stage output")
    _close(res[True][1], res[False][1], 3e-2, "input gradient")
    for k, gr in res[False][2].items():
        _close(res[True][2][k], gr, 6e-2, "grad " + k)


def test_fused_stage_in_eval_mode_and_without_grad():
    """inference use (proposal generation, part ranking): eval mode ignores DropPath, no_grad keeps nothing alive, bf16 input
    (the output of PatchMerging under autocast) is accepted"""
    from partdistillation_amd.modeling.backbone import swin, swin_core
    torch.manual_seed(1)
    layer = swin.BasicLayer(dim=128, depth=2, num_heads=4, window_size=12, drop_path=[0.2, 0.3]).cuda().eval()
    x = torch.randn(1, 24 * 36, 128, device="cuda")
    out = {}
    for fused in (True, False):
        swin.FUSED_STAGE = fused
        try:
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                assert swin_core.supported(layer, x.bfloat16()) == fused or not fused
                out[fused] = layer(x.bfloat16(), 24, 36)[0].float()
        finally:
            swin.FUSED_STAGE = True
    _close(out[True], out[False], 3e-2, "eval output (bf16 residual stream in the module path)")



@pytest.mark.parametrize("drop", [0.0, 0.3])
def test_recorded_stage_replays_equal_the_eager_launches(drop):
    """bf16 Linear weights (the training configuration: engine/flat_params.py shadows) put every Linear of the stage on pd_igemm_bf16 /
    pd_wgrad_bf16, and the block loops then run as recorded regions (cmdbuf.py): the first step records, the next ones replay with new
    inputs, new DropPath masks — and must give what the same launches issued one by one give"""
    from partdistillation_amd import cmdbuf
    from partdistillation_amd.modeling.backbone import swin, swin_core
    torch.manual_seed(7)
    H, W, dim, depth = 30, 26, 128, 2
    layer = swin.BasicLayer(dim=dim, depth=depth, num_heads=4, window_size=12, drop_path=[0.0, drop]).cuda().train()
    for blk in layer.blocks:
        torch.nn.init.normal_(blk.attn.relative_position_bias_table, std=0.5)
        for lin in (blk.attn.qkv, blk.attn.proj, blk.mlp.fc1, blk.mlp.fc2):
            lin.weight.data = lin.weight.data.to(torch.bfloat16)
    xs = [torch.randn(2, H * W, dim, device="cuda") for _ in range(3)]
    gos = [torch.randn(2, H * W, dim, device="cuda") for _ in range(3)]

    def run(enabled):
        out = []
        was = cmdbuf.ENABLED
        cmdbuf.ENABLED = enabled
        try:
            for i in range(3):
                torch.manual_seed(100 + i)                                         # the same DropPath draws in both modes
                layer.zero_grad(set_to_none=True)
                x = xs[i].clone().requires_grad_()
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = layer(x, H, W)[0]
                y.backward(gos[i])
                out.append((y.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in layer.named_parameters()}))
        finally:
            cmdbuf.ENABLED = was
        return out

    n0 = len(swin_core._RECS)
    rec = run(True)
    assert len(swin_core._RECS) == n0 + 2, "the stage did not run as recorded regions"
    eager = run(False)
    for i in range(3):
        assert torch.equal(rec[i][0], eager[i][0]), f"step {i}: output"
        assert torch.equal(rec[i][1], eager[i][1]), f"step {i}: input gradient"
        for k, g in eager[i][2].items():
            _close(rec[i][2][k].float(), g.float(), 1e-5, f"step {i}: grad {k}")   # LayerNorm / table gradients: atomics
    assert not torch.equal(rec[1][0], rec[2][0])


@pytest.mark.parametrize("rows,C", [(1000, 128), (777, 192), (513, 384), (300, 768), (260, 1024), (130, 1536), (5, 256), (301, 2048), (77, 3072), (200, 1280)])
def test_rows_layer_norm_f32_vs_torch(rows, C):
    """pd_layernorm_rows_f32_fwd / _bwd (the Swin stages' output norms, reference swin.py:675-680) against torch.nn.functional.layer_norm in
    float64: every stage width of Swin-T / B / L and the 4 C widths of their patch-merging norms (:339), outputs and all three gradients."""
    from partdistillation_amd.functions import swin_rows
    g = torch.Generator(device="cuda").manual_seed(rows + C)
    x = (torch.randn(2, rows, C, device="cuda", generator=g) * 3 + 0.5).requires_grad_()
    ln = torch.nn.LayerNorm(C).cuda()
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.normal_(0, 0.3)
    assert swin_rows.rows_layer_norm_supported(x, ln)
    y = swin_rows.rows_layer_norm(x, ln)
    go = torch.randn(y.shape, device="cuda", generator=g)
    gx, gw, gb = torch.autograd.grad(y, (x, ln.weight, ln.bias), go)
    xd, wd, bd = x.detach().double().requires_grad_(), ln.weight.detach().double().requires_grad_(), ln.bias.detach().double().requires_grad_()
    ref = torch.nn.functional.layer_norm(xd, (C,), wd, bd, ln.eps)
    rx, rw_, rb = torch.autograd.grad(ref, (xd, wd, bd), go.double())
    torch.testing.assert_close(y.double(), ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gx.double(), rx, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(gw.double(), rw_, rtol=1e-4, atol=1e-4 * rw_.abs().max().item())
    torch.testing.assert_close(gb.double(), rb, rtol=1e-4, atol=1e-4 * rb.abs().max().item())


def test_drop_path_scales_of_all_stages_are_drawn_at_once_and_used():
    """swin_core.draw_drop_path (reference DropPath :35-51, one Bernoulli(keep) / keep scale per block, branch and image): every stage gets its own
    [depth, 2, B] slice of ONE draw, the values are 0 or 1 / keep_prob of that block, and run_stage consumes the slice (a second forward without a new
    draw falls back to drawing its own)."""
    from partdistillation_amd.modeling.backbone import swin, swin_core
    torch.manual_seed(3)
    net = swin.SwinTransformer(embed_dim=64, depths=[2, 2], num_heads=[2, 4], window_size=12, drop_path_rate=0.5, out_indices=(0, 1)).cuda().train()
    B = 3
    swin_core.draw_drop_path(net, B, torch.device("cuda"))
    rates = [float(getattr(blk.drop_path, "drop_prob", 0.0) or 0.0) for layer in net.layers for blk in layer.blocks]     # (the first block's is nn.Identity: rate 0)
    o = 0
    for layer in net.layers:
        d = len(layer.blocks)
        dp = layer._dp_drawn
        assert tuple(dp.shape) == (d, 2, B) and dp.is_contiguous()
        for k in range(d):
            keep = 1.0 - rates[o + k]
            v = dp[k].flatten().tolist()
            assert all(abs(t) < 1e-12 or abs(t - 1.0 / keep) < 1e-5 for t in v), (k, v)
        o += d
    assert net.layers[0]._dp_drawn.data_ptr() != net.layers[1]._dp_drawn.data_ptr()
    x = torch.randn(B, 3, 96, 96, device="cuda")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(x)                                       # draws again (the slices above are replaced), runs the fused stages
    assert all(not hasattr(layer, "_dp_drawn") for layer in net.layers), "run_stage consumes its slice"
    assert all(torch.isfinite(v.float()).all() for v in out.values())
    net.eval()
    swin_core.draw_drop_path(net, B, torch.device("cuda"))
    assert all(not hasattr(layer, "_dp_drawn") for layer in net.layers), "no scales in eval mode"


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 6, 128), (1, 4, 10, 192), (3, 6, 6, 256), (2, 4, 4, 384), (1, 6, 4, 512), (2, 2, 4, 768), (1, 2, 2, 64)])
def test_patch_merging_gather_layer_norm_vs_torch(B, H, W, C):
    """pd_swin_merge_ln_fwd / _bwd (reference swin.py:325-339: the four strided slices concatenated, then LayerNorm over 4 C) against the
    permuted view + torch.nn.functional.layer_norm in float64: the bf16 output within bf16 rounding, dx / dgamma / dbeta from a bf16 upstream gradient."""
    from partdistillation_amd.functions import swin_rows
    g = torch.Generator(device="cuda").manual_seed(H * W + C)
    x = (torch.randn(B, H * W, C, device="cuda", generator=g) * 2 + 0.3).requires_grad_()
    ln = torch.nn.LayerNorm(4 * C).cuda()
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.normal_(0, 0.3)
    assert swin_rows.merge_layer_norm_supported(x, H, W, ln)
    y = swin_rows.merge_layer_norm(x, H, W, ln)
    assert y.dtype == torch.bfloat16 and tuple(y.shape) == (B, H * W // 4, 4 * C)
    go = torch.randn(y.shape, device="cuda", generator=g).to(torch.bfloat16)
    gx, gw, gb = torch.autograd.grad(y, (x, ln.weight, ln.bias), go)
    xd, wd, bd = x.detach().double().requires_grad_(), ln.weight.detach().double().requires_grad_(), ln.bias.detach().double().requires_grad_()
    xm = xd.view(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 4, 2, 5).reshape(B, -1, 4 * C)       # channel block 2 cp + rp (modeling/backbone/swin.py)
    # the same rows as the reference's cat([x[0::2, 0::2], x[1::2, 0::2], x[0::2, 1::2], x[1::2, 1::2]], -1)
    x4 = xd.view(B, H, W, C)
    cat = torch.cat([x4[:, 0::2, 0::2], x4[:, 1::2, 0::2], x4[:, 0::2, 1::2], x4[:, 1::2, 1::2]], -1).reshape(B, -1, 4 * C)
    assert torch.equal(xm, cat)
    ref = torch.nn.functional.layer_norm(xm, (4 * C,), wd, bd, ln.eps)
    rx, rw_, rb = torch.autograd.grad(ref, (xd, wd, bd), go.double())
    _close(y.double(), ref, 6e-3, "y (bf16 rounding)")
    torch.testing.assert_close(gx.double(), rx, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(gw.double(), rw_, rtol=1e-4, atol=1e-4 * rw_.abs().max().item())
    torch.testing.assert_close(gb.double(), rb, rtol=1e-4, atol=1e-4 * rb.abs().max().item())
    with pytest.raises(RuntimeError, match="even H and W"):
        from partdistillation_amd import lib
        lib.check(lib.load().pd_swin_merge_ln_fwd(x.data_ptr(), ln.weight.data_ptr(), ln.bias.data_ptr(), 1e-5, y.data_ptr(), y.data_ptr(), y.data_ptr(), 1, 3, 4, C,
                                                  lib.current_stream()))


@pytest.mark.parametrize("B,L,C,scaled,with_dsum", [(2, 50, 128, True, True), (3, 33, 192, False, True), (2, 20, 512, True, False), (1, 7, 1536, True, True),
                                                     (2, 9, 2048, False, False)])
def test_stage_tail_layer_norm_kernels_vs_torch(B, L, C, scaled, with_dsum):
    """pd_swin_tail_ln_fwd / _bwd: s = cur + scale[image] * float(r) exactly as the ATen chain forms it (cast, product, sum), y = LayerNorm(s) and the backward's
    (dsum + LayerNorm'(dy), its scaled bf16 copy, dgamma, dbeta) against float64"""
    from partdistillation_amd.functions import swin_rows as rows
    g = torch.Generator(device="cuda").manual_seed(B * L + C)
    cur = torch.randn(B * L, C, device="cuda", generator=g) * 2
    r = torch.randn(B * L, C, device="cuda", generator=g).to(torch.bfloat16)
    sc = torch.tensor([1.0 / 0.7, 0.0, 1.0 / 0.9][:B], device="cuda") if scaled else None          # (products that round)
    gamma = torch.rand(C, device="cuda", generator=g) + 0.5
    beta = torch.randn(C, device="cuda", generator=g) * 0.3
    assert rows.tail_ln_supported(C, gamma, beta)
    s, y, st = rows.tail_ln_fwd(cur, r, sc, L, gamma, beta, 1e-5)
    want_s = r.view(B, L, C).float()
    if sc is not None:
        want_s = want_s * sc.view(B, 1, 1)
    want_s = want_s.add_(cur.view(B, L, C)).view(B * L, C)
    assert torch.equal(s, want_s), "the stage output is the ATen chain's, bit for bit"
    sd, gd, bd = s.double().requires_grad_(), gamma.double().requires_grad_(), beta.double().requires_grad_()
    ref = torch.nn.functional.layer_norm(sd, (C,), gd, bd, 1e-5)
    torch.testing.assert_close(y.double(), ref, rtol=1e-5, atol=1e-5)
    dy = torch.randn(B * L, C, device="cuda", generator=g)
    dsum = torch.randn(B * L, C, device="cuda", generator=g) if with_dsum else None
    dsup, df, dg, db = rows.tail_ln_bwd(dy, dsum, s, st, gamma, sc, L)
    rs_, rg, rb = torch.autograd.grad(ref, (sd, gd, bd), dy.double())
    want = rs_ + (dsum.double() if dsum is not None else 0.0)
    torch.testing.assert_close(dsup.double(), want, rtol=1e-4, atol=1e-5)
    want_df = dsup.view(B, L, C) * sc.view(B, 1, 1) if sc is not None else dsup.view(B, L, C)
    assert torch.equal(df.view(B, L, C), want_df.to(torch.bfloat16)), "the 16-bit copy is the rounded scaled stream gradient"
    torch.testing.assert_close(dg.double(), rg, rtol=1e-4, atol=1e-4 * rg.abs().max().item())
    torch.testing.assert_close(db.double(), rb, rtol=1e-4, atol=1e-4 * rb.abs().max().item())


@pytest.mark.parametrize("drop", [0.0, 0.3])
def test_fused_stage_tail_equals_the_aten_chain_plus_row_norm(drop):
    """a BasicLayer with its output norm: the stage's closing kernel (pd_swin_tail_ln_*) against the ATen epilogue + pd_layernorm_rows_f32 — same DropPath draw,
    identical stage output, norm output / all gradients within fp32 summation order"""
    from partdistillation_amd.modeling.backbone import swin, swin_core
    from partdistillation_amd.functions import swin_rows
    torch.manual_seed(11)
    dim, depth, H, W, B = 128, 2, 24, 24, 2
    layer = swin.BasicLayer(dim=dim, depth=depth, num_heads=4, window_size=12, drop_path=[0.0, drop]).cuda().train()
    ln = torch.nn.LayerNorm(dim).cuda()
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.normal_(0, 0.2)
    x0 = torch.randn(B, H * W, dim, device="cuda")
    g1, g2 = torch.randn(B, H * W, dim, device="cuda"), torch.randn(B, H * W, dim, device="cuda")
    res = {}
    for fused in (True, False):
        swin_core.TAIL_FUSED = fused
        try:
            torch.manual_seed(5)                                     # the same DropPath draw
            x = x0.clone().requires_grad_()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = layer(x, H, W, ln)
            xo, normed = out[0], layer.normed
            assert (normed is not None) == fused
            if normed is None:
                normed = swin_rows.rows_layer_norm(xo, ln)
            params = [x, ln.weight, ln.bias] + [p for p in layer.parameters()]
            grads = torch.autograd.grad((xo * g1).sum() + (normed * g2).sum(), params, allow_unused=True)
            res[fused] = (xo.detach(), normed.detach(), grads)
        finally:
            swin_core.TAIL_FUSED = True
    assert torch.equal(res[True][0], res[False][0])
    _close(res[True][1], res[False][1], 1e-5, "norm output")
    for a, b in zip(res[True][2], res[False][2]):
        assert (a is None) == (b is None)
        if a is not None:
            _close(a.float(), b.float(), 2e-2 if a.dtype == torch.bfloat16 else 2e-3, "gradient")
