"""GPU parity of the token-wise kernels (include/pd_rowwise.h) against plain PyTorch fp32, and of the fused decoder
core (functions/decoder_core.py: hand-written backward) against the module-by-module autograd path of the same
decoder, which the reference goldens pin (tests/test_product_gpu.py)."""
import pytest
import torch
import torch.nn.functional as F

import common as C

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _r(shape, seed, scale=1.0, dtype=torch.float32):
    return (C.seeded(shape, seed) * scale).to(DEV).to(dtype)


@pytest.mark.parametrize("rows,Cw,xdt,cdt", [(200, 256, torch.bfloat16, torch.bfloat16), (200, 256, torch.float32, torch.float32),
                                              (4099, 256, torch.float32, torch.float32), (37, 512, torch.bfloat16, torch.bfloat16),
                                              (5, 1024, torch.float32, torch.bfloat16),
                                              # >= 8 192 rows: the backward's fat workgroups (16 / 8 wavefronts, <= 256 workgroups)
                                              (43008, 256, torch.float32, torch.float32), (8195, 256, torch.bfloat16, torch.bfloat16),
                                              (9000, 512, torch.float32, torch.bfloat16), (8192, 1024, torch.float32, torch.float32)])
def test_add_layernorm_fwd_bwd_vs_torch(rows, Cw, xdt, cdt):
    from partdistillation_amd.functions import rowwise as rw
    B = 2 if rows % 2 == 0 else 1
    x, res = _r((rows, Cw), 1, 2.0, xdt), _r((rows, Cw), 2)
    gamma, beta = _r((Cw,), 3) + 1.0, _r((Cw,), 4)
    pos = _r((rows // B, Cw), 5)
    z, y, y_c, ypos_c, mean, rstd = rw.add_ln_fwd(x, res, gamma, beta, 1e-5, c_dtype=cdt, want_yc=True, pos=pos, pos_div=B, want_ypos=True)
    zr = x.float() + res
    yr = F.layer_norm(zr, (Cw,), gamma, beta, 1e-5)
    tol = dict(rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(z, zr, **tol)
    torch.testing.assert_close(y, yr, **tol)
    torch.testing.assert_close(mean, zr.mean(1), **tol)
    torch.testing.assert_close(rstd, (zr.var(1, unbiased=False) + 1e-5).rsqrt(), rtol=1e-4, atol=1e-5)
    ctol = dict(rtol=1e-2, atol=1e-2) if cdt == torch.bfloat16 else tol
    torch.testing.assert_close(y_c.float(), yr, **ctol)
    torch.testing.assert_close(ypos_c.float(), yr + pos.repeat_interleave(B, 0), **ctol)
    # backward: g = dy + dy2 + dy_c + dypos_c
    dy, dy2 = _r((rows, Cw), 6), _r((rows, Cw), 7)
    dy_c, dypos_c = _r((rows, Cw), 8, 1.0, cdt), _r((rows, Cw), 9, 1.0, cdt)
    acc = torch.zeros((3, Cw), device=DEV)
    dpos = torch.zeros((rows // B, Cw), device=DEV)
    dz, dz_c = rw.add_ln_bwd(z, mean, rstd, gamma, dy=dy, dy2=dy2, dy_c=dy_c, dypos_c=dypos_c, dz_c_dtype=cdt, dgamma=acc[0],
                             dbeta=acc[1], dbias=acc[2], dpos_acc=dpos, pos_div=B)
    zt = zr.clone().requires_grad_()
    gt, bt = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    g = dy + dy2 + dy_c.float() + dypos_c.float()
    F.layer_norm(zt, (Cw,), gt, bt, 1e-5).backward(g)
    btol = dict(rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dz, zt.grad, **btol)
    torch.testing.assert_close(dz_c.float(), zt.grad, **(dict(rtol=1e-2, atol=2e-2) if cdt == torch.bfloat16 else btol))
    stol = dict(rtol=1e-3, atol=1e-3 * max(1.0, rows ** 0.5))
    torch.testing.assert_close(acc[0], gt.grad, **stol)
    torch.testing.assert_close(acc[1], bt.grad, **stol)
    torch.testing.assert_close(acc[2], zt.grad.sum(0), **stol)
    torch.testing.assert_close(dpos, dypos_c.float().view(rows // B, B, Cw).sum(1), **btol)


def test_add_layernorm_autograd_function_matches_layer_norm():
    from partdistillation_amd.functions.rowwise import add_layer_norm
    norm = torch.nn.LayerNorm(256).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(_r((256,), 11) + 1)
        norm.bias.copy_(_r((256,), 12))
    x, res, pos = _r((2, 301, 256), 13).requires_grad_(), _r((2, 301, 256), 14).requires_grad_(), _r((2, 301, 256), 15).requires_grad_()
    y, yp = add_layer_norm(x, res, norm, pos)
    w1, w2 = _r(y.shape, 16), _r(y.shape, 17)
    ((y * w1).sum() + (yp * w2).sum()).backward()
    got = [x.grad.clone(), res.grad.clone(), pos.grad.clone(), norm.weight.grad.clone(), norm.bias.grad.clone()]
    for t in (x, res, pos, norm.weight, norm.bias):
        t.grad = None
    yr = F.layer_norm(x + res, (256,), norm.weight, norm.bias, norm.eps)
    ((yr * w1).sum() + ((yr + pos) * w2).sum()).backward()
    torch.testing.assert_close(y, yr, rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(yp, yr + pos, rtol=1e-5, atol=2e-5)
    for a, b in zip(got, [x.grad, res.grad, pos.grad, norm.weight.grad, norm.bias.grad]):
        torch.testing.assert_close(a, b, rtol=1e-3, atol=2e-3)


@pytest.mark.parametrize("rows,N,dt", [(200, 256, torch.bfloat16), (200, 2048, torch.bfloat16), (32768, 256, torch.bfloat16),
                                        (77, 768, torch.float32), (0, 256, torch.float32)])
def test_colsum_and_relu_bwd_vs_torch(rows, N, dt):
    from partdistillation_amd.functions import rowwise as rw
    x = _r((rows, N), 21, 1.0, dt)
    acc = torch.ones(N, device=DEV)
    rw.colsum_acc(x, acc)
    torch.testing.assert_close(acc, 1 + x.float().sum(0), rtol=1e-3, atol=1e-3 * max(1.0, rows ** 0.5))
    h = _r((rows, N), 22, 1.0, dt)
    dh = x.clone()
    acc2 = torch.zeros(N, device=DEV)
    rw.relu_bwd_colsum(dh, h, acc2)
    want = x * (h > 0)
    assert torch.equal(dh, want)
    torch.testing.assert_close(acc2, want.float().sum(0), rtol=1e-3, atol=1e-3 * max(1.0, rows ** 0.5))


@pytest.mark.parametrize("cdt", [torch.float32, torch.bfloat16])
def test_mem_prep_fwd_bwd_vs_torch(cdt):
    from partdistillation_amd.functions import rowwise as rw
    B, Cw, H, W = 2, 256, 5, 7
    tokens = _r((B, H * W + 3, Cw), 31)                                  # a level is a slice of the encoder's token tensor
    x = tokens[:, 3:].transpose(1, 2).reshape(B, Cw, H, W)               # the channels-last view the pixel decoder hands over
    lvl, pos = _r((Cw,), 32), _r((H * W, Cw), 33)
    mem, mempos = rw.mem_prep_fwd(x, lvl, pos, cdt)
    want = (x.flatten(2) + lvl[None, :, None]).permute(2, 0, 1)          # [HW, B, C]
    tol = dict(rtol=1e-2, atol=1e-2) if cdt == torch.bfloat16 else dict(rtol=0, atol=0)
    torch.testing.assert_close(mem.float().view(H * W, B, Cw), want, **tol)
    torch.testing.assert_close(mempos.float().view(H * W, B, Cw), want + pos[:, None, :], **tol)
    mem2, _ = rw.mem_prep_fwd(x.contiguous(), lvl, pos, cdt)              # NCHW-contiguous input takes the copy route
    assert torch.equal(mem2, mem)
    dmem, dmempos = _r((H * W * B, Cw), 34, 1.0, cdt), _r((H * W * B, Cw), 35, 1.0, cdt)
    dtok = rw.mem_prep_bwd(dmem, dmempos, B, H, W, Cw)
    want = (dmem.float() + dmempos.float()).view(H * W, B, Cw).transpose(0, 1)
    torch.testing.assert_close(dtok, want, rtol=0, atol=0)


def test_attn_mask_u8_vs_torch():
    from partdistillation_amd.functions import rowwise as rw
    logits = _r((2, 100, 1333), 41)
    logits[0, 7] = -logits[0, 7].abs() - 0.1                              # blocked everywhere -> released
    logits[1, 99] = logits[1, 99].abs()
    m = rw.attn_mask_u8(logits)
    want = logits < 0
    want = want & ~want.all(-1, keepdim=True)
    assert torch.equal(m.bool(), want) and not m[0, 7].any() and want.any()
    mb = rw.attn_mask_u8(logits.bfloat16())
    wb = logits.bfloat16() < 0
    assert torch.equal(mb.bool(), wb & ~wb.all(-1, keepdim=True))


@pytest.mark.parametrize("n", [16384, 4096, 1024, 8, 1336])
def test_attn_mask_u8_bf16_rows_of_eight_with_special_values(n):
    """the 16-byte-piece kernel (bf16 rows, n % 8 == 0, n <= 16 384) decides v < 0 on the bf16 BITS: -0.0, NaN of either sign and +-inf must
    come out as torch's comparison has them; an all-blocked row is released; the last piece of a short row is partial"""
    from partdistillation_amd.functions import rowwise as rw
    logits = _r((3, 7, n), 43 + n).bfloat16()
    sp = torch.tensor([-0.0, 0.0, float("nan"), -float("nan"), float("inf"), -float("inf"), -1e-30, 1e-30], device=logits.device).bfloat16()
    logits[0, 0, :8] = sp
    logits[1, 3] = -logits[1, 3].abs() - 0.1
    logits[2, 6] = logits[2, 6].abs()
    m = rw.attn_mask_u8(logits)
    wb = logits < 0
    want = wb & ~wb.all(-1, keepdim=True)
    assert m.dtype == torch.uint8 and torch.equal(m.bool(), want) and not m[1, 3].any() and int(m.max()) == 1


# ----------------------------------------------------------------------------- fused decoder core
def _decoder_pair(dec_layers=4, queries=20, seed=900):
    from test_product_gpu import build_decoder
    cfg = dict(C.C1, queries=queries, dec_layers=dec_layers - 1, dec_ffn=512)
    a = build_decoder(cfg)
    table = C.table_of(a.state_dict())
    a.load_state_dict(C.seeded_weights(table, seed), strict=False)
    a = a.to(DEV)
    b = build_decoder(cfg).to(DEV)
    b.load_state_dict(a.state_dict())
    b.fused_core = False
    return a, b


def _decoder_inputs(seed, B=2, S=64):
    toks = [_r((B, (S // st) ** 2, 256), seed + i, 0.5) for i, st in enumerate((32, 16, 8))]
    ms = [t.transpose(1, 2).reshape(B, 256, S // st, S // st) for t, st in zip(toks, (32, 16, 8))]
    return toks, ms, _r((B, 256, S // 4, S // 4), seed + 7)


def _run_decoder(dec, seed, autocast=False):
    toks, ms, mf = _decoder_inputs(seed)
    toks = [t.requires_grad_() for t in toks]
    ms = [t.transpose(1, 2).reshape(m.shape) for t, m in zip(toks, ms)]
    mf.requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        out = dec(ms, mf)
    loss = (out["all_masks"].float() * _r(out["all_masks"].shape, seed + 20)).sum() * 0.05
    loss = loss + (out["all_logits"].float() * _r(out["all_logits"].shape, seed + 21)).sum()
    loss = loss + (out["decoder_output"].float() * _r(out["decoder_output"].shape, seed + 22)).sum() * 0.1
    loss.backward()
    grads = {k: p.grad.float().clone() for k, p in dec.named_parameters() if p.grad is not None}
    grads.update({f"tok{i}": t.grad.clone() for i, t in enumerate(toks)})
    grads["mf"] = mf.grad.clone()
    return out, loss.detach(), grads


def test_fused_decoder_core_fp32_matches_module_path():
    fused, plain = _decoder_pair()
    assert fused._core_dtype([torch.empty(1, device=DEV)]) == torch.float32
    o1, l1, g1 = _run_decoder(fused, 950)
    o2, l2, g2 = _run_decoder(plain, 950)
    torch.testing.assert_close(o1["all_logits"], o2["all_logits"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(o1["all_masks"], o2["all_masks"], rtol=1e-3, atol=2e-3)
    torch.testing.assert_close(o1["decoder_output"], o2["decoder_output"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(l1, l2, rtol=1e-4, atol=1e-2)
    assert set(g1) == set(g2), set(g1) ^ set(g2)
    for k in g2:
        scale = g2[k].abs().max().clamp_min(1e-6)
        err = ((g1[k] - g2[k]).abs().max() / scale).item()
        assert err < 2e-3, (k, err)


def test_fused_decoder_core_bf16_close_to_fp32():
    """bf16 GEMM operands (shadow weights + autocast), fp32 residual stream: within bf16 noise of the fp32 result"""
    fused, plain = _decoder_pair(seed=901)
    for n, p in fused.named_parameters():
        mod = n.rsplit(".", 1)[0]
        if ("norm" not in mod and "query_" not in n and "level_embed" not in n and "class_embed" not in n):
            p.data = p.data.to(torch.bfloat16)
    o1, l1, g1 = _run_decoder(fused, 960, autocast=True)
    assert fused._core_dtype([torch.empty(1, device=DEV)]) is None      # outside autocast bf16 weights do not qualify
    o2, l2, g2 = _run_decoder(plain, 960)
    err = (o1["decoder_output"].float() - o2["decoder_output"]).abs()          # LayerNorm outputs: O(1) values
    assert err.mean().item() < 0.02 and (err > 0.1).float().mean().item() < 0.01, (err.mean().item(), err.max().item())
    agree = ((o1["all_masks"].float() > 0) == (o2["all_masks"] > 0)).float().mean().item()
    assert agree > 0.97, agree
    for k in ("query_feat.weight", "query_embed.weight", "level_embed.weight", "decoder_norm.weight",
              "transformer_ffn_layers.1.linear1.weight", "transformer_cross_attention_layers.0.multihead_attn.in_proj_weight",
              "transformer_self_attention_layers.2.self_attn.in_proj_bias", "tok0", "tok2"):
        a, b = g1[k], g2[k]
        cos = F.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
        assert cos > 0.98, (k, cos)


# ----------------------------------------------------------------------------- MSDeformAttn prep + fused encoder core
@pytest.mark.parametrize("L,P", [(3, 4), (2, 3), (4, 4)])
def test_msda_prep_fwd_bwd_vs_torch(L, P):
    from partdistillation_amd.functions.encoder_core import msda_prep_bwd, msda_prep_fwd
    T, M = 333, 8
    offs, logits = _r((T, M * L * P * 2), 51, 3.0), _r((T, M * L * P), 52, 2.0)
    ref = _r((T, L, 2), 53).abs()
    shapes = torch.tensor([[8, 12], [16, 24], [32, 48], [5, 7]][:L], dtype=torch.long, device=DEV)
    loc, attn = msda_prep_fwd(offs, logits, ref, shapes, M, L, P)
    ot, lt = offs.clone().requires_grad_(), logits.clone().requires_grad_()
    normalizer = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
    loc_r = ref[:, None, :, None, :] + ot.view(T, M, L, P, 2) / normalizer[None, None, :, None, :]
    attn_r = F.softmax(lt.view(T, M, L * P), -1).view(T, M, L, P)
    assert torch.equal(loc, loc_r.detach())                              # same two roundings as the reference expression
    torch.testing.assert_close(attn, attn_r.detach(), rtol=1e-6, atol=1e-7)
    gloc, gattn = _r(loc.shape, 54), _r(attn.shape, 55)
    d_offs, d_logits = msda_prep_bwd(gloc, gattn, attn, shapes, T, M, L, P)
    assert d_offs.stride(0) == 3 * M * L * P                           # column ranges of one buffer
    ((loc_r * gloc).sum() + (attn_r * gattn).sum()).backward()
    torch.testing.assert_close(d_offs, ot.grad, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(d_logits, lt.grad, rtol=1e-4, atol=1e-6)
    from partdistillation_amd.functions.encoder_core import prep_amax_supported
    if prep_amax_supported(M, L, P):               # the same launch with the rows' absolute maxima (pd_msda_prep_bwd_amax): bit-identical outputs
        out = torch.empty((T, 3 * M * L * P), dtype=torch.float32, device=DEV)
        am = torch.full((T,), -1.0, device=DEV)
        d2, l2 = msda_prep_bwd(gloc, gattn, attn, shapes, T, M, L, P, out=out, amax=am)
        assert torch.equal(d2, d_offs) and torch.equal(l2, d_logits) and torch.equal(am, out.abs().amax(1))


def test_msda_forward_amax_is_the_forward_plus_exact_row_maxima():
    from partdistillation_amd import MultiScaleDeformableAttention as MSDA
    from partdistillation_amd.functions.encoder_core import msda_forward_amax
    B, M, D, L, P = 2, 8, 32, 3, 4
    shapes = torch.tensor([[6, 9], [12, 18], [24, 36]], dtype=torch.long, device=DEV)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    value = _r((B, S, M, D), 61, 2.0) * torch.logspace(-3, 2, S, device=DEV).view(1, S, 1, 1)
    loc = torch.rand((B, S, M, L, P, 2), device=DEV, generator=torch.Generator(device=DEV).manual_seed(62)) * 1.2 - 0.1
    attn = torch.softmax(_r((B, S, M, L * P), 63), -1).view(B, S, M, L, P)
    want = MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, attn, 64)
    am = torch.zeros(B * S, device=DEV)
    got = msda_forward_amax(value, shapes, lsi, loc, attn, 64, am)
    assert torch.equal(got, want) and torch.equal(am, want.view(B * S, -1).abs().amax(1))


def test_fused_encoder_core_matches_module_path():
    from test_product_gpu import build_pixel_decoder
    cfg = dict(C.C1, enc_layers=3, enc_ffn=512)
    a = build_pixel_decoder(cfg)
    a.load_state_dict(C.seeded_weights(C.table_of(a.state_dict()), 970), strict=False)
    a = a.to(DEV)
    b = build_pixel_decoder(cfg).to(DEV)
    b.load_state_dict(a.state_dict())
    b.transformer.encoder.fused_core = False
    feats = {f"res{i + 2}": _r((2, c, 96 // s, 128 // s), 980 + i, 0.5) for i, (c, s) in enumerate(zip(cfg["channels"], (4, 8, 16, 32)))}
    res = []
    for m in (a, b, a, b):      # first pass: MIOpen picks (and caches) its conv algorithms; the second pass is compared
        f = {k: v.clone().requires_grad_() for k, v in feats.items()}
        mf, low, ms = m.forward_features(f)
        loss = (mf * _r(mf.shape, 990)).sum() + sum((t * _r(t.shape, 991 + i)).sum() for i, t in enumerate(ms))
        loss.backward()
        g = {k: p.grad.clone() for k, p in m.named_parameters()}
        g.update({k: v.grad.clone() for k, v in f.items()})
        for p_ in m.parameters():
            p_.grad = None
        res.append((mf.detach(), [t.detach() for t in ms], g))
    (mf1, ms1, g1), (mf2, ms2, g2) = res[2:]
    torch.testing.assert_close(mf1, mf2, rtol=1e-4, atol=1e-4)
    for x, y in zip(ms1, ms2):
        torch.testing.assert_close(x, y, rtol=1e-4, atol=1e-4)
    assert set(g1) == set(g2)
    for k in g2:
        scale = g2[k].abs().max().clamp_min(1e-6)
        err = ((g1[k] - g2[k]).abs().max() / scale).item()
        # convolution weight gradients come from MIOpen, whose algorithm choice (and split-K atomics order) differs between
        # runs by up to ~5e-3 of the tensor's scale on this pool; everything else is the fused core's own arithmetic
        assert err < (2e-2 if g2[k].dim() == 4 else 2e-3), (k, err)


# ----------------------------------------------------------------------------- point sampling + sparse-mask criterion
@pytest.mark.parametrize("B,C,H,W,P", [(2, 256, 64, 48, 1000), (1, 8, 5, 7, 33)])
def test_point_sample_nhwc_vs_grid_sample(B, C, H, W, P):
    from partdistillation_amd.functions.rowwise import point_sample_nhwc
    x = _r((B, C, H, W), 61).contiguous(memory_format=torch.channels_last)
    coords = torch.rand((B, P, 2), device=DEV, generator=torch.Generator(device=DEV).manual_seed(5)) * 1.1 - 0.05   # some outside
    got = point_sample_nhwc(x, coords)
    ref = F.grid_sample(x, 2.0 * coords.unsqueeze(2) - 1.0, mode="bilinear", padding_mode="zeros", align_corners=False).squeeze(3)
    torch.testing.assert_close(got, ref.transpose(1, 2), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(point_sample_nhwc(x.contiguous(), coords), got, rtol=0, atol=0)      # NCHW input: copied first


def test_sparse_mask_criterion_matches_dense():
    """decoder without dense masks + batched criterion (mask features sampled, matched masks only) == dense path"""
    from test_product_gpu import build_criterion, build_decoder, dev_targets
    cfg = dict(C.C1, queries=20, dec_layers=3, dec_ffn=512, num_points=256)
    dec = build_decoder(cfg)
    dec.load_state_dict(C.seeded_weights(C.table_of(dec.state_dict()), 990), strict=False)
    dec = dec.to(DEV).train()
    crit = build_criterion(cfg)
    targets = dev_targets(C.make_targets(dict(cfg, image=64, batch=2), 77, size=64))
    res = []
    for dense in (True, False):
        dec.dense_masks = dense
        for p_ in dec.parameters():
            p_.grad = None
        toks, ms, mf = _decoder_inputs(995)
        mf = mf.contiguous(memory_format=torch.channels_last).requires_grad_()
        out = dec(ms, mf)
        assert (out["pred_masks"] is None) == (not dense)
        crit.rand = C.ReplayRand(31)
        losses = crit(out, targets)
        total = sum(v * crit.weight_dict[k] for k, v in losses.items())
        total.backward()
        res.append(({k: v.detach().clone() for k, v in losses.items()}, {k: p_.grad.clone() for k, p_ in dec.named_parameters() if p_.grad is not None},
                    mf.grad.clone()))
    (l1, g1, m1), (l2, g2, m2) = res
    assert set(l1) == set(l2) and set(g1) == set(g2)
    for k in l1:
        torch.testing.assert_close(l2[k], l1[k], rtol=1e-4, atol=1e-5, msg=lambda m: f"{k}: {m}")
    for k in g1:
        scale = g1[k].abs().max().clamp_min(1e-6)
        assert ((g2[k] - g1[k]).abs().max() / scale).item() < 2e-3, k
    assert ((m2 - m1).abs().max() / m1.abs().max()).item() < 2e-3


@pytest.mark.parametrize("B,C,h,w", [(2, 256, 16, 24), (1, 8, 1, 1), (1, 4, 5, 3)])
def test_upsample_add_fwd_bwd_vs_torch(B, C, h, w):
    from partdistillation_amd.functions.rowwise import upsample_add, upsample_add_supported
    lo = _r((B, C, h, w), 71).contiguous(memory_format=torch.channels_last).requires_grad_()
    cur = _r((B, C, 2 * h, 2 * w), 72).contiguous(memory_format=torch.channels_last).requires_grad_()
    assert upsample_add_supported(lo, cur)
    y = upsample_add(lo, cur)
    gy = _r(y.shape, 73)
    y.backward(gy)
    g_lo, g_cur = lo.grad.clone(), cur.grad.clone()
    lo.grad = cur.grad = None
    ref = cur + F.interpolate(lo, size=cur.shape[-2:], mode="bilinear", align_corners=False)
    ref.backward(gy)
    torch.testing.assert_close(y, ref, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(g_lo, lo.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(g_cur, cur.grad, rtol=0, atol=0)
    # lo as one level of a [B, tokens, C] tensor (the pixel decoder's case): read in place through its batch stride, same result
    if C % 4 == 0:
        tokens = torch.zeros((B, 3 * h * w + 5 * 4, C), device=DEV)
        tokens[:, 8:8 + h * w] = lo.detach().permute(0, 2, 3, 1).reshape(B, h * w, C)
        lo_view = tokens[:, 8:8 + h * w].transpose(1, 2).reshape(B, C, h, w)
        assert not lo_view.is_contiguous(memory_format=torch.channels_last) or B == 1
        assert torch.equal(upsample_add(lo_view, cur.detach()), y.detach())


@pytest.mark.parametrize("N,C,H,W,P", [(80, 1, 64, 64, 1000), (1, 4, 96, 128, 5000), (3, 2, 7, 5, 33), (2, 1, 1, 1, 4), (5, 1, 256, 300, 20000),
                                       (2, 1, 113, 225, 3000), (1, 1, 600, 40, 999)])
def test_point_sample_planar_vs_grid_sample(N, C, H, W, P):
    """pd_point_sample_planar_f32 / _bwd_f32 against F.grid_sample (bilinear, zeros, align_corners=False): points on and beyond the
    borders, per-map points, the gradient with respect to the maps."""
    import torch.nn.functional as F
    from partdistillation_amd.functions import rowwise as rw
    g = torch.Generator(device="cuda").manual_seed(N * 100 + P)
    x = torch.randn(N, C, H, W, device="cuda", generator=g, requires_grad=True)
    coords = torch.rand(N, P, 2, device="cuda", generator=g) * 1.2 - 0.1          # some outside [0, 1]
    coords[:, :4] = torch.tensor([[0.0, 0.0], [1.0, 1.0], [0.5, 0.0], [1.0, 0.5]], device="cuda")
    assert rw.point_sample_planar_supported(x, coords)
    y = rw.point_sample_planar(x, coords)
    ref = F.grid_sample(x, 2.0 * coords.unsqueeze(2) - 1.0, mode="bilinear", padding_mode="zeros", align_corners=False).squeeze(3)
    assert y.shape == ref.shape == (N, C, P)
    torch.testing.assert_close(y, ref, rtol=1e-5, atol=1e-5)
    go = torch.randn(N, C, P, device="cuda", generator=g)
    (gx,) = torch.autograd.grad(y, x, go)
    (rx,) = torch.autograd.grad(ref, x, go)
    torch.testing.assert_close(gx, rx, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(20, 100, 12544), (3, 7, 130), (1, 1, 1), (4, 5, 0)])
def test_matcher_point_terms_match_torch(dtype, shape):
    """pd_matcher_point_terms: the matcher's .float() / softplus / sigmoid / row sums in one pass (reference matcher.py:108-158),
    over the logit range softplus switches branches in (|x| up to 40)."""
    from partdistillation_amd.functions import rowwise as rw
    torch.manual_seed(sum(shape))
    x = (torch.randn(shape, device="cuda") * 12).to(dtype)
    if x.numel() > 4:
        x.view(-1)[:4] = torch.tensor([0.0, 20.0, 20.5, -45.0], device="cuda").to(dtype)
    xf, sg, sp_sum, sg_sum = rw.matcher_point_terms(x)
    ref = x.float()
    assert torch.equal(xf, ref)
    torch.testing.assert_close(sg, ref.sigmoid(), rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(sp_sum, torch.nn.functional.softplus(ref.double()).sum(-1).float(), rtol=2e-6, atol=1e-5)
    torch.testing.assert_close(sg_sum, ref.double().sigmoid().sum(-1).float(), rtol=2e-6, atol=1e-5)


def test_batched_transpose_of_strided_weight_stacks():
    """pd_transpose_batched_f32: several [batch, rows, cols] -> [batch, cols, rows] problems in one launch, sources strided like layer
    weights in a flat parameter buffer (batch stride > rows * cols, row stride > cols), ragged 32 x 32 tiles"""
    import ctypes
    from partdistillation_amd import lib as L
    from partdistillation_amd.functions.encoder_core import _TrProblem
    flat = _r((3 * 70000,), 11)
    shapes = [(3, 70000, 300, 100, 300), (2, 40000, 40, 33, 37), (1, 5, 5, 1, 5), (3, 1024 * 20, 1024, 20, 1024)]   # batch, batch stride, row stride, rows, cols
    descs = (_TrProblem * len(shapes))()
    outs = []
    for d, (b, bs, rs, rows, cols) in zip(descs, shapes):
        out = torch.empty((b, cols, rows), device=DEV)
        d.src, d.dst, d.src_batch_stride, d.src_row_stride, d.batch, d.rows, d.cols = flat.data_ptr() + 4 * 16, out.data_ptr(), bs, rs, b, rows, cols
        outs.append(out)
    L.check(L.load().pd_transpose_batched_f32(ctypes.byref(descs), len(shapes), L.current_stream()))
    for out, (b, bs, rs, rows, cols) in zip(outs, shapes):
        src = torch.as_strided(flat, (b, rows, cols), (bs, rs, 1), 16)
        assert torch.equal(out, src.transpose(1, 2))


@pytest.mark.parametrize("B,C,H,W,sizes", [(2, 256, 64, 64, [(8, 8), (16, 16), (32, 32)]), (1, 8, 13, 9, [(5, 7), (26, 18), (13, 9), (1, 1)]),
                                           (2, 64, 32, 48, [(12, 20)])])
def test_multi_size_bilinear_resize_equals_interpolate(B, C, H, W, sizes):
    """pd_resize_bilinear_nhwc_f32: every pooled copy of the mask features by one launch (reference mask2former_transformer_decoder.py:452:
    F.interpolate(..., mode="bilinear", align_corners=False) per level), rows [B, h w, C]; fp32 equal to ATen's to rounding, bf16 = its cast"""
    from partdistillation_amd.functions.rowwise import resize_bilinear_rows, resize_bilinear_rows_supported
    x = _r((B, C, H, W), 5).contiguous(memory_format=torch.channels_last)
    assert resize_bilinear_rows_supported(x, sizes)
    got = resize_bilinear_rows(x, sizes)
    got16 = resize_bilinear_rows(x, sizes, torch.bfloat16)
    for (h, w), g, g16 in zip(sizes, got, got16):
        ref = F.interpolate(x, size=(h, w), mode="bilinear", align_corners=False).flatten(2).transpose(1, 2)     # [B, hw, C]
        torch.testing.assert_close(g, ref, rtol=1e-6, atol=1e-6)
        assert torch.equal(g16, g.to(torch.bfloat16))


def test_gradient_gather_mixed_dtypes_and_sum_of_squares():
    """pd_multi_gather_sumsq through functions/fused.GatherPlan: bf16 / fp32 / missing gradients of ragged sizes gathered into the flat fp32
    buffer (16-byte lanes where source and destination are aligned, scalar tails, zeros for a missing gradient) + the global sum of squares;
    a second gather with unchanged addresses (the table upload is skipped) gives the same"""
    from partdistillation_amd.functions.fused import GatherPlan
    numels = [16384 * 2 + 40, 7, 100000, 24, 16384, 3]
    offs, tot = [], 0
    for n in numels:
        offs.append(tot)
        tot += (n + 7) // 8 * 8
    plan = GatherPlan(numels, offs, torch.device(DEV))
    g = torch.Generator(device=DEV).manual_seed(5)
    grads = [torch.randn(numels[0], device=DEV, generator=g).to(torch.bfloat16), torch.randn(numels[1], device=DEV, generator=g).to(torch.bfloat16),
             torch.randn(numels[2], device=DEV, generator=g), None, torch.randn(numels[4] + 1, device=DEV, generator=g).to(torch.bfloat16)[1:],
             torch.randn(numels[5], device=DEV, generator=g)]
    for _ in range(2):
        flat = torch.full((tot,), 7.0, device=DEV)
        ss = torch.zeros(1, dtype=torch.float64, device=DEV)
        plan.upload(grads)
        plan.gather(flat, ss)
        want = 0.0
        for gr, n, o in zip(grads, numels, offs):
            ref = torch.zeros(n, device=DEV) if gr is None else gr.float()
            assert torch.equal(flat[o:o + n], ref)
            want += float(ref.double().pow(2).sum())
        assert abs(float(ss) - want) <= 1e-6 * want


def test_add_rows_amax_one_pass_equals_the_separate_ops():
    """pd_add_rows_amax_f32: q = a + b, the optional copy of a and both row maxima equal the ATen add / copy / abs-max (bit for bit)."""
    from partdistillation_amd.functions import rowwise as rw
    g = torch.Generator(device="cuda").manual_seed(3)
    for rows, cols in ((43, 256), (1000, 64), (7, 1024)):
        a = torch.randn(rows, cols, device="cuda", generator=g) * torch.logspace(-3, 3, rows, device="cuda")[:, None]
        b = torch.randn(rows, cols, device="cuda", generator=g)
        for cp in (False, True):
            q, ac, am, qm = rw.add_rows_amax(a, b, copy_a=cp)
            assert torch.equal(q, a + b) and torch.equal(am, a.abs().amax(1)) and torch.equal(qm, (a + b).abs().amax(1))
            assert (ac is None) == (not cp) and (ac is None or (torch.equal(ac, a) and ac.data_ptr() != a.data_ptr()))
