"""GPU tests of the fused Swin window-attention kernels (include/pd_window_attention.h) against a plain torch fp32
restatement of the reference's WindowAttention.forward body (modeling/backbone/swin.py:146-173), the bias gathered with
the reference's relative_position_index (:110-125) and the mask built the reference's way (:425-441).
Tolerances: operands are bf16 (8-bit mantissa), scores / softmax fp32 — 2e-2 of each tensor's scale."""
import pytest
import torch

pytestmark = pytest.mark.gpu
WS, N = 12, 144


def _relative_position_index():
    ch, cw = torch.meshgrid(torch.arange(WS), torch.arange(WS), indexing="ij")
    coords = torch.stack([ch.reshape(-1), cw.reshape(-1)])
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += WS - 1
    rel[:, :, 1] += WS - 1
    rel[:, :, 0] *= 2 * WS - 1
    return rel.sum(-1)


def _reference_mask(H, W, shift):
    Hp, Wp = -(-H // WS) * WS, -(-W // WS) * WS
    img = torch.zeros((1, Hp, Wp, 1))
    cnt = 0
    for hs in (slice(0, -WS), slice(-WS, -shift), slice(-shift, None)):
        for wsl in (slice(0, -WS), slice(-WS, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = img.view(1, Hp // WS, WS, Wp // WS, WS, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, N)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)


def _ref(qkv, table, mask, heads, scale):
    B_, _, C3 = qkv.shape
    C = C3 // 3
    q, k, v = qkv.reshape(B_, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    attn = (q * scale) @ k.transpose(-2, -1)
    bias = table[_relative_position_index().view(-1).to(table.device)].view(N, N, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.view(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, N, N)
    return (attn.softmax(-1) @ v).transpose(1, 2).reshape(B_, N, C)


def _close(got, want, frac, what):
    err = (got.double() - want.double()).abs().max().item()
    scale = want.double().abs().max().item()
    assert err <= frac * scale, f"{what}: max abs err {err:.3e} > {frac} * {scale:.3e}"


@pytest.mark.parametrize("heads,images,H,W,shift", [(3, 2, 30, 30, 6), (3, 2, 30, 30, 0), (6, 1, 24, 36, 6), (4, 1, 12, 12, 0),
                                                     (2, 3, 25, 40, 6), (4, 456, 30, 30, 6), (32, 1, 36, 36, 6)])
def test_window_attention_fwd_bwd(heads, images, H, W, shift):
    from partdistillation_amd.functions import window_attention as wa
    g = torch.Generator(device="cuda").manual_seed(heads * 1000 + H)
    nW = (-(-H // WS)) * (-(-W // WS))
    B_, C = images * nW, heads * 32
    qkv = torch.randn(B_, N, 3 * C, device="cuda", generator=g).to(torch.bfloat16).requires_grad_()
    table = (torch.randn(529, heads, device="cuda", generator=g) * 0.5).requires_grad_()
    go = torch.randn(B_, N, C, device="cuda", generator=g).to(torch.bfloat16)
    scale = 32 ** -0.5
    regions = wa.shifted_window_regions(H, W, shift, "cuda") if shift else None
    mask = _reference_mask(H, W, shift).cuda() if shift else None
    if shift:
        assert regions[0].shape == (nW, N) and int(regions[1].sum()) > 0
    out = wa.window_attention(qkv, table, regions, scale, nW)
    dqkv, dtable = torch.autograd.grad(out, (qkv, table), go)
    qr, tr = qkv.detach().float().requires_grad_(), table.detach().clone().requires_grad_()
    ro = _ref(qr, tr, mask, heads, scale)
    rq, rt = torch.autograd.grad(ro, (qr, tr), go.float())
    assert out.dtype == torch.bfloat16 and out.shape == (B_, N, C) and dtable.dtype == torch.float32
    _close(out, ro, 2e-2, "out")
    _close(dqkv, rq, 2e-2, "dqkv")
    _close(dtable, rt, 2e-2, "dtable")
    # the MX-fp8 copies of out / dqkv (include/pd_mx8.h: a (token, head) piece = one 32-element block): the same bf16 tensors, and their
    # quantisation bit-exact against oracle/mx8_ref.py
    from oracle import mx8_ref as MX
    q_, t_ = qkv.detach(), table.detach()
    out0, lse0 = wa.fwd_raw(q_, t_, regions, scale, nW)
    for fmt in (0, 1):
        out1, lse1, (oq, osc) = wa.fwd_raw(q_, t_, regions, scale, nW, mx=fmt)
        assert torch.equal(out1, out0) and torch.equal(lse1, lse0)
        rq8, rs8 = MX.quantize(out1.view(-1, C).cpu(), fmt)
        assert torch.equal(oq.cpu(), rq8) and torch.equal(osc.cpu(), rs8)
        d0, dt0 = wa.bwd_raw(q_, t_, regions, out0, go, lse0, scale, nW)
        d1, dt1, (dq8, ds8) = wa.bwd_raw(q_, t_, regions, out0, go, lse0, scale, nW, mx=fmt)
        assert torch.equal(d1, d0)
        rq8, rs8 = MX.quantize(d1.view(-1, 3 * C).cpu(), fmt)
        assert torch.equal(dq8.cpu(), rq8) and torch.equal(ds8.cpu(), rs8)


def test_window_attention_rejects_bad_arguments():
    from partdistillation_amd import lib
    from partdistillation_amd.functions import window_attention as wa
    qkv = torch.zeros(5, N, 96, device="cuda", dtype=torch.bfloat16)
    table = torch.zeros(529, 1, device="cuda")
    with pytest.raises(lib.PdHipError, match="multiple of nW"):
        wa.fwd_raw(qkv, table, None, 1.0, 4)
    with pytest.raises(RuntimeError, match="GPU only"):
        wa.window_attention(qkv.cpu(), table.cpu(), None, 1.0, 1)
    out, lse = wa.fwd_raw(qkv[:0], table, None, 1.0, 1)
    assert out.shape == (0, N, 32)


def test_swin_block_fused_matches_library_attention_path():
    """one shifted + one plain Swin block under bf16 autocast: fused window attention == the torch SDPA path the same
    module takes when the kernel does not apply (same weights, same input), forward and parameter gradients"""
    from partdistillation_amd.functions import window_attention as wa
    from partdistillation_amd.modeling.backbone import swin
    torch.manual_seed(0)
    layer = swin.BasicLayer(dim=64, depth=2, num_heads=2, window_size=12, drop_path=0.0).cuda()
    for blk in layer.blocks:
        torch.nn.init.normal_(blk.attn.relative_position_bias_table, std=0.5)
    x = torch.randn(2, 30 * 26, 64, device="cuda")
    res = {}
    for fused in (True, False):
        orig = wa.supported
        if not fused:
            wa.supported = lambda *a, **k: False
        try:
            layer.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = layer(x, 30, 26)[0]
            (y.float() ** 2).mean().backward()
            res[fused] = (y.detach().float(), {k: p.grad.clone() for k, p in layer.named_parameters()})
        finally:
            wa.supported = orig
    _close(res[True][0], res[False][0], 2e-2, "block output")
    for k, gr in res[False][1].items():
        _close(res[True][1][k], gr, 5e-2, "grad " + k)
