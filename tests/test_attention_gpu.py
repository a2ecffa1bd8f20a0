"""GPU tests of the split-key masked attention kernels (pd_attention.h) against an fp64 torch reference."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, k, v, mask, H):
    Lq, B, C = q.shape
    Lk, d = k.shape[0], C // H
    qh = q.double().reshape(Lq, B, H, d).permute(1, 2, 0, 3)
    kh = k.double().reshape(Lk, B, H, d).permute(1, 2, 0, 3)
    vh = v.double().reshape(Lk, B, H, d).permute(1, 2, 0, 3)
    s = qh @ kh.transpose(-1, -2) * d ** -0.5
    if mask is not None:
        s = s.masked_fill(mask[:, None], float("-inf"))
    o = torch.softmax(s, -1) @ vh
    return o.permute(2, 0, 1, 3).reshape(Lq, B, C)


@pytest.mark.parametrize("Lq,Lk,B,H,masked", [(100, 16384, 2, 8, True), (100, 4096, 2, 8, True), (100, 1024, 2, 8, True),
                                              (100, 100, 2, 8, False), (200, 1600, 1, 8, True), (7, 70, 3, 2, True), (130, 257, 1, 4, False),
                                              (128, 513, 2, 8, True), (33, 31, 1, 1, True), (100, 100, 2, 8, True)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_masked_attention_fwd_bwd(Lq, Lk, B, H, masked, dtype):
    from partdistillation_amd.functions.attention import masked_attention_d32
    g = torch.Generator(device="cuda").manual_seed(Lq + Lk)
    C = H * 32
    q = torch.randn(Lq, B, C, device="cuda", generator=g).to(dtype).requires_grad_()
    k = torch.randn(Lk, B, C, device="cuda", generator=g).to(dtype).requires_grad_()
    v = torch.randn(Lk, B, C, device="cuda", generator=g).to(dtype).requires_grad_()
    mask = None
    if masked:
        mask = torch.rand(B, Lq, Lk, device="cuda", generator=g) < 0.7
        mask[:, :, 0] = False                                       # every row keeps one key (the decoder guarantees it)
    go = torch.randn(Lq, B, C, device="cuda", generator=g).to(dtype)
    o = masked_attention_d32(q, k, v, mask, H)
    dq, dk, dv = torch.autograd.grad(o, (q, k, v), go)
    qr, kr, vr = [t.detach().double().requires_grad_() for t in (q, k, v)]
    ro = _ref(qr, kr, vr, mask, H)
    rq, rk, rv = torch.autograd.grad(ro, (qr, kr, vr), go.double())
    tol = dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(o.double(), ro, **tol)
    torch.testing.assert_close(dq.double(), rq, **tol)
    torch.testing.assert_close(dk.double(), rk, **tol)
    torch.testing.assert_close(dv.double(), rv, **tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fully_blocked_rows_give_zero(dtype):
    from partdistillation_amd.functions.attention import masked_attention_d32
    q = torch.randn(5, 1, 64, device="cuda").to(dtype); k = torch.randn(300, 1, 64, device="cuda").to(dtype)
    v = torch.randn(300, 1, 64, device="cuda").to(dtype)
    mask = torch.zeros(1, 5, 300, dtype=torch.bool, device="cuda")
    mask[0, 2] = True
    q.requires_grad_()
    o = masked_attention_d32(q, k, v, mask, 2)
    assert torch.isfinite(o).all() and o[2].abs().sum() == 0
    o.sum().backward()
    assert torch.isfinite(q.grad).all() and q.grad[2].abs().sum() == 0


@pytest.mark.parametrize("Lq,Lk,B,H,masked,cols,at", [(100, 16384, 2, 8, True, 768, 256), (100, 1024, 2, 8, True, 768, 512), (7, 97, 1, 2, False, 192, 64),
                                                       (128, 300, 3, 4, True, 256, 128)])
def test_strided_keys_and_values_equal_their_dense_copies(Lq, Lk, B, H, masked, cols, at):
    """pd_attn_fwd_d32_ld / pd_attn_bwd_d32_ld: k and v as column slices [at, at + H * 32) of wider [Lk * B, cols] matrices (the decoder's
    key / value projections of the layers that share a memory level come out of one product) — forward output, lse and all three gradients
    must be BIT-identical to the calls on dense copies of the slices."""
    from partdistillation_amd.functions.attention import attn_bwd_raw, attn_fwd_raw
    torch.manual_seed(Lq * 7 + Lk)
    C = H * 32
    q = torch.randn(Lq * B, C, device="cuda").bfloat16()
    kw, vw = torch.randn(Lk * B, cols, device="cuda").bfloat16(), torch.randn(Lk * B, cols, device="cuda").bfloat16()
    ks, vs = kw[:, at:at + C], vw[:, at:at + C]
    kd, vd = ks.contiguous(), vs.contiguous()
    m8 = None
    if masked:
        m8 = (torch.rand(B, Lq, Lk, device="cuda") < 0.4).to(torch.uint8)
        m8[:, :, 0] = 0
    scale = 32 ** -0.5
    o1, l1 = attn_fwd_raw(q, ks, vs, m8, B, H, scale)
    o2, l2 = attn_fwd_raw(q, kd, vd, m8, B, H, scale)
    assert torch.equal(o1, o2) and torch.equal(l1, l2)
    d_o = torch.randn_like(o1)
    g1 = attn_bwd_raw(q, ks, vs, m8, o1, d_o, l1, B, H, scale)
    g2 = attn_bwd_raw(q, kd, vd, m8, o2, d_o, l2, B, H, scale)
    for a, b in zip(g1, g2):
        assert a.is_contiguous() and torch.equal(a, b)


def test_copy_segments_concatenates_slices_in_one_launch():
    from partdistillation_amd.functions import rowwise as rw
    torch.manual_seed(3)
    srcs = [torch.randn(3 * 256, 256, device="cuda").bfloat16() for _ in range(3)] + [torch.randn(768, device="cuda").bfloat16() for _ in range(3)]
    wcat, bcat = torch.empty(768, 256, device="cuda", dtype=torch.bfloat16), torch.empty(768, device="cuda", dtype=torch.bfloat16)
    odd = torch.arange(37, device="cuda", dtype=torch.uint8)
    odd_dst = torch.zeros(40, device="cuda", dtype=torch.uint8)
    pairs = [(wcat[j * 256:(j + 1) * 256], srcs[j][256:512]) for j in range(3)] + [(bcat[j * 256:(j + 1) * 256], srcs[3 + j][256:512]) for j in range(3)]
    pairs.append((odd_dst[3:40], odd))                                   # unaligned destination, odd byte count
    rw.copy_segments(pairs)
    assert torch.equal(wcat, torch.cat([s_[256:512] for s_ in srcs[:3]])) and torch.equal(bcat, torch.cat([s_[256:512] for s_ in srcs[3:]]))
    assert torch.equal(odd_dst[3:], odd) and int(odd_dst[:3].sum()) == 0
