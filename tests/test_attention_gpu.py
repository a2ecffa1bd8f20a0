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
