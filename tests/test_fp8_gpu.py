"""GPU tests of the fp8 operand kernels (include/pd_fp8.h) and the fp8 Linear built on them (BASELINE config 5).
Quantisation is checked BIT-EXACT against torch's own OCP e4m3fn / e5m2 casts of the scaled, clamped values; the Linear
against an fp32 torch reference at the tolerance the formats allow (e4m3: 3 mantissa bits -> 2^-4 per element, averaged
down by the K-long sums; stated per check)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("n,scale", [(8, 1.0), (4096, 3.0), (144 * 1536 * 5, 0.01), (16 * 1000 * 1000, 100.0)])
def test_quantize_matches_torch_cast(dtype, fmt, n, scale):
    from partdistillation_amd.functions import fp8
    g = torch.Generator(device="cuda").manual_seed(n)
    x = (torch.randn(n, device="cuda", generator=g) * scale).to(dtype)
    x[::7] = 0
    q, inv = fp8.quantize(x, fmt)
    fmax = 448.0 if fmt == 0 else 57344.0
    amax = x.float().abs().max()
    assert q.dtype == (torch.float8_e4m3fn if fmt == 0 else torch.float8_e5m2) and q.shape == x.shape
    torch.testing.assert_close(inv, amax / fmax, rtol=1e-6, atol=0)
    scale = torch.full((), fmax, device="cuda") / amax                                 # one correctly rounded fp32 division
    want = (x.float() * scale).clamp(-fmax, fmax).to(q.dtype)
    assert torch.equal(q.view(torch.uint8), want.view(torch.uint8))
    # round trip: relative error of a normal-range element is at most half an ulp (2^-4 e4m3, 2^-3 e5m2)
    back = q.float() * inv
    big = x.float().abs() > amax * 2e-2
    rel = ((back - x.float()).abs() / x.float().abs().clamp_min(1e-30))[big].max().item()
    assert rel <= (2 ** -4 if fmt == 0 else 2 ** -3) * 1.001


def test_quantize_edge_cases():
    from partdistillation_amd import lib
    from partdistillation_amd.functions import fp8
    z = torch.zeros(64, device="cuda")
    q, inv = fp8.quantize(z)
    assert float(inv) == 1.0 and int(q.view(torch.uint8).sum()) == 0                       # amax 0 -> scale 1
    e = torch.empty(0, device="cuda")
    q, inv = fp8.quantize(e)
    assert q.numel() == 0
    with pytest.raises(lib.PdHipError, match="multiple of 8"):
        fp8.quantize(torch.zeros(12, device="cuda"))
    with pytest.raises(RuntimeError, match="GPU only"):
        fp8.quantize(torch.zeros(16))
    x = torch.tensor([1.0, -2.0, float("inf"), 3.0, 0, 0, 0, 0], device="cuda")            # inf saturates, scale from inf -> 0
    q, inv = fp8.quantize(x)
    assert torch.isfinite(q.float()[:2]).all()


@pytest.mark.parametrize("M,N,K,bias", [(288, 768, 384, True), (1440, 1536, 1536, True), (144 * 8, 384, 1536, False)])
def test_fp8_linear_fwd_bwd(M, N, K, bias):
    from partdistillation_amd.functions import fp8
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn(2, M // 2, K, device="cuda", generator=g).to(torch.bfloat16).requires_grad_()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).requires_grad_()
    b = torch.randn(N, device="cuda", generator=g).requires_grad_() if bias else None
    go = torch.randn(2, M // 2, N, device="cuda", generator=g).to(torch.bfloat16)
    assert fp8.supported(x, w, 384)
    y = fp8.linear(x, w, b)
    grads = torch.autograd.grad(y, (x, w) + ((b,) if bias else ()), go)
    xr, wr = x.detach().float().requires_grad_(), w.detach().clone().requires_grad_()
    br = b.detach().clone().requires_grad_() if bias else None
    yr = torch.nn.functional.linear(xr, wr, br)
    gr = torch.autograd.grad(yr, (xr, wr) + ((br,) if bias else ()), go.float())

    def close(a, r, frac, what):
        err = (a.float() - r).abs().max().item()
        assert err <= frac * r.abs().max().item(), f"{what}: {err:.3e} vs scale {r.abs().max().item():.3e}"
    assert y.dtype == torch.bfloat16
    close(y, yr, 4e-2, "y")            # e4m3 x e4m3
    close(grads[0], gr[0], 8e-2, "dx")  # e5m2 (2 mantissa bits) x e4m3
    close(grads[1], gr[1], 2e-2, "dw")  # bf16 GEMM
    if bias:
        close(grads[2], gr[2], 2e-2, "db")


def test_swin_block_fp8_close_to_bf16():
    """the Swin layer with MODEL.SWIN.FP8_GEMM semantics (FP8 dict switched on) against its own bf16 run"""
    from partdistillation_amd.modeling.backbone import swin
    torch.manual_seed(0)
    layer = swin.BasicLayer(dim=384, depth=2, num_heads=12, window_size=12, drop_path=0.0).cuda()
    x = torch.randn(2, 24 * 24, 384, device="cuda")
    out = {}
    try:
        for on in (False, True):
            swin.FP8["enabled"], swin.FP8["min_k"] = on, 384
            layer.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = layer(x, 24, 24)[0]
            (y.float() ** 2).mean().backward()
            out[on] = (y.detach().float(), layer.blocks[1].mlp.fc1.weight.grad.clone())
    finally:
        swin.FP8["enabled"] = False
    assert not torch.equal(out[True][0], out[False][0])                                   # the fp8 path really ran
    for i, frac in ((0, 5e-2), (1, 1.5e-1)):
        err = (out[True][i] - out[False][i]).abs().max().item()
        assert err <= frac * out[False][i].abs().max().item(), (i, err)
