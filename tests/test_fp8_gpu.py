"""GPU tests of the fp8 Linear of BASELINE config 5 (functions/fp8.py on the MX-fp8 kernels of include/pd_mx8.h; the kernels
themselves: tests/test_mx8_gpu.py) against an fp32 torch reference at the tolerance the formats allow (e4m3: 3 mantissa bits ->
2^-4 per element, e5m2: 2^-3, averaged down by the K-long sums; stated per check), and of a Swin stage with MODEL.SWIN.FP8_GEMM
semantics — module by module (fp32 weights) and fused (bf16 weights) — against its own bf16 run."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,bias", [(288, 768, 384, True), (1440, 1536, 1536, True), (144 * 8, 384, 1536, False), (250, 128, 128, True)])
def test_fp8_linear_fwd_bwd(M, N, K, bias, wdtype):
    from partdistillation_amd.functions import fp8
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn(2, M // 2, K, device="cuda", generator=g).to(torch.bfloat16).requires_grad_()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(wdtype).requires_grad_()
    b = torch.randn(N, device="cuda", generator=g).to(wdtype).requires_grad_() if bias else None
    go = torch.randn(2, M // 2, N, device="cuda", generator=g).to(torch.bfloat16)
    assert fp8.supported(x, w, 128)
    y = fp8.linear(x, w, b)
    grads = torch.autograd.grad(y, (x, w) + ((b,) if bias else ()), go)
    xr, wr = x.detach().float().requires_grad_(), w.detach().float().requires_grad_()
    br = b.detach().float().requires_grad_() if bias else None
    yr = torch.nn.functional.linear(xr, wr, br)
    gr = torch.autograd.grad(yr, (xr, wr) + ((br,) if bias else ()), go.float())

    def close(a, r, frac, what):
        err = (a.float() - r).abs().max().item()
        assert err <= frac * r.abs().max().item(), f"{what}: {err:.3e} vs scale {r.abs().max().item():.3e}"
    assert y.dtype == torch.bfloat16 and grads[1].dtype == wdtype
    close(y, yr, 4e-2, "y")            # e4m3 x e4m3
    close(grads[0], gr[0], 8e-2, "dx")  # e5m2 (2 mantissa bits) x e4m3
    close(grads[1], gr[1], 2e-2, "dw")  # bf16 GEMM
    if bias:
        close(grads[2], gr[2], 2e-2, "db")


def test_fp8_linear_shapes_it_declines():
    from partdistillation_amd.functions import fp8
    x = torch.zeros(4, 192, device="cuda", dtype=torch.bfloat16)
    assert not fp8.supported(x, torch.zeros(576, 192, device="cuda"), 128)              # K = 192 is not whole 128-byte steps
    assert not fp8.supported(torch.zeros(4, 256, device="cuda"), torch.zeros(64, 256, device="cuda"), 128)   # N = 64: the input gradient contracts over it
    assert not fp8.supported(torch.zeros(4, 256, device="cuda"), torch.zeros(128, 256, device="cuda"), 384)  # below FP8_MIN_K
    with pytest.raises(RuntimeError, match="GPU only"):
        fp8.linear(torch.zeros(4, 128), torch.zeros(128, 128), None)


@pytest.mark.parametrize("fused", [False, True])
def test_swin_stage_fp8_close_to_bf16(fused):
    """the Swin layer with MODEL.SWIN.FP8_GEMM semantics (FP8 dict switched on) against its own bf16 run: module by module with fp32
    weights (functions/fp8.py per Linear) and as the fused stage with bf16 weights (swin_core.py: quantisation in the GEMM epilogues)"""
    from partdistillation_amd.functions import mx8
    from partdistillation_amd.modeling.backbone import swin
    torch.manual_seed(0)
    layer = swin.BasicLayer(dim=384, depth=2, num_heads=12, window_size=12, drop_path=0.0).cuda()
    if fused:
        for blk in layer.blocks:
            for lin in (blk.attn.qkv, blk.attn.proj, blk.mlp.fc1, blk.mlp.fc2):
                lin.weight.data = lin.weight.data.to(torch.bfloat16)
                lin.bias.data = lin.bias.data.to(torch.bfloat16)
    x = torch.randn(2, 24 * 24, 384, device="cuda")
    out = {}
    calls = {"n": 0}
    f0 = mx8.linear
    mx8.linear = lambda *a, **k: (calls.__setitem__("n", calls["n"] + 1), f0(*a, **k))[1]
    try:
        for on in (False, True):
            swin.FP8["enabled"], swin.FP8["min_k"] = on, 384
            layer.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = layer(x, 24, 24)[0]
            (y.float() ** 2).mean().backward()
            out[on] = (y.detach().float(), layer.blocks[1].mlp.fc1.weight.grad.float().clone())
            assert calls["n"] == (16 if on else 0)                                       # 4 forward + 4 input-gradient GEMMs per block
    finally:
        swin.FP8["enabled"] = False
        mx8.linear = f0
    assert not torch.equal(out[True][0], out[False][0])
    for i, frac in ((0, 5e-2), (1, 1.5e-1)):
        err = (out[True][i] - out[False][i]).abs().max().item()
        assert err <= frac * out[False][i].abs().max().item(), (i, err)
