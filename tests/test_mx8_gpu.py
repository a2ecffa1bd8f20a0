"""pd_mx8_* (include/pd_mx8.h, csrc/mx8.hip) through the C-ABI against the plain-PyTorch restatement of the format (oracle/mx8_ref.py):
quantisation BIT-EXACT (elements and E8M0 scale bytes), the GEMM against the dequantised operands' product (fp32 accumulation in
another order + ONE rounding to bf16: asserted as 2^-7 of the tensor's maximum), every tile / stage schedule forced, ragged row
counts, e5m2 gradients, the fused bias / GELU / GELU' epilogues and the quantised output copy (bit-exact given the bf16 output)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mx():
    from partdistillation_amd import lib
    lib.load()
    from partdistillation_amd.functions import mx8
    return mx8


def _ref():
    from oracle import mx8_ref
    return mx8_ref


def _rand(shape, seed, spread=2.0):
    """bf16 values whose magnitude varies by rows / blocks over many binades (what a block-scaled format is for)"""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g) * torch.exp(spread * torch.randn(shape[0], 1, generator=g))
    return x.to(torch.bfloat16).to(DEV)


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("rows,cols", [(1, 32), (7, 96), (300, 768), (129, 3072)])
def test_quantize_bit_exact(fmt, rows, cols):
    mx, R = _mx(), _ref()
    x = _rand((rows, cols), 1 + rows, 6.0)
    x[0, :32] = 0                                                               # an all-zero block
    if rows > 2:
        x[1, 0], x[2, 1] = 448.0, 449.0                                         # the format maximum's edge
        x[2, 40] = float("nan")                                                 # a NaN stays a NaN element without changing its block's scale
    q, s = mx.quantize(x, fmt)
    rq, rs = R.quantize(x.cpu(), fmt)
    assert torch.equal(s.cpu(), rs)
    a, b = q.cpu(), rq
    nan = torch.isnan(x.float().cpu())
    assert torch.equal(a[~nan], b[~nan])
    if nan.any():
        assert torch.isnan(R.dequantize(a, s.cpu(), fmt)[nan]).all()


def test_quantize_strided_rows_and_grouped():
    mx, R = _mx(), _ref()
    big = _rand((50, 512), 3)
    view = big[:, 128:128 + 256]                                                # row stride 512, 256 columns
    q, s = mx.quantize(view, 0)
    rq, rs = R.quantize(view.cpu(), 0)
    assert torch.equal(q.cpu(), rq) and torch.equal(s.cpu(), rs)
    ws = [_rand((64, 128), 4), _rand((192, 64), 5), _rand((32, 32), 6), _rand((128, 768), 7)]
    for fmt in (0, 1):
        outs = mx.quantize_grouped(ws, fmt)
        for w, (gq, gs) in zip(ws, outs):
            rq, rs = R.quantize(w.cpu(), fmt)
            assert gq.shape == w.shape and gs.shape == (w.shape[0], w.shape[1] // 32)
            assert torch.equal(gq.cpu(), rq) and torch.equal(gs.cpu(), rs)


def _close(got, ref, tol=2.0 ** -7):
    err = (got.float().cpu() - ref).abs().max().item()
    assert err <= tol * max(ref.abs().max().item(), 1e-6), (err, ref.abs().max().item())


SHAPES = [(128, 64, 128), (300, 128, 256), (1000, 768, 768), (257, 2304, 768), (640, 768, 3072), (33, 64, 1024)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("bn,nst", [(0, 0), (64, 1), (64, 2), (128, 1), (128, 2)])
def test_gemm_vs_dequantised_product(M, N, K, bn, nst):
    from partdistillation_amd import lib
    mx, R = _mx(), _ref()
    L = lib.load()
    x, w = _rand((M, K), 11 + M), _rand((N, K), 12 + N, 1.0)
    a, wq = mx.quantize(x, 0), mx.quantize(w, 0)
    L.pd_debug_set(b"mx_bn", bn); L.pd_debug_set(b"mx_nst", nst)
    try:
        y = mx.linear(a, wq)
    finally:
        L.pd_debug_set(b"mx_bn", 0); L.pd_debug_set(b"mx_nst", 0)
    ref = R.gemm((a[0].cpu(), a[1].cpu()), (wq[0].cpu(), wq[1].cpu()))
    assert y.shape == ref.shape and y.dtype == torch.bfloat16
    _close(y, ref)
    # and the format itself is a faithful operand: the MX product is within fp8's element precision of the bf16 operands' product
    exact = x.float().cpu().double() @ w.float().cpu().double().t()
    scale = (x.float().cpu().norm(dim=1, keepdim=True) * w.float().cpu().norm(dim=1)[None]).double()
    assert ((ref.double() - exact).abs() <= 0.14 * scale + 1e-30).all()


@pytest.mark.parametrize("a_fmt", [0, 1])
@pytest.mark.parametrize("bias_dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogues(a_fmt, bias_dtype):
    mx, R = _mx(), _ref()
    M, N, K = 500, 256, 384
    x, w = _rand((M, K), 21), _rand((N, K), 22, 1.0)
    bias = torch.randn(N, device=DEV).to(bias_dtype)
    a, wq = mx.quantize(x, a_fmt), mx.quantize(w, 0)
    acpu, wcpu = (a[0].cpu(), a[1].cpu()), (wq[0].cpu(), wq[1].cpu())
    base = R.gemm(acpu, wcpu, a_fmt, bias.float().cpu())
    # bias + GELU, the pre-activation kept, the result again as MX e4m3
    y, pre, (oq, osc) = mx.linear(a, wq, bias, act=mx.ACT_GELU, want_pre=True, a_fmt=a_fmt, out_mx=0)
    _close(pre, base)
    _close(y, F.gelu(base))
    rq, rs = R.quantize(y.cpu(), 0)                                             # the quantised copy describes the bf16 output exactly
    assert torch.equal(oq.cpu(), rq) and torch.equal(osc.cpu(), rs)
    # GELU' gate (the input gradient of fc2 feeding fc1's), quantised e5m2
    h = _rand((M, N), 23, 0.3)
    hf = h.float().cpu()
    gp = 0.5 * (1 + torch.erf(hf / math.sqrt(2))) + hf * torch.exp(-0.5 * hf * hf) / math.sqrt(2 * math.pi)
    y2, (oq2, os2) = mx.linear(a, wq, None, gate=h, gate_mode=mx.GATE_GELU, a_fmt=a_fmt, out_mx=1)
    _close(y2, R.gemm(acpu, wcpu, a_fmt) * gp)
    rq2, rs2 = R.quantize(y2.cpu(), 1)
    assert torch.equal(oq2.cpu(), rq2) and torch.equal(os2.cpu(), rs2)


def test_chained_layers_through_the_quantised_output():
    """fc1 (+ GELU, quantised output) -> fc2 reading that output: equals quantising fc1's bf16 output in a pass of its own"""
    mx = _mx()
    M, C = 777, 256
    x, w1, w2 = _rand((M, C), 31), _rand((4 * C, C), 32, 0.5), _rand((C, 4 * C), 33, 0.5)
    a, q1, q2 = mx.quantize(x, 0), mx.quantize(w1, 0), mx.quantize(w2, 0)
    y, aq = mx.linear(a, q1, act=mx.ACT_GELU, out_mx=0)
    z = mx.linear(aq, q2)
    z2 = mx.linear(mx.quantize(y, 0), q2)
    assert torch.equal(z, z2)


def test_unsupported_shapes_raise():
    from partdistillation_amd import lib
    mx = _mx()
    assert mx.supported(10, 64, 128) and not mx.supported(10, 96, 128) and not mx.supported(10, 64, 192)
    x, w = _rand((8, 192), 41), _rand((64, 192), 42)
    with pytest.raises(lib.PdHipError):
        mx.linear(mx.quantize(x), mx.quantize(w))
