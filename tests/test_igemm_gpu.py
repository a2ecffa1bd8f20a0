"""pd_igemm_bf16 (include/pd_igemm.h, csrc/igemm_bf16.hip) through the C-ABI against plain PyTorch fp32 references of the same op:
the R50 bottleneck convolutions forward / input gradient with their fused epilogues (every geometry of the backbone), the Swin
Linear shapes with bias / GELU / GELU', ragged row counts, the split-K schedule and its run-to-run determinism.
Tolerance: operands are bf16 (exact products, fp32 accumulation), the result is rounded once to bf16 -> |err| <= 2^-8 |ref| + accumulation
noise; asserted as 1e-2 of the tensor's maximum."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ig():
    from partdistillation_amd import lib
    lib.load()
    from partdistillation_amd.functions import igemm
    return igemm


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).to(DEV)


def _close(got, ref, tol=1e-2):
    err = (got.float() - ref).abs().max().item()
    assert err <= tol * max(ref.abs().max().item(), 1e-6), (err, ref.abs().max().item())


# (ci, co, k, stride, H) — the bottleneck geometries of R50 at a reduced spatial size (incl. the strided 3 x 3 and 1 x 1 shortcuts)
R50 = [(64, 64, 1, 1, 24), (64, 64, 3, 1, 24), (64, 256, 1, 1, 24), (256, 64, 1, 1, 24), (256, 128, 1, 1, 24), (128, 128, 3, 2, 24),
       (256, 512, 1, 2, 24), (128, 512, 1, 1, 12), (512, 128, 1, 1, 12), (128, 128, 3, 1, 12), (512, 256, 1, 1, 12), (256, 256, 3, 2, 12),
       (512, 1024, 1, 2, 12), (256, 1024, 1, 1, 6), (1024, 256, 1, 1, 6), (256, 256, 3, 1, 6), (1024, 512, 1, 1, 6), (512, 512, 3, 2, 6),
       (1024, 2048, 1, 2, 6), (512, 2048, 1, 1, 3), (2048, 512, 1, 1, 3), (512, 512, 3, 1, 3)]


@pytest.mark.parametrize("ci,co,k,stride,H", R50)
def test_conv_forward_fused_epilogue_vs_torch(ci, co, k, stride, H):
    ig = _ig()
    B, W, pad = 2, H + 2, k // 2
    x = _rand((B, H, W, ci), 1)
    w = _rand((co, k, k, ci), 2, (k * k * ci) ** -0.5)
    scale = torch.rand(co, device=DEV) + 0.5
    bias = torch.randn(co, device=DEV)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = _rand((B, Ho, Wo, co), 3)
    y = ig.conv_nhwc(x, w, k=k, stride=stride, pad=pad, scale=scale, bias=bias, res=res, act=ig.ACT_RELU)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), None, stride, pad).permute(0, 2, 3, 1)
    ref = F.relu(ref * scale + bias + res.float())
    assert y.shape == ref.shape
    _close(y, ref)
    y2 = ig.conv_nhwc(x, w, k=k, stride=stride, pad=pad)                       # bare convolution
    _close(y2, F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), None, stride, pad).permute(0, 2, 3, 1))


@pytest.mark.parametrize("ci,co,k,stride,H", [g for g in R50 if not (g[2] == 1 and g[3] == 2)])
def test_conv_input_gradient_with_addend_and_relu_mask_vs_autograd(ci, co, k, stride, H):
    ig = _ig()
    B, W, pad = 2, H + 2, k // 2
    if stride == 2 and (H % 2 or W % 2):
        W += 1
    x = _rand((B, H, W, ci), 4).float().requires_grad_()
    w = _rand((co, k, k, ci), 5, (k * k * ci) ** -0.5)
    z = F.conv2d(x.permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), None, stride, pad)
    dz = _rand(tuple(z.permute(0, 2, 3, 1).shape), 6)
    (gx,) = torch.autograd.grad(z, x, dz.float().permute(0, 3, 1, 2))
    wt = w.permute(3, 1, 2, 0).contiguous()                                        # [ci][k][k][co]
    addend = _rand((B, H, W, ci), 7)
    below = _rand((B, H, W, ci), 8)                                               # the activation whose ReLU mask gates the gradient
    dx = ig.conv_dgrad_nhwc(dz, wt, (H, W), k=k, stride=stride, pad=pad, res=addend, gate=below, gate_mode=ig.GATE_RELU)
    ref = (gx + addend.float()) * (below.float() > 0)
    _close(dx, ref)
    _close(ig.conv_dgrad_nhwc(dz, wt, (H, W), k=k, stride=stride, pad=pad), gx)


def test_upsampled_addend_is_added_at_even_pixels_only():
    """the shortcut's stride-2 1 x 1 input gradient stays COMPACT ([B, H/2, W/2, C]) and enters the main branch's input gradient as
    an addend that exists at even (y, x) only"""
    ig = _ig()
    B, H, W, ci, co = 2, 12, 16, 256, 128
    dz = _rand((B, H, W, co), 9)
    w = _rand((co, 1, 1, ci), 10, ci ** -0.5)
    wt = w.permute(3, 1, 2, 0).contiguous()
    compact = _rand((B, H // 2, W // 2, ci), 11)
    dx = ig.conv_dgrad_nhwc(dz, wt, (H, W), k=1, res=compact, res_mode=ig.RES_UP2)
    ref = dz.float().reshape(-1, co) @ w.float().view(co, ci)
    up = torch.zeros((B, H, W, ci), device=DEV)
    up[:, ::2, ::2] = compact.float()
    _close(dx, ref.view(B, H, W, ci) + up)


@pytest.mark.parametrize("M,K,N", [(8192, 128, 384), (4096 + 77, 256, 768), (2048, 512, 2048), (2048, 2048, 512), (1000, 1024, 1024),
                                   (3200, 192, 576), (800, 1536, 6144), (800, 6144, 1536), (43008, 256, 512), (128, 4608, 512)])
def test_linear_bias_gelu_and_gelu_gate_vs_torch(M, K, N):
    ig = _ig()
    x, w = _rand((M, K), 12), _rand((N, K), 13, K ** -0.5)
    b = torch.randn(N, device=DEV)
    ref = x.float() @ w.float().t() + b
    _close(ig.linear(x, w, b), ref)
    a, h = ig.linear(x, w, b, act=ig.ACT_GELU, want_pre=True)
    _close(h, ref)
    _close(a, F.gelu(ref))
    hh = _rand((M, N), 14)
    got = ig.linear(x, w, None, gate=hh, gate_mode=ig.GATE_GELU)                   # dA = (dF W2) * gelu'(h)
    hf = hh.float().requires_grad_()
    (gp,) = torch.autograd.grad(F.gelu(hf).sum(), hf)
    _close(got, (x.float() @ w.float().t()) * gp)


def test_split_k_is_deterministic_and_leaves_its_tickets_zero():
    ig = _ig()
    x, w = _rand((2048, 4608), 15), _rand((512, 4608), 16, 4608 ** -0.5)           # 64 tiles -> split
    outs = [ig.linear(x, w) for _ in range(5)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    _close(outs[0], x.float() @ w.float().t())
    from partdistillation_amd import lib
    ws = ig._WS[(str(x.device), lib.current_stream())]
    torch.cuda.synchronize()
    assert int(ws[:4096 * 4].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize("knobs", [dict(ig_bn=128, ig_nst=1), dict(ig_bn=128, ig_nst=2), dict(ig_bn=128, ig_nst=3), dict(ig_bn=64, ig_nst=1),
                                   dict(ig_bn=64, ig_nst=2), dict(ig_bn=64, ig_nst=3), dict(ig_bn=64, ig_nst=3, ig_splits=3)])
def test_every_tile_and_pipeline_variant_gives_the_same_result(knobs):
    """the schedule (column-tile width, LDS stages: 1 = one stage and four workgroups per CU, 2 = one step of prefetch, 3 = ring with
    counted waits, split-K) is chosen per shape by the library; every variant forced through pd_debug_set must agree"""
    ig = _ig()
    from partdistillation_amd import lib
    L = lib.load()
    x, w = _rand((1000, 1152), 21), _rand((384, 1152), 22, 1152 ** -0.5)
    xc = _rand((2, 14, 18, 128), 23)
    wc = _rand((256, 3, 3, 128), 24, 1152 ** -0.5)
    try:
        for k, v in knobs.items():
            lib.check(L.pd_debug_set(k.encode(), v))
        y = ig.linear(x, w)
        yc = ig.conv_nhwc(xc, wc, k=3, stride=2, pad=1)
    finally:
        for k in knobs:
            L.pd_debug_set(k.encode(), 0)
    _close(y, x.float() @ w.float().t())
    _close(yc, F.conv2d(xc.float().permute(0, 3, 1, 2), wc.float().permute(0, 3, 1, 2), None, 2, 1).permute(0, 2, 3, 1))


@pytest.mark.parametrize("M,K,N,knobs", [(4000, 128, 384, {}), (10368, 512, 2048, {}), (1000 + 13, 192, 576, {}), (139392, 128, 128, {}), (2592, 1024, 3072, {}),
                                         (777, 64, 8, {}), (5000, 256, 256, dict(wg_nst=1, wg_splits=1)), (5000, 256, 256, dict(wg_nst=2, wg_splits=7)),
                                         (300, 1536, 384, dict(wg_nst=1, wg_splits=1))])
def test_linear_weight_gradient_vs_torch(M, K, N, knobs):
    """pd_wgrad_bf16: dW = dY^T X (+ bias gradient, + row scale) on strided row views, every schedule; deterministic from run to run"""
    ig = _ig()
    from partdistillation_amd import lib
    L = lib.load()
    dyf, xf = _rand((M, N + 8), 31), _rand((M, K + 16), 32)
    dy, x = dyf[:, :N], xf[:, 8:8 + K]                                            # row strides larger than the widths, 16-byte aligned bases
    rs = torch.rand(N, device=DEV) + 0.5
    db = torch.full((N,), 3.0, device=DEV)
    try:
        for k_, v in knobs.items():
            lib.check(L.pd_debug_set(k_.encode(), v))
        dw = ig.wgrad(dy, x, db=db, row_scale=rs)
        dw2 = ig.wgrad(dy, x)
        dw3 = ig.wgrad(dy, x)
        dw32 = ig.wgrad(dy, x, out_dtype=torch.float32)
    finally:
        for k_ in knobs:
            L.pd_debug_set(k_.encode(), 0)
    ref = dy.float().t() @ x.float()
    _close(dw2, ref, 5e-3)
    _close(dw, ref * rs[:, None], 5e-3)
    assert torch.equal(dw2, dw3)
    assert dw32.dtype == torch.float32 and torch.equal(dw32.to(torch.bfloat16), dw2)           # the same sums, before the rounding
    _close(dw32, ref, 2e-3)
    torch.testing.assert_close(db, 3.0 + dy.float().sum(0), rtol=2e-3, atol=2e-3 * float(dy.float().abs().sum(0).max()))


def test_weight_gradient_leaves_the_shared_workspace_tickets_alone():
    """pd_wgrad_bf16 and the split-K schedule of pd_igemm_bf16 share one workspace: the tickets at its start must still read zero"""
    ig = _ig()
    x, w = _rand((256, 4096), 5), _rand((128, 4096), 6)
    y0 = ig.linear(x, w)                                                         # 2 tiles, 64 K-steps: split-K (tickets in use)
    ig.wgrad(_rand((40000, 256), 7), _rand((40000, 128), 8))                     # slabs written into the same buffer
    y1 = ig.linear(x, w)
    assert torch.equal(y0, y1)
    _close(y1, x.float() @ w.float().t())


def test_unsupported_geometry_raises():
    ig = _ig()
    from partdistillation_amd.lib import PdHipError
    with pytest.raises(PdHipError):
        ig.linear(_rand((64, 48), 1), _rand((64, 48), 2))
    with pytest.raises(RuntimeError):
        ig.linear(torch.zeros((64, 64), dtype=torch.bfloat16), torch.zeros((64, 64), dtype=torch.bfloat16))


def test_wgrad_seq_equals_the_single_launches():
    """pd_wgrad_bf16_seq (the four weight gradients of a Swin block by one call: main launches back to back, ONE launch for all their slice
    sums) gives bit-identical dW / dB to pd_wgrad_bf16 per problem (same slices, same summation order), sliced and unsliced problems mixed"""
    ig = _ig()
    M = 5000
    shapes = [(1536, 512), (512, 512), (2048, 512), (512, 2048), (64, 64)]                 # (N, K): qkv, proj, fc1, fc2 of Swin-B stage 3 + a one-tile problem
    items, singles = [], []
    for i, (N, K) in enumerate(shapes):
        dy, x = _rand((M, N), 50 + i), _rand((M, K), 60 + i)
        dw, db = torch.empty((N, K), dtype=torch.bfloat16 if i % 2 == 0 else torch.float32, device=DEV), torch.zeros(N, device=DEV)
        items.append((dy, x, dw, db))
        db1 = torch.zeros(N, device=DEV)
        singles.append((ig.wgrad(dy, x, None, db1, out_dtype=dw.dtype), db1))
    ig.wgrad_seq(items)
    for (dy, x, dw, db), (dw1, db1) in zip(items, singles):
        assert torch.equal(dw, dw1) and torch.equal(db, db1)
        _close(dw, dy.float().t() @ x.float())


def test_grouped_tables_are_keyed_by_shape_as_well_as_address():
    """igemm.transposed / mx8.quantize_grouped cache table staging, offsets and the output size per operand list.  Two weights of different
    shapes at ONE address (the allocator recycles a freed copy's address for the next layer's) must not share an entry: the second call
    used to get the first one's buffer size — an out-of-bounds transpose."""
    from partdistillation_amd.functions import igemm, mx8
    buf = (torch.randn(512 * 128, device="cuda") * 2).to(torch.bfloat16)
    w1, w2 = buf[:128 * 128].view(128, 128), buf.view(512, 128)
    assert w1.data_ptr() == w2.data_ptr()
    for w in (w1, w2, w1):
        (t,) = igemm.transposed([w])
        assert t.shape == (w.shape[1], w.shape[0]) and torch.equal(t, w.t())
        ((q, s),) = mx8.quantize_grouped([w])
        q1, s1 = mx8.quantize(w)
        assert q.shape == w.shape and torch.equal(q, q1) and torch.equal(s, s1)


@pytest.mark.parametrize("M,K,N,S", [(32768, 256, 768, 256), (2048, 256, 768, 256), (1000, 320, 512, 128), (130, 64, 256, 256)])
def test_column_slabs_are_the_dense_results_of_the_separate_linears(M, K, N, S):
    """PdIgemm.out_col_slab: several Linears over the same rows as ONE product whose [N / S, M, S] output holds each Linear's result as a
    dense matrix — bit-identical to the separate launches (same tiles, same K order), ragged last row tile included; misuse raises."""
    from partdistillation_amd.functions import igemm as ig
    from partdistillation_amd.lib import PdHipError
    torch.manual_seed(M + N)
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    got = ig.linear(x, w, b, out_col_slab=S)
    assert got.shape == (N // S, M, S) and got.is_contiguous()
    for j in range(N // S):
        want = ig.linear(x, w[j * S:(j + 1) * S].contiguous(), b[j * S:(j + 1) * S].contiguous())
        assert torch.equal(got[j], want), (j, float((got[j].float() - want.float()).abs().max()))
    with pytest.raises((PdHipError, AssertionError)):
        ig.linear(x, w, b, out_col_slab=96)
