"""GPU tests of the bf16 NHWC backbone convolutions (csrc/conv_bf16.hip) against torch's fp32 convolution of the same
bf16-valued operands: every distinct geometry of R50 (1x1 / 3x3, stride 1 / 2, 64..2048 channels), ragged images whose pixel
count is not a multiple of the 128-row tile, the fused frozen-BN + residual + ReLU epilogue and the dgrad addend."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"

GEOMS = [  # ci, co, k, stride, H, W, batch
    (64, 64, 1, 1, 24, 40, 2), (64, 256, 1, 1, 24, 40, 2), (256, 64, 1, 1, 17, 23, 3), (64, 64, 3, 1, 24, 40, 2),
    (128, 128, 3, 2, 24, 40, 2), (256, 512, 1, 2, 24, 40, 2), (128, 128, 3, 2, 23, 41, 1), (256, 512, 1, 2, 23, 41, 1),
    (512, 128, 1, 1, 12, 20, 2), (256, 256, 3, 1, 12, 20, 2), (1024, 2048, 1, 2, 12, 10, 1), (512, 512, 3, 1, 6, 5, 2),
    (2048, 512, 1, 1, 6, 5, 2), (192, 320, 3, 1, 9, 9, 1),
]


def _mk(shape, seed, scale=1.0, cl=True):
    g = torch.Generator(device="cpu").manual_seed(seed)
    t = (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).to(DEV)
    return t.contiguous(memory_format=torch.channels_last) if cl and t.dim() == 4 else t


@pytest.mark.parametrize("ci,co,k,s,H,W,B", GEOMS)
def test_conv_fwd_epilogue_vs_torch_fp32(ci, co, k, s, H, W, B):
    from partdistillation_amd.functions import conv_bf16 as C
    x, w = _mk((B, ci, H, W), 1), _mk((co, ci, k, k), 2, (ci * k * k) ** -0.5)
    assert C.supported(x, w, s, k // 2)
    ref = F.conv2d(x.float(), w.float(), None, s, k // 2)
    y = C.conv_fwd(x, w, stride=s, pad=k // 2)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    tol = dict(rtol=2 ** -7, atol=2e-2)                              # one bf16 rounding of an O(1) fp32-accumulated sum
    torch.testing.assert_close(y.float(), ref, **tol)
    scale = torch.rand(co, device=DEV) + 0.5
    bias = torch.randn(co, device=DEV)
    res = _mk(ref.shape, 3)
    y2 = C.conv_fwd(x, w, scale, bias, res, True, s, k // 2)
    ref2 = F.relu(ref * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1) + res.float())
    torch.testing.assert_close(y2.float(), ref2, **tol)
    y3 = C.conv_fwd(x, w, scale, bias, None, False, s, k // 2)
    torch.testing.assert_close(y3.float(), ref * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1), **tol)


@pytest.mark.parametrize("ci,co,k,s,H,W,B", GEOMS)
def test_conv_dgrad_vs_torch_fp32(ci, co, k, s, H, W, B):
    from partdistillation_amd.functions import conv_bf16 as C
    x, w = _mk((B, ci, H, W), 1), _mk((co, ci, k, k), 2, (co * k * k) ** -0.5)
    xf = x.float().requires_grad_(True)
    ref_y = F.conv2d(xf, w.float(), None, s, k // 2)
    dz = _mk(ref_y.shape, 4)
    (ref,) = torch.autograd.grad(ref_y, xf, dz.float())
    wt = C.transposed_filter(w)
    assert wt.shape == (ci, k, k, co) and wt.is_contiguous()
    dx = C.conv_dgrad(dz, wt, x.shape, k, s, k // 2)
    tol = dict(rtol=2 ** -7, atol=2e-2)
    assert dx.shape == x.shape and dx.is_contiguous(memory_format=torch.channels_last)
    torch.testing.assert_close(dx.float(), ref, **tol)
    add = _mk(x.shape, 5)
    dx2 = C.conv_dgrad(dz, wt, x.shape, k, s, k // 2, addend=add)
    torch.testing.assert_close(dx2.float(), ref + add.float(), **tol)


@pytest.mark.parametrize("ci,co,k,s,H,W,B", GEOMS + [(64, 64, 7, 2, 40, 24, 2), (8, 24, 3, 1, 33, 17, 3)])
def test_conv_wgrad_vs_torch_fp32(ci, co, k, s, H, W, B):
    """pd_conv_bf16_wgrad (transpose-read kernel + partial-tile reduce) against the fp32 filter gradient: tiles that span several
    taps (ci = 64, 8), pixel ranges that end inside an image row, strides, a 7 x 7 filter."""
    from partdistillation_amd.functions import conv_bf16 as C
    x, w = _mk((B, ci, H, W), 1), _mk((co, ci, k, k), 2, 0.05)
    wf = w.float().requires_grad_(True)
    ref_y = F.conv2d(x.float(), wf, None, s, k // 2)
    dz = _mk(ref_y.shape, 4)
    (ref,) = torch.autograd.grad(ref_y, wf, dz.float())
    dw = C.conv_wgrad(dz, x, k, s, k // 2, like=w)
    assert dw.shape == w.shape and dw.dtype == torch.bfloat16 and dw.stride() == w.stride()
    err = (dw.float() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2 ** -7, err                                        # one bf16 rounding of an fp32-accumulated sum


def test_conv_wgrad_tiny_and_single_pixel_problems():
    """fewer pixels than one 32-pixel stage, one image of one pixel, more splits than stages: the plan's lower bounds"""
    from partdistillation_amd.functions import conv_bf16 as C
    for (ci, co, k, s, H, W, B) in [(64, 64, 1, 1, 1, 1, 1), (128, 256, 3, 1, 2, 3, 1), (8, 8, 1, 1, 5, 5, 2), (256, 64, 3, 2, 3, 3, 4)]:
        x, w = _mk((B, ci, H, W), 7), _mk((co, ci, k, k), 8, 0.05)
        wf = w.float().requires_grad_(True)
        ref_y = F.conv2d(x.float(), wf, None, s, k // 2)
        dz = _mk(ref_y.shape, 9)
        (ref,) = torch.autograd.grad(ref_y, wf, dz.float())
        dw = C.conv_wgrad(dz, x, k, s, k // 2, like=w)
        assert (dw.float() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item() + 1e-6


def test_deferred_grouped_wgrads_equal_the_immediate_ones():
    """Conv2dOwnWgrad inside deferred_wgrads(): backward hands out unwritten filter gradients and ONE grouped launch per tile shape
    fills them at flush() — all four tile shapes, strides, ragged pixel counts, two uses of the context in a row."""
    from partdistillation_amd.functions import conv_bf16 as C
    geoms = [(64, 64, 1, 1, 24, 40, 2), (64, 256, 1, 1, 24, 40, 2), (256, 64, 1, 1, 17, 23, 3), (64, 64, 3, 1, 24, 40, 2),
             (128, 128, 3, 2, 23, 41, 1), (256, 512, 1, 2, 23, 41, 1), (512, 512, 3, 1, 6, 5, 2), (2048, 512, 1, 1, 6, 5, 2)]
    for rep in range(2):
        xs, ws, ys, gs = [], [], [], []
        for i, (ci, co, k, s, H, W, B) in enumerate(geoms):
            x = _mk((B, ci, H, W), 10 * rep + i).requires_grad_(True)
            w = _mk((co, ci, k, k), 100 + i, 0.05).requires_grad_(True)
            assert C.own_wgrad_supported(x, w, (s, s), (k // 2, k // 2))
            y = C.conv2d_own_wgrad(x, w, (s, s), (k // 2, k // 2))
            xs.append(x), ws.append(w), ys.append(y), gs.append(_mk(y.shape, 200 + i))
        total = sum((y.float() * g.float()).sum() for y, g in zip(ys, gs))
        with C.deferred_wgrads():
            total.backward()
            assert len(C._Deferred.queue) == len(geoms)
        assert not C._Deferred.queue and not C._Deferred.active
        for (ci, co, k, s, H, W, B), x, w, g in zip(geoms, xs, ws, gs):
            wf = w.detach().float().requires_grad_(True)
            xf = x.detach().float().requires_grad_(True)
            rx, rw = torch.autograd.grad(F.conv2d(xf, wf, None, s, k // 2), (xf, wf), g.float())
            assert w.grad.stride() == w.stride()
            assert (w.grad.float() - rw).abs().max().item() / rw.abs().max().item() < 2 ** -7
            assert (x.grad.float() - rx).abs().max().item() / rx.abs().max().item() < 2 ** -6
            imm = C.conv_wgrad(g, x.detach(), k, s, k // 2, like=w)
            assert (w.grad.float() - imm.float()).abs().max().item() / rw.abs().max().item() < 2 ** -7


def test_rows_problems_with_bias_sums_in_the_grouped_launch():
    """Linear weight gradients over many rows as 1 x 1 "convolutions" (the decoder's key / value projections), with the bias
    gradient (column sums of dy, fp32 accumulator) produced by the same launch."""
    from partdistillation_amd.functions import conv_bf16 as C
    entries, checks = [], []
    for i, (M, N, K) in enumerate([(32768, 256, 256), (2048, 256, 256), (777, 64, 128), (5000, 128, 64), (100, 64, 64)]):
        dy, x = _mk((M, N), 300 + i), _mk((M, K), 400 + i)
        dw = torch.empty((N, K), dtype=torch.bfloat16, device=DEV)
        acc = torch.full((N,), 0.5, device=DEV)
        entries.append(C.rows_entry(dy, x, dw, acc))
        checks.append((dy, x, dw, acc))
    C.run_now(entries)
    for dy, x, dw, acc in checks:
        ref = dy.float().t() @ x.float()
        assert (dw.float() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item()
        torch.testing.assert_close(acc - 0.5, dy.float().sum(0), rtol=1e-4, atol=2e-2)


def test_conv_rejects_what_it_does_not_cover():
    from partdistillation_amd import lib
    from partdistillation_amd.functions import conv_bf16 as C
    x, w = _mk((1, 3, 16, 16), 1), _mk((64, 3, 7, 7), 2)
    assert not C.supported(x, w, 2, 3)
    with pytest.raises(lib.PdHipError):
        C.conv_fwd(x, w, stride=2, pad=3)
    x, w = _mk((1, 64, 8, 8), 1), _mk((96, 64, 1, 1), 2)
    assert not C.supported(x, w, 1, 0)


@pytest.mark.parametrize("B,C,H,W", [(2, 64, 64, 96), (1, 8, 7, 5), (3, 16, 33, 18)])
def test_own_maxpool_matches_aten_forward_and_backward_including_ties(B, C, H, W):
    """pd_maxpool3s2_{fwd,bwd}_bf16 vs F.max_pool2d(3, 2, 1) on post-ReLU-like bf16 maps (most windows tie at 0: the gradient must go to
    the FIRST maximum in window order, as ATen routes it): outputs and input gradients bit-identical."""
    import torch.nn.functional as F
    from partdistillation_amd.functions.fused import max_pool3x3s2, max_pool3x3s2_supported
    g = torch.Generator(device="cuda").manual_seed(H * W)
    x = torch.randn(B, C, H, W, device="cuda", generator=g).relu().bfloat16().contiguous(memory_format=torch.channels_last)
    x[:, :, ::3] = 0
    assert max_pool3x3s2_supported(x)
    x1, x2 = x.clone().requires_grad_(), x.clone().requires_grad_()
    y1, y2 = max_pool3x3s2(x1), F.max_pool2d(x2, 3, 2, 1)
    assert torch.equal(y1, y2)
    go = torch.randn(y2.shape, device="cuda", generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    y1.backward(go); y2.backward(go)
    torch.testing.assert_close(x1.grad.float(), x2.grad.float(), rtol=1e-2, atol=1e-2)   # up to 4 bf16 addends per pixel: summation order
    assert torch.equal(x1.grad != 0, x2.grad != 0)
