"""pd_stem7x7_fwd / pd_stem7x7_wgrad (include/pd_stem.h) through the C-ABI against torch's fp32 convolution on the same bf16-rounded operands:
detectron2 0.6 BasicStem.conv1 (Conv2d(3, 64, 7, stride 2, padding 3, bias=False) + FrozenBN + ReLU), the reference's R50 stem."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(B, H, W, seed, xdtype):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = (torch.randn(B, 3, H, W, device=DEV, generator=g) * 1.5).contiguous(memory_format=torch.channels_last)
    if xdtype == torch.bfloat16:
        x = x.bfloat16()
    w = (torch.randn(64, 3, 7, 7, device=DEV, generator=g) * 0.08).bfloat16().contiguous(memory_format=torch.channels_last)
    scale = torch.rand(64, device=DEV, generator=g) + 0.5
    bias = torch.randn(64, device=DEV, generator=g) * 0.3
    return x, w, scale, bias


@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (1, 37, 53), (1, 256, 512), (2, 130, 258), (1, 8, 520)])
@pytest.mark.parametrize("xdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("relu", [True, False])
def test_stem_forward_and_filter_gradient_vs_torch(B, H, W, xdtype, relu):
    """forward: one bf16 rounding of the fp32 result (<= 2^-8 relative + accumulation-order noise); filter gradient: bf16 result of fp32 sums
    over up to 10^5 pixels: within 4e-3 of the tensor's maximum (the R50 body's per-launch bound, tests/test_r50_fused_gpu.py)."""
    from partdistillation_amd.functions.stem import StemConv
    x, w, scale, bias = _case(B, H, W, B * 1000 + H + W, xdtype)
    wp = w.detach().clone().requires_grad_()
    y = StemConv.apply(x, wp, scale, bias, relu)
    xr, wr = x.bfloat16().float(), w.float().detach().requires_grad_()
    ref = F.conv2d(xr, wr, None, 2, 3) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    ref = ref.relu() if relu else ref
    assert y.shape == ref.shape and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    err = (y.float() - ref).abs()
    assert (err <= 1e-2 * ref.abs() + 2e-3 * ref.abs().max()).all(), (err.max().item(), ref.abs().max().item())
    gy = torch.randn(y.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(7)).bfloat16().contiguous(memory_format=torch.channels_last)
    y.backward(gy)
    # the reference gradient with the PRODUCT's ReLU mask (a pre-activation within rounding of zero may round to either side)
    g_ref = gy.float() * ((y.float() > 0).float() if relu else 1.0) * scale.view(1, -1, 1, 1)
    dw_ref = torch.nn.grad.conv2d_weight(xr, w.shape, g_ref, stride=2, padding=3)
    assert wp.grad.dtype == torch.bfloat16 and wp.grad.shape == w.shape and wp.grad.is_contiguous(memory_format=torch.channels_last)
    e = (wp.grad.float() - dw_ref).abs().max().item() / dw_ref.abs().max().item()
    assert e < 4e-3, e


def test_stem_module_path_takes_the_own_kernels_and_matches_the_library_path():
    """BasicStem under bf16 autocast: PD_OWN_STEM on (pd_stem7x7_*) vs off (MIOpen + pd_affine_act): same pooled output and filter gradient
    to bf16 noise, and the own path really ran."""
    from partdistillation_amd.functions import stem as S
    from partdistillation_amd.modeling.backbone import resnet as R
    torch.manual_seed(0)
    m = R.BasicStem(3, 64, "FrozenBN").to(DEV)
    m.conv1.norm.weight.uniform_(0.5, 1.5); m.conv1.norm.bias.normal_(0, 0.2); m.conv1.norm.running_mean.normal_(0, 0.2); m.conv1.norm.running_var.uniform_(0.5, 2.0)
    m.conv1.weight.data = m.conv1.weight.data.bfloat16().contiguous(memory_format=torch.channels_last)
    x = torch.randn(2, 3, 128, 160, device=DEV).contiguous(memory_format=torch.channels_last)
    calls = {"n": 0}
    f0 = S.stem_conv
    S.stem_conv = lambda *a, **k: (calls.__setitem__("n", calls["n"] + 1), f0(*a, **k))[1]
    out = {}
    try:
        for own in (True, False):
            R.OWN_STEM = own
            m.conv1.weight.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x)
            (y.float() * torch.linspace(-1, 1, y.numel(), device=DEV).view_as(y)).sum().backward()
            out[own] = (y.detach().float(), m.conv1.weight.grad.detach().float())
    finally:
        R.OWN_STEM, S.stem_conv = True, f0
    assert calls["n"] == 1
    assert ((out[True][0] - out[False][0]).abs() <= 2e-2 * out[False][0].abs() + 2e-2).all()
    assert ((out[True][1] - out[False][1]).abs().max() / out[False][1].abs().max()).item() < 5e-2
