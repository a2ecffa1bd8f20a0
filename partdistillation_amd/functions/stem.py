"""The ResNet stem on own kernels (include/pd_stem.h): 7 x 7 / 2 convolution 3 -> 64 + frozen-BN affine + ReLU in ONE launch, and its
filter gradient (the image needs no gradient: no input gradient, no frozen-BN / ReLU backward pass).  Contract: detectron2 0.6 BasicStem.conv1
(un-vendored; selected by the reference's Base-COCO-InstanceSegmentation.yaml:2-15).  No fallback here: the caller (modeling/backbone/resnet.py)
asks `supported` first and otherwise keeps the library convolution."""
import torch
from torch.autograd import Function

from .. import lib as _lib


def _stream():
    return _lib.current_stream()


def supported(x, conv):
    from ..compat.layers import FrozenBatchNorm2d
    w = conv.weight
    return (x.is_cuda and x.dim() == 4 and x.shape[1] == 3 and x.dtype in (torch.float32, torch.bfloat16) and not x.requires_grad
            and x.is_contiguous(memory_format=torch.channels_last) and w.dtype == torch.bfloat16 and tuple(w.shape) == (64, 3, 7, 7)
            and w.is_contiguous(memory_format=torch.channels_last) and conv.stride == (2, 2) and conv.padding == (3, 3) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.bias is None and isinstance(conv.norm, FrozenBatchNorm2d)
            and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)


class StemConv(Function):
    """y = relu(conv7x7s2(x, w) * scale + bias): x [B, 3, H, W] channels-last fp32 / bf16 (no gradient), w [64, 3, 7, 7] channels-last bf16"""

    @staticmethod
    def forward(ctx, x, w, scale, bias, relu):
        B, _, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((B, 64, Ho, Wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        _lib.check(_lib.load().pd_stem7x7_fwd(x.data_ptr(), int(x.dtype == torch.float32), w.data_ptr(), scale.data_ptr(), bias.data_ptr(), y.data_ptr(),
                                              B, H, W, int(relu), _stream()))
        ctx.relu = relu
        ctx.save_for_backward(x, w, scale, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, scale, y = ctx.saved_tensors
        if not ctx.needs_input_grad[1]:
            return None, None, None, None, None
        B, _, H, W = x.shape
        gy = gy if gy.is_contiguous(memory_format=torch.channels_last) else gy.contiguous(memory_format=torch.channels_last)
        L = _lib.load()
        ws = torch.empty(int(L.pd_stem_wgrad_workspace_floats()), dtype=torch.float32, device=x.device)
        dw = torch.empty_like(w)                                       # channels-last like the filter (preserve_format)
        _lib.check(L.pd_stem7x7_wgrad(x.data_ptr(), int(x.dtype == torch.float32), gy.data_ptr(), y.data_ptr() if y is not None else None, scale.data_ptr(),
                                      dw.data_ptr(), 0, ws.data_ptr(), B, H, W, int(ctx.relu), _stream()))
        return None, dw, None, None, None


def stem_conv(x, conv, relu=True):
    scale, bias = conv.norm.scale_bias()
    return StemConv.apply(x, conv.weight, scale.float().contiguous(), bias.float().contiguous(), relu)
