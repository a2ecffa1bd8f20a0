"""Python face of the bf16 NHWC backbone convolutions (include/pd_conv.h, csrc/conv_bf16.hip): one launch = convolution +
frozen-BN affine + residual add + ReLU; the input gradient = transposed convolution + the gradient arriving over the
shortcut.  GPU only; no fallback."""
import torch
from torch.autograd import Function

from .. import lib as _lib


def _stream():
    return _lib.current_stream()


def supported(x, weight, stride, padding, dilation=1, groups=1):
    co, ci, kh, kw = weight.shape
    return (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and kh == kw and groups == 1 and dilation == 1
            and bool(_lib.load().pd_conv_bf16_supported(ci, co, kh, stride, padding)))


def _nhwc(t):
    return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)


def conv_fwd(x, w, scale=None, bias=None, residual=None, relu=False, stride=1, pad=0):
    """x [B,ci,H,W], w [co,ci,k,k], residual [B,co,Ho,Wo]: bf16, channels_last storage; scale / bias fp32 [co] -> y (channels_last)"""
    B, ci, H, W = x.shape
    co, _, k, _ = w.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty((B, co, Ho, Wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = _lib.load().pd_conv_bf16_fwd(x.data_ptr(), w.data_ptr(), scale.data_ptr() if scale is not None else None,
                                          bias.data_ptr() if bias is not None else None,
                                          residual.data_ptr() if residual is not None else None, y.data_ptr(),
                                          B, H, W, ci, Ho, Wo, co, k, stride, pad, int(relu), _stream())
    _lib.check(rc)
    return y


def conv_dgrad(dz, wt, x_shape, k, stride=1, pad=0, addend=None):
    """dz [B,co,Ho,Wo] (channels_last), wt = filter stored [ci][k][k][co] -> dx [B,ci,H,W] (+ addend)"""
    B, ci, H, W = x_shape
    co, Ho, Wo = dz.shape[1], dz.shape[2], dz.shape[3]
    dx = torch.empty((B, ci, H, W), dtype=torch.bfloat16, device=dz.device, memory_format=torch.channels_last)
    with torch.cuda.device(dz.device):
        rc = _lib.load().pd_conv_bf16_dgrad(dz.data_ptr(), wt.data_ptr(), addend.data_ptr() if addend is not None else None, dx.data_ptr(),
                                            B, H, W, ci, Ho, Wo, co, k, stride, pad, _stream())
    _lib.check(rc)
    return dx


def transposed_filter(w):
    """[co,ci,k,k] (channels_last storage = [co][k][k][ci]) -> a tensor whose storage is [ci][k][k][co]"""
    return w.permute(1, 2, 3, 0).contiguous()
