"""fp32 3 x 3 convolution (stride 1, pad 1) on the bf16 matrix cores with fp32-level accuracy (pd_conv3x3_nhwc_f32x3,
include/pd_gemm.h): forward and input gradient are implicit GEMMs on the exact 3-way bf16 split of csrc/gemm_x3.hip; the weight
gradient is the transpose-read split kernel of the encoder's weight gradients with the im2col gather in its staging.  Channels-last tensors only (the pixel decoder keeps its maps NHWC)."""
import torch
from torch.autograd import Function

from .. import lib as _lib


H2 = __import__("os").environ.get("PD_H2_CONV", "1") != "0"   # the fp16 two-plane form (pd_conv3x3_nhwc_f16x2 / pd_conv3x3_wgrad_nhwc_f16x2: three products per
                     # term instead of six, pixels scaled by powers of two from their channel maxima); False: the 3-plane bf16 kernels
H2_1X1 = __import__("os").environ.get("PD_H2_1X1", "1") != "0"   # ... and the 1 x 1 convolutions (forward / input gradient / filter gradient as GEMMs on the NHWC rows)
WGRAD_X3 = True      # False: MIOpen's fp32 weight gradient (1.53 ms at 2 x 256 x 256^2, the transposed-read split kernel: see DESIGN.md)


def supported(x, conv):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv.weight.dtype == torch.float32
            and tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (1, 1)
            and tuple(conv.dilation) == (1, 1) and conv.groups == 1 and x.shape[1] % 16 == 0 and conv.weight.shape[0] % 16 == 0
            and x.is_contiguous(memory_format=torch.channels_last))


def _pixel_amax(x):
    """absolute maximum over the channels of every pixel of a channels-last map -> [B*H*W] (the row maxima of its NHWC rows)"""
    from . import amax_cache
    from .gemm import row_amax
    am = amax_cache.get(x)
    return am if am is not None else row_amax(x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]))


def _raw(x, wk, bias, co, x_amax=None):
    """x channels-last [B,Ci,H,W]; wk [Co,3,3,Ci] contiguous -> channels-last [B,Co,H,W]"""
    B, ci, H, W = x.shape
    y = torch.empty((B, co, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    from .gemm import _timed_fwd, row_amax
    if x_amax is not None:
        w_amax = row_amax(wk.view(co, -1))
        with _timed_fwd(2.0 * B * H * W * 9 * ci * co, "gemm_tn_f16x2<256, 256, 128, 16, 0, conv 3x3>", 4.0 * (B * H * W * (ci + co) + 9 * ci * co)):
            _lib.check(_lib.load().pd_conv3x3_nhwc_f16x2(x.data_ptr(), wk.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                                                         x_amax.data_ptr(), w_amax.data_ptr(), None, B, H, W, ci, co, _lib.current_stream()))
        return y
    with _timed_fwd(2.0 * B * H * W * 9 * ci * co, "gemm_tn_f32x3<conv 3x3>"):
        _lib.check(_lib.load().pd_conv3x3_nhwc_f32x3(x.data_ptr(), wk.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                                                     B, H, W, ci, co, _lib.current_stream()))
    return y


class Conv3x3X3(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        if not x.is_cuda:
            raise RuntimeError("pd_conv3x3_nhwc_f32x3 runs on the GPU only (no CPU fallback in partdistillation_amd)")
        x = x.contiguous(memory_format=torch.channels_last)
        wk = weight.permute(0, 2, 3, 1).contiguous()                       # [Co,3,3,Ci]: free when the filter is stored channels-last
        x_am = _pixel_amax(x) if H2 else None
        ctx.save_for_backward(x, weight, x_am)
        ctx.has_bias = bias is not None
        return _raw(x, wk, bias, weight.shape[0], x_am)

    @staticmethod
    def backward(ctx, dy):
        x, weight, x_am = ctx.saved_tensors
        dy = dy.contiguous(memory_format=torch.channels_last)
        co, ci = weight.shape[0], weight.shape[1]
        dx = dw = db = None
        dy_am = _pixel_amax(dy) if x_am is not None else None
        if ctx.needs_input_grad[0]:
            # dX[p, ci] = sum_tap sum_co dY[p - off(tap), co] W[co, tap, ci]: the same kernel on dY with the taps flipped and the
            # filter transposed to [Ci][3][3][Co] (2.4 MB at 256 channels)
            wt = weight.permute(0, 2, 3, 1).reshape(co, 9, ci).flip(1).permute(2, 1, 0).contiguous()
            dx = _raw(dy, wt, None, ci, dy_am)
        want_w, want_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        if (want_w or want_b) and WGRAD_X3 and ci % 128 == 0 and co % 4 == 0:
            # the same split on the weight gradient: contraction over the pixels, tiles transposed on their way out of LDS
            from .gemm import _wgrad_workspace
            L = _lib.load()
            B, _, H, W = x.shape
            buf = torch.zeros(co * 9 * ci + co, dtype=torch.float32, device=x.device)
            dwk, dbv = buf[:co * 9 * ci].view(co, 3, 3, ci), buf[co * 9 * ci:]
            ws = _wgrad_workspace(x.device, int((L.pd_gemm_wgrad_f16x2_ws_floats if x_am is not None else L.pd_gemm_wgrad_f32x3_ws_floats)(co, 9 * ci)))
            if x_am is not None:
                _lib.check(L.pd_conv3x3_wgrad_nhwc_f16x2(dy.data_ptr(), x.data_ptr(), dwk.data_ptr(), dbv.data_ptr() if want_b else None,
                                                         dy_am.data_ptr(), x_am.data_ptr(), ws.data_ptr() if ws is not None else None,
                                                         ws.numel() if ws is not None else 0, B, H, W, ci, co, _lib.current_stream()))
            else:
                _lib.check(L.pd_conv3x3_wgrad_nhwc_f32x3(dy.data_ptr(), x.data_ptr(), dwk.data_ptr(), dbv.data_ptr() if want_b else None,
                                                         ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
                                                         B, H, W, ci, co, _lib.current_stream()))
            dw = dwk.permute(0, 3, 1, 2) if want_w else None                # [Co,Ci,3,3] view with channels-last strides
            db = dbv if want_b else None
        elif want_w or want_b:
            _, dw, db = torch.ops.aten.convolution_backward(dy, x, weight, [co] if ctx.has_bias else None, [1, 1], [1, 1], [1, 1], False,
                                                            [0, 0], 1, [False, True, ctx.has_bias])
        return dx, dw, db


def conv3x3(x, weight, bias=None):
    return Conv3x3X3.apply(x, weight, bias)


class Conv1x1OwnWgrad(Function):
    """fp32 1 x 1 convolution on channels-last maps = a GEMM on the NHWC rows.  H2 (default): forward, input gradient and filter /
    bias gradient on the fp16 two-plane kernels (pd_gemm_tn_f16x2 / pd_gemm_wgrad_acc_f16x2_ws) with the pixels' channel maxima as
    row scales (taken from functions/amax_cache when the producing kernel left them there).  Otherwise: library forward / input
    gradient, 3-plane bf16 filter gradient.  The pixel decoder's input projections, lateral and mask-feature convolutions
    (reference msdeformattn.py:200-257)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.has_bias = bias is not None
        ctx.x_bf16 = x.dtype == torch.bfloat16
        if H2 and H2_1X1 and x.is_contiguous(memory_format=torch.channels_last):
            from .gemm import cast_rows_amax, gemm_tn_h2, row_amax
            B, ci, Hh, Ww = x.shape
            co = weight.shape[0]
            if ctx.x_bf16:
                # a bf16 backbone map (the reference's `features[f].float()`, msdeformattn.py:324, 338): its fp32 copy and its pixels' channel
                # maxima in ONE pass, and the input gradient goes back as bf16 from the GEMM's epilogue (no cast launches either way)
                x2, x_am = cast_rows_amax(x.permute(0, 2, 3, 1).reshape(-1, ci))
                x = x2.view(B, Hh, Ww, ci).permute(0, 3, 1, 2)
            else:
                x_am = _pixel_amax(x)
            w2 = weight.reshape(co, ci)
            y = gemm_tn_h2(x.permute(0, 2, 3, 1).reshape(-1, ci), w2, bias, a_amax=x_am, b_amax=row_amax(w2))
            ctx.save_for_backward(x, weight, x_am)
            return y.view(B, Hh, Ww, co).permute(0, 3, 1, 2)
        if ctx.x_bf16:
            x = x.float()
        ctx.save_for_backward(x, weight, None)
        return torch.ops.aten.convolution(x, weight, bias, [1, 1], [0, 0], [1, 1], False, [0, 0], 1)

    @staticmethod
    def backward(ctx, dy):
        from .gemm import gemm_tn_h2, gemm_tn_h2_bf16out, gemm_wgrad_acc, row_amax
        x, weight, x_am = ctx.saved_tensors
        dy = dy.contiguous(memory_format=torch.channels_last)
        co, ci = weight.shape[0], weight.shape[1]
        dx = dw = db = None
        rows = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])          # NHWC storage -> [pixels, channels] view
        h2 = x_am is not None
        dy_am = _pixel_amax(dy) if h2 else None
        if ctx.needs_input_grad[0]:
            if h2:
                wt = weight.reshape(co, ci).t().contiguous()                       # [Ci, Co]: the B operand of dX = dY W
                B, _, Hh, Ww = x.shape
                mm = gemm_tn_h2_bf16out if ctx.x_bf16 else gemm_tn_h2
                dx = mm(rows(dy), wt, None, a_amax=dy_am, b_amax=row_amax(wt)).view(B, Hh, Ww, ci).permute(0, 3, 1, 2)
            else:
                dx = torch.ops.aten.convolution_backward(dy, x, weight, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]
                dx = dx.to(torch.bfloat16) if ctx.x_bf16 else dx
        want_w, want_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        if want_w or want_b:
            buf = torch.zeros(co * ci + co, dtype=torch.float32, device=x.device)
            dw2, dbv = buf[:co * ci].view(co, ci), buf[co * ci:]
            if h2:
                gemm_wgrad_acc(rows(dy), rows(x), dw2, dbv if want_b else None, h2=True, y_amax=dy_am, x_amax=x_am)
            else:
                gemm_wgrad_acc(rows(dy), rows(x), dw2, dbv if want_b else None, x3=True)
            dw = dw2.view(co, ci, 1, 1).as_strided(weight.shape, weight.stride()) if want_w else None
            db = dbv if want_b else None
        return dx, dw, db


def conv1x1_supported(x, conv):
    return (WGRAD_X3 and x.is_cuda and (x.dtype == torch.float32 or (x.dtype == torch.bfloat16 and H2 and H2_1X1 and x.shape[1] % 8 == 0))
            and x.dim() == 4 and conv.weight.dtype == torch.float32
            and tuple(conv.kernel_size) == (1, 1) and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (0, 0)
            and tuple(conv.dilation) == (1, 1) and conv.groups == 1 and x.shape[1] % 4 == 0 and conv.weight.shape[0] % 4 == 0
            and x.is_contiguous(memory_format=torch.channels_last) and torch.is_grad_enabled())


def conv1x1(x, weight, bias=None):
    return Conv1x1OwnWgrad.apply(x, weight, bias)
