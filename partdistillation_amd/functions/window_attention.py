"""Swin (shifted-)window attention, 12 x 12 windows, head_dim 32, bf16 (pd_window_attn_*_w12,
include/pd_window_attention.h): everything between the qkv Linear and the proj Linear of the reference's
WindowAttention.forward (modeling/backbone/swin.py:135-175) in one kernel each way."""
import numpy as np
import torch
from torch.autograd import Function

from .. import lib as _lib

WINDOW, TOKENS, HEAD_DIM = 12, 144, 32
_REGIONS = {}


def supported(qkv, window_size, num_heads, dropout_p):
    """the fused kernels cover the shipped Swin-B / Swin-L setting (window 12, head_dim 32) in bf16 on the GPU; other
    settings (window 7 Swin-T/S, fp32 parity runs) take torch's library attention in WindowAttention.forward"""
    return (qkv.is_cuda and qkv.dtype == torch.bfloat16 and tuple(window_size) == (WINDOW, WINDOW) and dropout_p == 0.0
            and qkv.shape[-1] == 3 * num_heads * HEAD_DIM and qkv.shape[-2] == TOKENS)


def shifted_window_regions(H, W, shift, device):
    """region label of every token of every window of the padded, cyclically shifted grid — the labels the reference
    paints into img_mask (swin.py:425-433); its additive mask is -100 where two tokens' labels differ (:438-441).
    -> (uint8 [nW, 144], uint8 [nW] = window has more than one region)"""
    key = (H, W, shift, str(device))
    if key not in _REGIONS:
        ws = WINDOW
        Hp, Wp = int(np.ceil(H / ws)) * ws, int(np.ceil(W / ws)) * ws
        img = torch.zeros((Hp, Wp), dtype=torch.uint8)
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                img[hs, wsl] = cnt
                cnt += 1
        reg = img.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1, TOKENS).contiguous()
        flags = (reg != reg[:, :1]).any(1).to(torch.uint8)
        _REGIONS[key] = (reg.to(device), flags.to(device))
    return _REGIONS[key]


def fwd_raw(qkv, table, regions, scale, n_windows, mx=None):
    """-> (out bf16 [B_, 144, C], lse) and, with mx = an fp8 format of functions/mx8.py, out again as an MX operand (q uint8 [B_ * 144, C],
    exponents uint8 [B_ * 144, C / 32])"""
    B_, heads = qkv.shape[0], table.shape[1]
    assert qkv.is_contiguous() and qkv.dtype == torch.bfloat16 and table.dtype == torch.float32 and table.is_contiguous()
    out = torch.empty((B_, TOKENS, heads * HEAD_DIM), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((B_, heads, TOKENS), dtype=torch.float32, device=qkv.device)
    reg, flags = regions if regions is not None else (None, None)
    C = heads * HEAD_DIM
    oq = torch.empty((B_ * TOKENS, C), dtype=torch.uint8, device=qkv.device) if mx is not None else None
    osc = torch.empty((B_ * TOKENS, C // 32), dtype=torch.uint8, device=qkv.device) if mx is not None else None
    _lib.check(_lib.load().pd_window_attn_fwd_w12(qkv.data_ptr(), table.data_ptr(), reg.data_ptr() if reg is not None else None,
                                                  flags.data_ptr() if flags is not None else None, out.data_ptr(), lse.data_ptr(),
                                                  B_, n_windows, heads, float(scale), oq.data_ptr() if mx is not None else None,
                                                  osc.data_ptr() if mx is not None else None, mx if mx is not None else 0, _lib.current_stream()))
    return (out, lse) if mx is None else (out, lse, (oq, osc))


def bwd_raw(qkv, table, regions, out, d_out, lse, scale, n_windows, dtable=None, mx=None):
    """dtable: a ZERO-FILLED fp32 [529, heads] slot for the bias-table gradient (a stage hands out slices of one buffer, one fill
    for all its blocks); None allocates one.  mx = an fp8 format: -> (dqkv, dtable, (q, exponents)), dqkv again as an MX operand"""
    B_, heads = qkv.shape[0], table.shape[1]
    assert d_out.is_contiguous() and d_out.dtype == torch.bfloat16
    dqkv = torch.empty_like(qkv)
    if dtable is None:
        dtable = torch.zeros_like(table)
    else:
        assert dtable.shape == table.shape and dtable.dtype == table.dtype and dtable.is_contiguous()
    reg, flags = regions if regions is not None else (None, None)
    C3 = qkv.shape[-1]
    dq = torch.empty((B_ * TOKENS, C3), dtype=torch.uint8, device=qkv.device) if mx is not None else None
    dsc = torch.empty((B_ * TOKENS, C3 // 32), dtype=torch.uint8, device=qkv.device) if mx is not None else None
    _lib.check(_lib.load().pd_window_attn_bwd_w12(qkv.data_ptr(), table.data_ptr(), reg.data_ptr() if reg is not None else None,
                                                  flags.data_ptr() if flags is not None else None, out.data_ptr(), d_out.data_ptr(),
                                                  lse.data_ptr(), dqkv.data_ptr(), dtable.data_ptr(), B_, n_windows, heads,
                                                  float(scale), dq.data_ptr() if mx is not None else None, dsc.data_ptr() if mx is not None else None,
                                                  mx if mx is not None else 0, _lib.current_stream()))
    return (dqkv, dtable) if mx is None else (dqkv, dtable, (dq, dsc))


class WindowAttention12(Function):
    """qkv bf16 [B_, 144, 3C] (the qkv Linear's output, untouched), table fp32 [529, heads], regions = None or
    shifted_window_regions(...) -> bf16 [B_, 144, C] ready for the proj Linear."""

    @staticmethod
    def forward(ctx, qkv, table, regions, scale, n_windows):
        if not qkv.is_cuda:
            raise RuntimeError("pd_window_attn_fwd_w12 runs on the GPU only (no CPU fallback in partdistillation_amd)")
        qkv, table = qkv.contiguous(), table.contiguous()
        out, lse = fwd_raw(qkv, table, regions, scale, n_windows)
        ctx.save_for_backward(qkv, table, out, lse)
        ctx.regions, ctx.scale, ctx.n_windows = regions, scale, n_windows
        return out

    @staticmethod
    def backward(ctx, d_out):
        qkv, table, out, lse = ctx.saved_tensors
        dqkv, dtable = bwd_raw(qkv, table, ctx.regions, out, d_out.contiguous(), lse, ctx.scale, ctx.n_windows)
        return dqkv, dtable, None, None, None


def window_attention(qkv, table, regions, scale, n_windows):
    return WindowAttention12.apply(qkv, table, regions, scale, n_windows)
