"""Masked multi-head attention, few queries x many keys, head_dim 32 (pd_attn_*_d32, include/pd_attention.h)."""
import torch
from torch.autograd import Function

from .. import lib as _lib

_DT = {torch.float32: _lib.PD_F32, torch.bfloat16: _lib.PD_BF16}
_WS = {}


def _workspace(B, H, Lq, Lk, device):
    n = int(_lib.load().pd_attn_workspace_floats(B, H, Lq, Lk))
    from .. import cmdbuf
    if cmdbuf.active() is not None:                          # recorded region: scratch owned by the recording (its arena)
        return torch.empty(max(n, 1), dtype=torch.float32, device=device)
    key = str(device)
    ws = _WS.get(key)
    if ws is None or ws.numel() < n:
        ws = torch.empty(n, dtype=torch.float32, device=device)
        _WS[key] = ws
    return ws


def _stream():
    return _lib.current_stream()


def _ld_kv(k, v):
    """k / v [rows, H*32]: dense, or equal-stride column slices of wider row-major matrices (pd_attn_*_d32_ld)"""
    assert k.dim() == 2 and k.shape == v.shape and k.stride(1) == 1 and v.stride(1) == 1 and k.stride(0) == v.stride(0) >= k.shape[1]
    return k.stride(0)


def attn_fwd_raw(q, k, v, m8, B, nheads, scale):
    """q [Lq*B, H*32] contiguous, k / v [Lk*B, H*32] (seq-first rows l*B + b; dense or column slices of a wider matrix),
    m8 uint8 [B,Lq,Lk] | None -> (o, lse).  No autograd: building block of hand-written backward passes."""
    Lq, Lk = q.shape[0] // B, k.shape[0] // B
    o = torch.empty_like(q)
    lse = torch.empty((B, nheads, Lq), dtype=torch.float32, device=q.device)
    ws = _workspace(B, nheads, Lq, Lk, q.device)
    _lib.check(_lib.load().pd_attn_fwd_d32_ld(q.data_ptr(), k.data_ptr(), v.data_ptr(), m8.data_ptr() if m8 is not None else None,
                                              o.data_ptr(), lse.data_ptr(), ws.data_ptr(), B, nheads, Lq, Lk, float(scale),
                                              _DT[q.dtype], _ld_kv(k, v), _stream()))
    return o, lse


def attn_bwd_raw(q, k, v, m8, o, d_o, lse, B, nheads, scale):
    Lq, Lk = q.shape[0] // B, k.shape[0] // B
    dq = torch.empty_like(q)
    dk, dv = (torch.empty(k.shape, dtype=k.dtype, device=k.device) for _ in range(2))          # dense, whatever k / v are
    ws = _workspace(B, nheads, Lq, Lk, q.device)
    _lib.check(_lib.load().pd_attn_bwd_d32_ld(q.data_ptr(), k.data_ptr(), v.data_ptr(), m8.data_ptr() if m8 is not None else None,
                                              o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                              ws.data_ptr(), B, nheads, Lq, Lk, float(scale), _DT[q.dtype], _ld_kv(k, v), _stream()))
    return dq, dk, dv


class MaskedAttention32(Function):
    """softmax(q k^T * scale + mask(-inf)) v per head; q [Lq,B,H*32], k/v [Lk,B,H*32] (seq-first, contiguous);
    mask bool [B,Lq,Lk] (True = blocked) or None.  Returns o [Lq,B,H*32]."""

    @staticmethod
    def forward(ctx, q, k, v, mask, nheads, scale):
        if not q.is_cuda:
            raise RuntimeError("pd_attn_fwd_d32 runs on the GPU only (no CPU fallback in partdistillation_amd)")
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        Lq, B, C = q.shape
        Lk = k.shape[0]
        assert C == nheads * 32 and q.dtype in _DT and k.dtype == q.dtype and v.dtype == q.dtype
        m8 = None
        if mask is not None:
            m8 = mask.contiguous().view(torch.uint8)
            assert m8.shape == (B, Lq, Lk)
        o = torch.empty_like(q)
        lse = torch.empty((B, nheads, Lq), dtype=torch.float32, device=q.device)
        ws = _workspace(B, nheads, Lq, Lk, q.device)
        with torch.cuda.device(q.device):
            rc = _lib.load().pd_attn_fwd_d32(q.data_ptr(), k.data_ptr(), v.data_ptr(), m8.data_ptr() if m8 is not None else None,
                                             o.data_ptr(), lse.data_ptr(), ws.data_ptr(), B, nheads, Lq, Lk, float(scale),
                                             _DT[q.dtype], _stream())
        _lib.check(rc)
        ctx.save_for_backward(q, k, v, m8, o, lse)
        ctx.nheads, ctx.scale = nheads, float(scale)
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, m8, o, lse = ctx.saved_tensors
        d_o = d_o.contiguous()
        Lq, B, C = q.shape
        Lk = k.shape[0]
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = _workspace(B, ctx.nheads, Lq, Lk, q.device)
        with torch.cuda.device(q.device):
            rc = _lib.load().pd_attn_bwd_d32(q.data_ptr(), k.data_ptr(), v.data_ptr(), m8.data_ptr() if m8 is not None else None,
                                             o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                             ws.data_ptr(), B, ctx.nheads, Lq, Lk, ctx.scale, _DT[q.dtype], _stream())
        _lib.check(rc)
        return dq, dk, dv, None, None, None


def masked_attention_d32(q, k, v, mask, nheads, scale=None):
    return MaskedAttention32.apply(q, k, v, mask, nheads, (32 ** -0.5) if scale is None else scale)
