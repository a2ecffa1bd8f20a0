"""ORACLE — test infrastructure only.

Python face of oracle/msda_ref.c (plain-C multi-scale deformable attention,
forward + backward, f32/f64) plus a pure-torch restatement used to cross-check
it.  Reference: functions/ms_deform_attn_func.py:55-75 (what the reference
runs) == ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304 (what it meant to run).
"""
import ctypes

import numpy as np
import torch
import torch.nn.functional as F

from . import clib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _prep(value, shapes, lvl_start, loc, attn):
    dt = value.dtype
    assert dt in (torch.float32, torch.float64)
    npdt = np.float32 if dt == torch.float32 else np.float64
    v = np.ascontiguousarray(value.detach().cpu().numpy().astype(npdt))
    lo = np.ascontiguousarray(loc.detach().cpu().numpy().astype(npdt))
    at = np.ascontiguousarray(attn.detach().cpu().numpy().astype(npdt))
    sh = np.ascontiguousarray(shapes.detach().cpu().numpy().astype(np.int64))
    ls = np.ascontiguousarray(lvl_start.detach().cpu().numpy().astype(np.int64))
    N, S, M, D = v.shape
    _, Lq, _, L, P, _ = lo.shape
    return v, sh, ls, lo, at, (N, S, M, D, L, Lq, P), ("f32" if dt == torch.float32 else "f64"), npdt


def msda_forward(value, shapes, lvl_start, loc, attn):
    """C oracle forward -> torch tensor [N, Lq, M*D] (CPU)."""
    v, sh, ls, lo, at, dims, suf, npdt = _prep(value, shapes, lvl_start, loc, attn)
    N, S, M, D, L, Lq, P = dims
    out = np.zeros((N, Lq, M * D), dtype=npdt)
    fn = getattr(clib.lib(), f"pd_oracle_msda_forward_{suf}")
    fn(_p(v), _p(sh), _p(ls), _p(lo), _p(at), _p(out), *[ctypes.c_int(x) for x in dims])
    return torch.from_numpy(out)


def msda_backward(value, shapes, lvl_start, loc, attn, grad_out):
    """C oracle backward -> (grad_value, grad_loc, grad_attn) CPU tensors."""
    v, sh, ls, lo, at, dims, suf, npdt = _prep(value, shapes, lvl_start, loc, attn)
    go = np.ascontiguousarray(grad_out.detach().cpu().numpy().astype(npdt))
    gv, gl, ga = np.zeros_like(v), np.zeros_like(lo), np.zeros_like(at)
    fn = getattr(clib.lib(), f"pd_oracle_msda_backward_{suf}")
    fn(_p(v), _p(sh), _p(ls), _p(lo), _p(at), _p(go), _p(gv), _p(gl), _p(ga), *[ctypes.c_int(x) for x in dims])
    return torch.from_numpy(gv), torch.from_numpy(gl), torch.from_numpy(ga)


def msda_torch(value, shapes, loc, attn):
    """Pure-torch restatement via grid_sample (differentiable; the fallback the
    reference actually executes, ms_deform_attn_func.py:55-75)."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    sizes = [int(h) * int(w) for h, w in shapes.tolist()]
    per_level = value.split(sizes, dim=1)
    grids = 2 * loc - 1
    sampled = []
    for lvl, (h, w) in enumerate(shapes.tolist()):
        v = per_level[lvl].reshape(N, sizes[lvl], M * D).transpose(1, 2).reshape(N * M, D, h, w)
        g = grids[:, :, :, lvl].transpose(1, 2).reshape(N * M, Lq, P, 2)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    sampled = torch.stack(sampled, dim=-2).reshape(N * M, D, Lq, L * P)
    a = attn.transpose(1, 2).reshape(N * M, 1, Lq, L * P)
    out = (sampled * a).sum(-1).view(N, M * D, Lq)
    return out.transpose(1, 2).contiguous()
