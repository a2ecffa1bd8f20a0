"""Autograd face of the HIP multi-scale deformable attention operator.

Mirrors the reference's ``MSDeformAttnFunction`` (reference
ops/functions/ms_deform_attn_func.py:35-52): same ``apply`` signature,
``once_differentiable`` backward, ``None`` grads for shapes / indices / step.
Unlike the reference module (whose import of this Function is commented out,
ops/modules/ms_deform_attn.py:28, so the grid_sample fallback runs even on
GPU), THIS is the path the pixel decoder executes.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ..... import MultiScaleDeformableAttention as MSDA

# optional per-launch timing hook used by bench.py (events on the launch stream)
_TIMING = {"on": False, "fwd": [], "bwd": []}


def enable_timing(on=True):
    _TIMING["on"] = on
    _TIMING["fwd"].clear()
    _TIMING["bwd"].clear()


def timing_ms():
    """-> (list of forward launch ms, list of backward launch ms); call after a sync."""
    return ([a.elapsed_time(b) for a, b in _TIMING["fwd"]], [a.elapsed_time(b) for a, b in _TIMING["bwd"]])


def _timed(kind, fn, *args):
    if not _TIMING["on"]:
        return fn(*args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = fn(*args)
    b.record()
    _TIMING[kind].append((a, b))
    return out


class MSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        ctx.im2col_step = im2col_step
        output = _timed("fwd", MSDA.ms_deform_attn_forward, value, value_spatial_shapes, value_level_start_index,
                        sampling_locations, attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lvl, loc, attn = ctx.saved_tensors
        gv, gl, ga = _timed("bwd", MSDA.ms_deform_attn_backward, value, shapes, lvl, loc, attn,
                            grad_output.contiguous(), ctx.im2col_step)
        return gv, None, None, gl, ga, None


def ms_deform_attn(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, im2col_step=128):
    return MSDeformAttnFunction.apply(value, spatial_shapes, level_start_index, sampling_locations,
                                      attention_weights, im2col_step)
