"""Drop-in for the reference's pybind module ``MultiScaleDeformableAttention``
(reference ops/src/vision.cpp:19-22): same two function names, argument order
and return values, backed by libpd_hip.so instead of the CUDA extension.

    import partdistillation_amd.MultiScaleDeformableAttention as MSDA
    out = MSDA.ms_deform_attn_forward(value, shapes, level_start, loc, attn, im2col_step)
    gv, gl, ga = MSDA.ms_deform_attn_backward(value, shapes, level_start, loc, attn, grad_out, im2col_step)

Error behaviour mirrors reference ms_deform_attn_cuda.cu:34-58 and
ms_deform_attn.h:26-45: RuntimeError for non-contiguous or non-GPU tensors and
for ``batch % min(batch, im2col_step) != 0``; float32/float64 only.
"""
import torch

from . import lib as _lib

_DT = {torch.float32: _lib.PD_F32, torch.float64: _lib.PD_F64}


def _check(value, shapes, lvl, loc, attn, extra=()):
    names = ["value", "spatial_shapes", "level_start_index", "sampling_loc", "attn_weight"]
    tensors = [value, shapes, lvl, loc, attn]
    for n, t in extra:
        names.append(n)
        tensors.append(t)
    if not value.is_cuda:
        raise RuntimeError("Not implemented on the CPU")          # reference ms_deform_attn.h:45
    for n, t in zip(names, tensors):
        if not t.is_contiguous():
            raise RuntimeError(f"{n} tensor has to be contiguous")
        if not t.is_cuda:
            raise RuntimeError(f"{n} must be a CUDA tensor")
    if value.dtype not in _DT:
        raise RuntimeError(f'"ms_deform_attn" not implemented for \'{value.dtype}\'')
    for n, t in zip(names, tensors):
        if n in ("spatial_shapes", "level_start_index"):
            if t.dtype != torch.int64:
                raise RuntimeError(f"{n} must be int64")
        elif t.dtype != value.dtype:
            raise RuntimeError(f"{n} must have dtype {value.dtype}, got {t.dtype}")
    if value.dim() != 4 or loc.dim() != 6 or attn.dim() != 5:
        raise RuntimeError("bad ranks: value [N,S,M,D], sampling_loc [N,Lq,M,L,P,2], attn_weight [N,Lq,M,L,P]")
    N, S, M, D = value.shape
    L = shapes.shape[0]
    Lq, P = loc.shape[1], loc.shape[4]
    return N, S, M, D, L, Lq, P


def _im2col_check(batch, im2col_step):
    step = min(batch, int(im2col_step))
    if int(im2col_step) <= 0 or (batch > 0 and batch % step != 0):
        raise RuntimeError(f"batch({batch}) must divide im2col_step({step})")   # reference .cu:58


def _stream():
    return _lib.current_stream()


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    N, S, M, D, L, Lq, P = _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    _im2col_check(N, im2col_step)
    if out.numel() == 0:
        return out
    with torch.cuda.device(value.device):
        rc = _lib.load().pd_msda_forward(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
            attn_weight.data_ptr(), out.data_ptr(), N, S, M, D, L, Lq, P, int(im2col_step), _DT[value.dtype], _stream())
    _lib.check(rc)
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            im2col_step):
    N, S, M, D, L, Lq, P = _check(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                  extra=[("grad_output", grad_output)])
    gv = torch.empty_like(value)
    gl = torch.empty_like(sampling_loc)
    ga = torch.empty_like(attn_weight)
    _im2col_check(N, im2col_step)
    if N * Lq == 0:
        return [gv.zero_(), gl, ga]
    with torch.cuda.device(value.device):
        rc = _lib.load().pd_msda_backward(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
            attn_weight.data_ptr(), grad_output.data_ptr(), gv.data_ptr(), gl.data_ptr(), ga.data_ptr(),
            N, S, M, D, L, Lq, P, int(im2col_step), _DT[value.dtype], _stream())
    _lib.check(rc)
    return [gv, gl, ga]


# ------------------------------------------------------------------------------------------------------------ torch.library
# The two entry points above are also registered as PyTorch custom operators (namespace `pd`) with schemas, fake (meta)
# implementations and an autograd formula, so that torch.compile / torch.export / FakeTensor tracing see a real operator where
# the reference has its pybind functions (ops/src/vision.cpp:19-22):
#     torch.ops.pd.ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step) -> Tensor
#     torch.ops.pd.ms_deform_attn_backward(value, ..., attn_weight, grad_output, im2col_step) -> (Tensor, Tensor, Tensor)
# Only a "cuda" kernel is registered: on any other device the dispatcher raises, like the reference's AT_ERROR("Not implemented
# on the CPU") (ms_deform_attn.h:45) — there is no CPU fallback.  The eager training path keeps calling the functions above
# directly (one Python frame less per launch); `MSDeformAttnFunction` and `torch.ops.pd.ms_deform_attn_forward` are the same kernels.
def _fwd_schema(value: torch.Tensor, spatial_shapes: torch.Tensor, level_start_index: torch.Tensor, sampling_loc: torch.Tensor,
                attn_weight: torch.Tensor, im2col_step: int) -> torch.Tensor:
    return ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)


def _bwd_schema(value: torch.Tensor, spatial_shapes: torch.Tensor, level_start_index: torch.Tensor, sampling_loc: torch.Tensor,
                attn_weight: torch.Tensor, grad_output: torch.Tensor, im2col_step: int) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    gv, gl, ga = ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step)
    return gv, gl, ga


ms_deform_attn_forward_op = torch.library.custom_op("pd::ms_deform_attn_forward", _fwd_schema, mutates_args=(), device_types="cuda")
ms_deform_attn_backward_op = torch.library.custom_op("pd::ms_deform_attn_backward", _bwd_schema, mutates_args=(), device_types="cuda")


@ms_deform_attn_forward_op.register_fake
def _(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    return value.new_empty((value.shape[0], sampling_loc.shape[1], value.shape[2] * value.shape[3]))


@ms_deform_attn_backward_op.register_fake
def _(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    return torch.empty_like(value), torch.empty_like(sampling_loc), torch.empty_like(attn_weight)


def _op_setup_context(ctx, inputs, output):
    value, shapes, lvl, loc, attn, step = inputs
    ctx.save_for_backward(value, shapes, lvl, loc, attn)
    ctx.im2col_step = step


def _op_backward(ctx, grad_output):
    value, shapes, lvl, loc, attn = ctx.saved_tensors
    gv, gl, ga = ms_deform_attn_backward_op(value, shapes, lvl, loc, attn, grad_output.contiguous(), ctx.im2col_step)
    return gv, None, None, gl, ga, None


ms_deform_attn_forward_op.register_autograd(_op_backward, setup_context=_op_setup_context)
