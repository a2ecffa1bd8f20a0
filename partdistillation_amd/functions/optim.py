"""Python face of the fused clipped-AdamW kernels (include/pd_optim.h)."""
import torch

from .. import lib as _lib

_DT = {torch.float32: _lib.PD_F32, torch.float64: _lib.PD_F64}


def _need_cuda(t):
    if not t.is_cuda:
        raise RuntimeError("partdistillation_amd optimizer kernels run on the GPU only (no CPU fallback)")


def sumsq_accumulate(x, accum):
    """accum (float64 [1], cuda) += sum(x**2); x flat contiguous fp32/fp64."""
    _need_cuda(x)
    with torch.cuda.device(x.device):
        rc = _lib.load().pd_sumsq_accumulate(x.data_ptr(), x.numel(), _DT[x.dtype], accum.data_ptr(),
                                             _lib.current_stream())
    _lib.check(rc)


def adamw_clipped_(param, grad, exp_avg, exp_avg_sq, *, lr, betas, eps, weight_decay, step, grad_sumsq, max_norm, shadow=None, dyn=None):
    _need_cuda(param)
    with torch.cuda.device(param.device):
        if shadow is not None:
            rc = _lib.load().pd_adamw_clipped_shadow(
                param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), shadow.data_ptr(), param.numel(),
                float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step),
                grad_sumsq.data_ptr() if grad_sumsq is not None else None, float(max_norm),
                dyn.data_ptr() if dyn is not None else None, _lib.current_stream())
            _lib.check(rc)
            return
        rc = _lib.load().pd_adamw_clipped(
            param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), param.numel(), _DT[param.dtype],
            float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step),
            grad_sumsq.data_ptr() if grad_sumsq is not None else None, float(max_norm),
            dyn.data_ptr() if dyn is not None else None, _lib.current_stream())
    _lib.check(rc)
