"""fp8 Linear for BASELINE config 5 ("fp8 MFMA GEMMs") on the module-by-module path: y = x W^T + b as an MX-fp8 GEMM on own kernels
(include/pd_mx8.h; functions/mx8.py) — e4m3 activations x e4m3 weights forward, e5m2 gradients x e4m3 transposed weights for the input
gradient, fp32 accumulation, 32-element blocks with E8M0 exponents quantised locally (no tensor-wide amax pass).  The weight gradient
stays a bf16 GEMM (pd_wgrad_bf16): it contracts over the TOKEN axis, which would need block-scaled copies of both activations along
that axis.  The fused Swin stage (modeling/backbone/swin_core.py) issues the same kernels with the quantisation of the MLP's hidden
activations riding in the GEMM epilogues; this Function is what a single nn.Linear gets (reference swin.py:34-36, 127-129)."""
import torch
from torch.autograd import Function

from . import igemm, mx8

E4M3, E5M2 = mx8.E4M3, mx8.E5M2


def _bf16_rows(t, cols):
    t = t.reshape(-1, cols)
    t = t if t.dtype == torch.bfloat16 else t.to(torch.bfloat16)
    return t if t.is_contiguous() else t.contiguous()


class Fp8Linear(Function):
    """x [..., K] (bf16 / fp32), W [N, K] (fp32 master weight or its bf16 copy), K % 128 == 0 and N % 128 == 0."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        if not x.is_cuda:
            raise RuntimeError("the fp8 Linear runs on the GPU only (no CPU fallback in partdistillation_amd)")
        shape = x.shape
        x2 = _bf16_rows(x, shape[-1])
        w16 = _bf16_rows(weight.detach(), weight.shape[1])
        b = None if bias is None else (bias.detach() if bias.dtype in (torch.float32, torch.bfloat16) else bias.detach().float())
        y = mx8.linear(mx8.quantize(x2, E4M3), mx8.quantize(w16, E4M3), b)
        ctx.save_for_backward(x2, weight)
        ctx.has_bias, ctx.shape, ctx.xdt = bias is not None, shape, x.dtype
        return y.view(*shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        dy2 = _bf16_rows(dy, dy.shape[-1])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = igemm.transposed([_bf16_rows(weight.detach(), weight.shape[1])])[0]        # [K, N]: the input gradient contracts over N
            dx = mx8.linear(mx8.quantize(dy2, mx8.GRAD_FORMAT), mx8.quantize(wt, E4M3), a_fmt=mx8.GRAD_FORMAT).view(ctx.shape)
            dx = dx if dx.dtype == ctx.xdt else dx.to(ctx.xdt)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if want_db:
            db = torch.zeros(weight.shape[0], dtype=torch.float32, device=dy.device)
        if ctx.needs_input_grad[1]:
            if igemm.wgrad_supported(dy2, x2):
                dw = igemm.wgrad(dy2, x2, None, db, out_dtype=torch.float32 if weight.dtype == torch.float32 else torch.bfloat16)   # db rides along
                want_db = False
            else:
                dw = torch.mm(dy2.t(), x2)
            dw = dw if dw.dtype == weight.dtype else dw.to(weight.dtype)
        if want_db:
            db += dy2.sum(0, dtype=torch.float32)
        return dx, dw, db


def supported(x, weight, min_k):
    """both GEMMs of the Linear — forward (contraction K) and input gradient (contraction N) — fit pd_mx8_gemm"""
    N, K = weight.shape
    return (x.is_cuda and K >= min_k and K % 128 == 0 and N % 128 == 0 and weight.dtype in (torch.float32, torch.bfloat16)
            and x.dtype in (torch.float32, torch.bfloat16))


def linear(x, weight, bias):
    return Fp8Linear.apply(x, weight, bias)
