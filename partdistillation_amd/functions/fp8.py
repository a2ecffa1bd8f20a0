"""fp8 Linear for BASELINE config 5 ("fp8 MFMA GEMMs"): operands quantised by the HIP kernels of include/pd_fp8.h
(per-tensor current scaling, scale computed on the device), the GEMM itself the library's (hipBLASLt through
torch._scaled_mm: e4m3 x e4m3 forward, e5m2 x e4m3 input gradient, fp32 accumulate, bf16 out).  The weight gradient
stays a bf16 GEMM: it contracts over the TOKEN axis, which would need transposed fp8 copies of both activations — at the
Swin shapes (K <= 1536, tokens 10^4..10^5) that costs more HBM traffic than the fp8 GEMM saves."""
import torch
from torch.autograd import Function

from .. import lib as _lib

E4M3, E5M2 = 0, 1
_TORCH_DT = {E4M3: torch.float8_e4m3fn, E5M2: torch.float8_e5m2}
_PD_DT = {torch.float32: _lib.PD_F32, torch.bfloat16: _lib.PD_BF16}


def quantize(x, fmt=E4M3):
    """x fp32 / bf16 (GPU, contiguous, numel % 8 == 0) -> (fp8 tensor of x's shape, scale_inv 0-dim fp32): x ~ q * scale_inv."""
    if not x.is_cuda:
        raise RuntimeError("pd_fp8_quantize runs on the GPU only (no CPU fallback in partdistillation_amd)")
    assert x.is_contiguous() and x.dtype in _PD_DT, (x.dtype, x.shape)
    L = _lib.load()
    stats = torch.zeros(2, dtype=torch.float32, device=x.device)                   # [amax, scale_inv]
    q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    s = _lib.current_stream()
    _lib.check(L.pd_fp8_amax(x.data_ptr(), x.numel(), _PD_DT[x.dtype], stats.data_ptr(), s))
    _lib.check(L.pd_fp8_quantize(x.data_ptr(), x.numel(), _PD_DT[x.dtype], stats.data_ptr(), fmt, q.data_ptr(),
                                 stats.data_ptr() + 4, s))
    return q.view(_TORCH_DT[fmt]), stats[1]


def _mm(a, a_inv, b_nk, b_inv, bias=None):
    """a [M,K] fp8 row-major, b_nk [N,K] fp8 row-major -> bf16 [M,N] = (a b^T) * a_inv * b_inv (+ bias)"""
    return torch._scaled_mm(a, b_nk.t(), scale_a=a_inv, scale_b=b_inv, bias=bias, out_dtype=torch.bfloat16)


class Fp8Linear(Function):
    """y = x W^T + b with x [..., K] (bf16 / fp32), W [N, K] fp32 master weight, K % 16 == N % 16 == rows % 16 == 0."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        if x2.dtype != torch.bfloat16:
            x2 = x2.to(torch.bfloat16)
        x2 = x2.contiguous()
        xq, x_inv = quantize(x2, E4M3)
        wq, w_inv = quantize(weight.detach().contiguous(), E4M3)
        y = _mm(xq, x_inv, wq, w_inv, None if bias is None else bias.detach().to(torch.bfloat16))
        ctx.save_for_backward(x2, weight)
        ctx.has_bias, ctx.shape = bias is not None, shape
        return y.view(*shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.dtype != torch.bfloat16:
            dy2 = dy2.to(torch.bfloat16)
        dy2 = dy2.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            gq, g_inv = quantize(dy2, E5M2)
            wtq, wt_inv = quantize(weight.detach().t().contiguous(), E4M3)        # [K, N]: the weight is small
            dx = _mm(gq, g_inv, wtq, wt_inv).view(ctx.shape)
        if ctx.needs_input_grad[1]:
            dw = torch.mm(dy2.t(), x2).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy2.sum(0, dtype=torch.float32)
        return dx, dw, db


def supported(x, weight, min_k):
    rows = x.numel() // x.shape[-1]
    return (x.is_cuda and weight.shape[1] >= min_k and weight.shape[0] % 16 == 0 and weight.shape[1] % 16 == 0
            and rows % 16 == 0 and weight.dtype in (torch.float32, torch.bfloat16))


def linear(x, weight, bias):
    return Fp8Linear.apply(x, weight, bias)
