"""ORACLE — test infrastructure only.  Plain-PyTorch restatement of the MX-fp8 operand format of include/pd_mx8.h (BASELINE config 5:
"fp8 MFMA GEMMs") and of the GEMM on such operands.

The reference repository has no fp8 path (its Swin Linears are nn.Linear under AMP, part_distillation/modeling/backbone/swin.py:34-36,
127-129), so there is no reference code to pin against: "parity unpinned" for the FORMAT.  What pins this file instead is the published
format — OCP Microscaling Formats (MX) v1.0: 32-element blocks, one E8M0 shared exponent (value 2^(e - 127)), fp8 e4m3 / e5m2 elements —
torch's own OCP casts (torch.float8_e4m3fn / float8_e5m2, round to nearest even) for the element conversion, and the round-trip / known-answer
tests of tests/test_oracle.py::test_mx8_*.  One stated deviation from the OCP recipe: the shared exponent is the SMALLEST X with
amax * 2^-X <= format maximum (OCP: floor(log2 amax) - emax, which clips elements whose scaled magnitude exceeds the maximum); X is
clamped to [-126, 126].  The network-level contract (config 5's losses vs the bf16 run) is asserted in tests/test_product_gpu.py."""
import torch

E4M3, E5M2 = 0, 1
BLOCK = 32
_DT = {E4M3: torch.float8_e4m3fn, E5M2: torch.float8_e5m2}
_EMAX = {E4M3: 8, E5M2: 15}
_FMAX = {E4M3: 448.0, E5M2: 57344.0}


def quantize(x, fmt=E4M3):
    """x [..., K] (any float dtype; K % 32 == 0) -> (q uint8 [..., K], s uint8 [..., K / 32])"""
    xf = x.detach().float()
    K = xf.shape[-1]
    assert K % BLOCK == 0
    xb = xf.reshape(*xf.shape[:-1], K // BLOCK, BLOCK)
    amax = torch.nan_to_num(xb.abs(), nan=0.0).amax(-1)                         # a NaN does not poison the block's scale
    bits = amax.contiguous().view(torch.int32)
    X = ((bits >> 23) & 255) - 127 - _EMAX[fmt] + ((bits & 0x7FFFFF) > 0x600000).to(torch.int32)
    X = X.clamp(-126, 126)
    scaled = torch.ldexp(xb, (-X).unsqueeze(-1)).clamp(-_FMAX[fmt], _FMAX[fmt])
    q = scaled.to(_DT[fmt]).view(torch.uint8).reshape(xf.shape)
    return q, (X + 127).to(torch.uint8)


def dequantize(q, s, fmt=E4M3):
    """-> fp32 [..., K]"""
    K = q.shape[-1]
    v = q.view(_DT[fmt]).float().reshape(*q.shape[:-1], K // BLOCK, BLOCK)
    return torch.ldexp(v, (s.to(torch.int32) - 127).unsqueeze(-1)).reshape(q.shape)


def gemm(a, w, a_fmt=E4M3, bias=None):
    """a = (q [M, K], s), w = (q [N, K], s) e4m3 -> fp32 [M, N] = dequant(a) dequant(w)^T (+ bias), accumulated in fp64 and rounded once"""
    out = dequantize(*a, a_fmt).double() @ dequantize(*w, E4M3).double().t()
    if bias is not None:
        out = out + bias.double()
    return out.float()
