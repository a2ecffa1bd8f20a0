"""Python face of the MX-fp8 GEMM family (include/pd_mx8.h, csrc/mx8.hip): BASELINE config 5's "fp8 MFMA GEMMs" — the Swin Linears
forward and input gradient on v_mfma_scale_f32_32x32x64_f8f6f4 with OCP Microscaling operands (fp8 elements + one E8M0 exponent byte
per 32 elements of the contraction axis).  GPU only; no fallback, no library GEMM."""
import ctypes

import torch

from .. import lib as _lib
from .igemm import ACT_GELU, ACT_NONE, GATE_GELU, GATE_NONE   # noqa: F401  (same epilogue vocabulary as pd_igemm_bf16)

E4M3, E5M2 = 0, 1
BLOCK = 32
GRAD_FORMAT = int(__import__("os").environ.get("PD_MX8_GRAD_FORMAT", E5M2))   # element format of the gradients entering the input-gradient GEMMs
                                                     # (activations and weights: e4m3)


class PdMx8Gemm(ctypes.Structure):                                   # include/pd_mx8.h
    _fields_ = [(n, ctypes.c_void_p) for n in ("a_q", "a_s", "w_q", "w_s", "bias", "gate", "out", "out_pre", "out_q", "out_s")] + \
               [(n, ctypes.c_int32) for n in ("m", "n", "k", "a_format", "act", "gate_mode", "bias_bf16", "out_format")]


class PdMx8Tensor(ctypes.Structure):                                 # include/pd_mx8.h
    _fields_ = [("x", ctypes.c_void_p), ("q", ctypes.c_void_p), ("s", ctypes.c_void_p), ("numel", ctypes.c_int64)]


def supported(m, n, k):
    """can x [m, k] @ w [n, k]^T run on pd_mx8_gemm"""
    return bool(_lib.load().pd_mx8_gemm_supported(int(m), int(n), int(k)))


def _p(t):
    return t.data_ptr() if t is not None else None


def quantize(x, fmt=E4M3):
    """x [rows, cols] bf16 (unit column stride, cols % 32 == 0) -> (q uint8 [rows, cols], s uint8 [rows, cols / 32])"""
    if not x.is_cuda:
        raise RuntimeError("pd_mx8_quantize_bf16 runs on the GPU only (no CPU fallback in partdistillation_amd)")
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1 and x.shape[1] % BLOCK == 0, (x.dtype, x.shape, x.stride())
    rows, cols = x.shape
    q = torch.empty((rows, cols), dtype=torch.uint8, device=x.device)
    s = torch.empty((rows, cols // BLOCK), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.load().pd_mx8_quantize_bf16(x.data_ptr(), rows, cols, x.stride(0), fmt, q.data_ptr(), s.data_ptr(), _lib.current_stream()))
    return q, s


_QT = {}


def quantize_grouped(tensors, fmt=E4M3):
    """contiguous bf16 [n_i, k_i] tensors (k_i % 32 == 0: the weights of a stage) -> [(q_i [n_i, k_i], s_i [n_i, k_i / 32])], ONE launch into
    two fresh flat buffers; the staging of the descriptor table is cached per list of addresses"""
    from .fused import PinnedRing
    from .. import cmdbuf
    L = _lib.load()
    dev = tensors[0].device
    key = tuple((t.data_ptr(), tuple(t.shape)) for t in tensors)       # (address and shape: a re-used address may hold another model's layer)
    hit = _QT.get(key)
    if hit is None:
        offs, total = [], 0
        for t in tensors:
            assert t.dtype == torch.bfloat16 and t.is_contiguous() and t.dim() == 2 and t.shape[1] % BLOCK == 0, (t.dtype, t.shape)
            offs.append(total)
            total += (t.numel() + 127) // 128 * 128
        tb = int(L.pd_mx8_quantize_table_bytes(len(tensors)))
        hit = _QT[key] = (offs, total, PinnedRing(tb, torch.uint8, pin=True), torch.empty(tb, dtype=torch.uint8, device=dev))
        if len(_QT) > 64:
            _QT.pop(next(iter(_QT)))
    offs, total, ring, tdev = hit
    if cmdbuf.active() is not None:                          # recorded region: its own device table (arena); the sources must not move
        tdev = torch.empty(tdev.numel(), dtype=torch.uint8, device=dev)
        for t in tensors:
            cmdbuf.require_stable(t.data_ptr(), "weight to quantise")
    qbuf = torch.empty(total, dtype=torch.uint8, device=dev)
    sbuf = torch.empty(total // BLOCK, dtype=torch.uint8, device=dev)
    descs = (PdMx8Tensor * len(tensors))()
    for d, t, o in zip(descs, tensors, offs):
        d.x, d.q, d.s, d.numel = t.data_ptr(), qbuf.data_ptr() + o, sbuf.data_ptr() + o // BLOCK, t.numel()
    host = ring.acquire()
    rc = L.pd_mx8_quantize_grouped(descs, len(tensors), fmt, host.data_ptr(), tdev.data_ptr(), _lib.current_stream())
    ring.release()
    _lib.check(rc)
    return [(qbuf[o:o + t.numel()].view(t.shape), sbuf[o // BLOCK:(o + t.numel()) // BLOCK].view(t.shape[0], t.shape[1] // BLOCK))
            for t, o in zip(tensors, offs)]


def linear(a, w, bias=None, act=ACT_NONE, gate=None, gate_mode=GATE_NONE, want_pre=False, a_fmt=E4M3, out_mx=None):
    """a = (q [M, K], s [M, K / 32]) of a_fmt, w = (q [N, K], s [N, K / 32]) e4m3
    -> act(a w^T + bias) * gelu'(gate)  [M, N] bf16 (, the pre-activation bf16) (, the result again as MX fp8 of format out_mx along N)"""
    aq, asc = a
    wq, wsc = w
    M, K = aq.shape
    N = wq.shape[0]
    assert aq.dtype == wq.dtype == asc.dtype == wsc.dtype == torch.uint8 and wq.shape[1] == K and aq.is_contiguous() and wq.is_contiguous() \
        and asc.is_contiguous() and wsc.is_contiguous() and asc.shape == (M, K // BLOCK) and wsc.shape == (N, K // BLOCK)
    dev = aq.device
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    pre = torch.empty((M, N), dtype=torch.bfloat16, device=dev) if want_pre else None
    oq = torch.empty((M, N), dtype=torch.uint8, device=dev) if out_mx is not None else None
    osc = torch.empty((M, N // BLOCK), dtype=torch.uint8, device=dev) if out_mx is not None else None
    d = PdMx8Gemm(_p(aq), _p(asc), _p(wq), _p(wsc), _p(bias), _p(gate), _p(out), _p(pre), _p(oq), _p(osc), M, N, K, a_fmt, act,
                  gate_mode if gate is not None else GATE_NONE, int(bias is not None and bias.dtype == torch.bfloat16), out_mx if out_mx is not None else E4M3)
    from .. import cmdbuf
    if cmdbuf.active() is not None:                          # the problem struct is host memory the replay re-reads
        for t_, nm in ((aq, "a_q"), (asc, "a_s"), (wq, "w_q"), (wsc, "w_s"), (bias, "bias"), (gate, "gate"), (out, "out"), (pre, "out_pre"), (oq, "out_q"), (osc, "out_s")):
            if t_ is not None:
                cmdbuf.require_stable(t_.data_ptr(), "pd_mx8_gemm operand " + nm)
    _lib.check(_lib.load().pd_mx8_gemm(ctypes.byref(d), _lib.current_stream()))
    res = [out]
    if want_pre:
        res.append(pre)
    if out_mx is not None:
        res.append((oq, osc))
    return res[0] if len(res) == 1 else tuple(res)
