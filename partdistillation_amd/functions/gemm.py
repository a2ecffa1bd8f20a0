"""fp32 nn.Linear on the matrix cores (pd_gemm_tn_f32 / pd_gemm_wgrad_f32, include/pd_gemm.h): forward, input
gradient and weight gradient as hand-written MFMA GEMMs; exact fp32 arithmetic (no TF32 / bf16 rounding)."""
import ctypes

import torch
from torch.autograd import Function

from .. import lib as _lib


def _stream():
    return _lib.current_stream()


def gemm_tn(a, b, bias=None, relu=False):
    """a [M,K], b [N,K] fp32 row-major (last dim contiguous) -> a @ b.T (+bias) (ReLU)."""
    if not a.is_cuda:
        raise RuntimeError("pd_gemm_tn_f32 runs on the GPU only (no CPU fallback in partdistillation_amd)")
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0]
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        rc = _lib.load().pd_gemm_tn_f32(a.data_ptr(), b.data_ptr(), bias.data_ptr() if bias is not None else None,
                                        c.data_ptr(), M, N, K, a.stride(0), b.stride(0), N, int(relu), _stream())
    _lib.check(rc)
    return c


_TIMING = {"on": False, "wgrad": [], "fwd": []}


class _timed_fwd:
    """HIP events around one split-GEMM launch on the launch stream (bench.py's per-launch roofline figures)"""

    def __init__(self, flops, label, nbytes=0.0):
        """flops = 2 M N K (fp32-equivalent); nbytes = the operands and the result once each (algorithmic HBM bytes)"""
        self.flops, self.label, self.nbytes = flops, label, nbytes

    def __enter__(self):
        if _TIMING["on"]:
            self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if _TIMING["on"]:
            self.b.record()
            _TIMING["fwd"].append((self.a, self.b, (self.flops, self.label, self.nbytes)))


def gemm_tn_x3(a, b, bias=None, relu=False):
    """a [M,K], b [N,K] fp32 -> a @ b.T (+bias) (ReLU) with fp32-level accuracy on the bf16 matrix cores
    (pd_gemm_tn_f32x3: exact 3-way bf16 split of every operand, 6 partial products, fp32 accumulation)."""
    if not a.is_cuda:
        raise RuntimeError("pd_gemm_tn_f32x3 runs on the GPU only (no CPU fallback in partdistillation_amd)")
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0]
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    with _timed_fwd(2.0 * M * N * K, "gemm_tn_f32x3(_wide)"):
        _lib.check(_lib.load().pd_gemm_tn_f32x3(a.data_ptr(), b.data_ptr(), bias.data_ptr() if bias is not None else None, c.data_ptr(),
                                                M, N, K, a.stride(0), b.stride(0), N, int(relu), _stream()))
    return c


def row_amax(x):
    """absolute row maxima of an fp32 matrix (pd_row_amax_f32): the scaling input of gemm_tn_h2 for an operand whose producer
    does not emit them"""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().pd_row_amax_f32(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), out.data_ptr(), _stream()))
    return out


def cast_rows_amax(x):
    """x bf16 [rows, cols] contiguous (cols % 8 == 0) -> (x as fp32, absolute row maxima [rows]) in one pass (pd_cast_bf16_f32_amax)"""
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.is_contiguous() and x.shape[1] % 8 == 0
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    am = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().pd_cast_bf16_f32_amax(x.data_ptr(), x.shape[0], x.shape[1], y.data_ptr(), am.data_ptr(), _stream()))
    return y, am


def gemm_tn_h2_bf16out(a, b, bias=None, a_amax=None, b_amax=None):
    """gemm_tn_h2 (mode 0) with the result rounded to bf16 in the epilogue (pd_gemm_tn_f16x2_bf16out)"""
    if not a.is_cuda:
        raise RuntimeError("pd_gemm_tn_f16x2_bf16out runs on the GPU only (no CPU fallback in partdistillation_amd)")
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and a.stride(1) == 1 and b.stride(1) == 1 and b.shape[1] == a.shape[1]
    M, K = a.shape
    N = b.shape[0]
    c = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    p = lambda t: t.data_ptr() if t is not None else None
    _lib.check(_lib.load().pd_gemm_tn_f16x2_bf16out(a.data_ptr(), b.data_ptr(), p(bias), c.data_ptr(), p(a_amax), p(b_amax), M, N, K, a.stride(0),
                                                    b.stride(0), N, _stream()))
    return c


def h2_bits_supported(M, N):
    return N % 256 == 0 and M >= 1024


def gemm_tn_h2(a, b, bias=None, mode=0, bits=None, colsum=None, a_amax=None, b_amax=None, c_amax=None, want_bits=False):
    """a [M,K] @ b [N,K].T (+bias) in fp32 on the fp16 matrix cores: two planes per operand, three products per term, operand rows
    scaled by powers of two from their absolute maxima (pd_gemm_tn_f16x2, include/pd_gemm.h).  mode 0 plain, 1 relu (want_bits: also
    return the sign bits), 2 masked by `bits` with `colsum` += column sums.  a_amax [M] / b_amax [N]: row maxima of a / b (None:
    the operand is O(1)); c_amax [M] zero-filled: receives the row maxima of the result."""
    if not a.is_cuda:
        raise RuntimeError("pd_gemm_tn_f16x2 runs on the GPU only (no CPU fallback in partdistillation_amd)")
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K
    for v, n in ((a_amax, M), (b_amax, N), (c_amax, M)):
        assert v is None or (v.dtype == torch.float32 and v.is_contiguous() and v.numel() == n)
    L = _lib.load()
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    if mode == 1 and want_bits:
        bits = torch.empty(int(L.pd_gemm_tn_f16x2_bits_words(M, N)), dtype=torch.int32, device=a.device)
    p = lambda t: t.data_ptr() if t is not None else None
    args = (a.data_ptr(), b.data_ptr(), p(bias), c.data_ptr(), p(bits), p(colsum), p(a_amax), p(b_amax), p(c_amax), M, N, K, a.stride(0), b.stride(0), N, mode,
            _stream())
    if not _TIMING["on"]:                                  # the hot path: no label formatting, no context manager
        _lib.check(L.pd_gemm_tn_f16x2(*args))
        return (c, bits) if (mode == 1 and want_bits) else c
    which = int(L.pd_gemm_tn_f16x2_which(M, N, K, mode, int(bits is not None), int(a_amax is not None and b_amax is not None)))
    # (one label per kernel instantiation, as rocprofv3 names them: the per-launch averages of bench.py and of the profile agree)
    if which in (4, 5) and c_amax is not None:
        which = 0                                          # (a launch that wants output row maxima takes the tiled kernel)
    label = (f"gemm_rows_f16x2_k256<true, {mode}>" if which == 2 else f"gemm_ra_f16x2_k256<{mode}>" if which == 3
             else "gemm_kres_f16x2<true, 0>" if which == 4 else "gemm_kpc_f16x2<true, 0, false, false>" if which == 5
             else f"gemm_tn_f16x2<{'256, 256, 128, 16' if which == 1 else '128, 128, 64, 16'}, {mode}>")
    with _timed_fwd(2.0 * M * N * K, label, 4.0 * (M * K + N * K + M * N)):
        _lib.check(L.pd_gemm_tn_f16x2(*args))
    return (c, bits) if (mode == 1 and want_bits) else c


def split3(w, transpose=False):
    """fp32 weight [N,K] -> its three bf16 planes [3,N,K] (or [3,K,N] of w.T): pd_split3_bf16, once per step per weight."""
    assert w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and w.stride(1) == 1
    N, K = w.shape
    out = torch.empty((3, K, N) if transpose else (3, N, K), dtype=torch.bfloat16, device=w.device)
    _lib.check(_lib.load().pd_split3_bf16(w.data_ptr(), N, K, w.stride(0), int(transpose), out.data_ptr(), _stream()))
    return out


def pre_supported(M, N, K):
    return N % 256 == 0 and K % 16 == 0 and M >= 1024 and (-(-M // 256)) * (N // 256) >= 128


def gemm_tn_x3_pre(a, planes, bias=None, mode=0, bits=None, colsum=None, want_bits=False):
    """a [M,K] fp32 @ (planes [3,N,K] of a weight).T — pd_gemm_tn_f32x3_pre.  mode 0 plain, 1 relu (want_bits: also return the sign
    bits), 2 masked by `bits` with `colsum` += column sums."""
    assert a.is_cuda and a.dtype == torch.float32 and a.stride(1) == 1 and planes.dtype == torch.bfloat16 and planes.is_contiguous()
    M, K = a.shape
    N = planes.shape[1]
    assert planes.shape[2] == K
    L = _lib.load()
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    if mode == 1 and want_bits:
        bits = torch.empty(int(L.pd_gemm_tn_f32x3_relu_bits_words(M, N)), dtype=torch.int32, device=a.device)
    _lib.check(L.pd_gemm_tn_f32x3_pre(a.data_ptr(), planes.data_ptr(), bias.data_ptr() if bias is not None else None, c.data_ptr(),
                                      bits.data_ptr() if bits is not None else None, colsum.data_ptr() if colsum is not None else None,
                                      M, N, K, a.stride(0), N, mode, _stream()))
    return (c, bits) if (mode == 1 and want_bits) else c


def relu_bits_supported(M, N):
    return N % 256 == 0 and M >= 1024


def gemm_tn_x3_relu_bits(a, b, bias):
    """relu(a @ b.T + bias) and the sign bits of the result in the kernel's accumulator order (pd_gemm_tn_f32x3_relu_bits) for
    gemm_tn_x3_relumask.  -> (c fp32 [M,N], bits int32 [words])"""
    assert a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0]
    L = _lib.load()
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    bits = torch.empty(int(L.pd_gemm_tn_f32x3_relu_bits_words(M, N)), dtype=torch.int32, device=a.device)
    with _timed_fwd(2.0 * M * N * K, "gemm_tn_f32x3_wide<relu + sign bits>"):
        _lib.check(L.pd_gemm_tn_f32x3_relu_bits(a.data_ptr(), b.data_ptr(), bias.data_ptr() if bias is not None else None, c.data_ptr(),
                                                bits.data_ptr(), M, N, K, a.stride(0), b.stride(0), N, _stream()))
    return c, bits


def gemm_tn_x3_relumask(a, b, bits, colsum):
    """(a @ b.T) where the recorded ReLU output was > 0, else 0 -> fp32 [M,N]; colsum[N] += its column sums
    (pd_gemm_tn_f32x3_relumask: the FFN's ReLU backward and first-Linear bias gradient in the GEMM epilogue)."""
    assert a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0]
    assert bits.dtype == torch.int32 and colsum.dtype == torch.float32 and colsum.numel() == N
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    with _timed_fwd(2.0 * M * N * K, "gemm_tn_f32x3_wide<relu mask + column sums>"):
        _lib.check(_lib.load().pd_gemm_tn_f32x3_relumask(a.data_ptr(), b.data_ptr(), bits.data_ptr(), c.data_ptr(), colsum.data_ptr(),
                                                         M, N, K, a.stride(0), b.stride(0), N, _stream()))
    return c


# optional per-launch timing hook used by bench.py (HIP events on the launch stream; (start, stop, flops) per launch)
def enable_timing(on=True):
    _TIMING["on"] = on
    _TIMING["wgrad"].clear()
    _TIMING["fwd"].clear()


def timing(which="wgrad"):
    """-> [(ms, flops)] of the weight-gradient ("wgrad") or forward / input-gradient split-GEMM ("fwd") launches since
    enable_timing(True); call after a synchronize."""
    return [(a.elapsed_time(b), f) for a, b, f in _TIMING[which]]


WGRAD_X3 = True      # weight gradients through the 3-way bf16 split kernel (pd_gemm_wgrad_acc_f32x3_ws): fp32-accurate (error vs fp64 at
                     # the exact kernel's level, tests/test_gemm_gpu.py) and, since its tiles are staged as they lie in memory and
                     # transposed by ds_read_b64_tr_b16 on the way out of LDS, 1.4-1.6x the exact-fp32 MFMA kernel at M = 43 008
                     # (tools/bench_wgrad_x3.py: 1024 x 256: 151 vs 230 us, 256 x 256: 51 vs 78, 256 x 2304: 330 vs 517).
                     # False: the exact-fp32 MFMA kernel (pd_gemm_wgrad_acc_f32).


_WGRAD_WS = {}


def _wgrad_workspace(device, need):
    """one persistent fp32 scratch per device AND stream for the partial tiles of pd_gemm_wgrad_acc_f32x3_ws (<= 34 MB;
    consecutive launches on a stream reuse it in order)"""
    if need <= 0:
        return None
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _WGRAD_WS[key] = torch.empty(max(need, 8912896), dtype=torch.float32, device=device)
    return ws


def gemm_wgrad_acc(dy, x, dw, db=None, x3=None, h2=False, y_amax=None, x_amax=None):
    """dw [N,K] += dy.T @ x, db [N] += dy.sum(0): accumulates into caller-initialised fp32 buffers (no memset launches).
    x3 (default WGRAD_X3): the fp32-accurate kernel on the bf16 matrix cores instead of the exact-fp32 MFMA one.
    h2: the fp16 two-plane form (pd_gemm_wgrad_acc_f16x2_ws) with the operands' absolute row maxima y_amax / x_amax [M]."""
    M, N = dy.shape
    K = x.shape[1]
    assert dw.shape == (N, K) and dw.is_contiguous() and (db is None or db.numel() == N)
    assert dy.dtype == torch.float32 and x.dtype == torch.float32 and dy.stride(1) == 1 and x.stride(1) == 1
    if _TIMING["on"]:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
    L = _lib.load()
    if h2:
        ws = _wgrad_workspace(dy.device, int(L.pd_gemm_wgrad_f16x2_ws_floats(N, K)))
        _lib.check(L.pd_gemm_wgrad_acc_f16x2_ws(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else None,
                                                y_amax.data_ptr() if y_amax is not None else None, x_amax.data_ptr() if x_amax is not None else None,
                                                ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
                                                M, N, K, dy.stride(0), x.stride(0), K, _stream()))
    elif WGRAD_X3 if x3 is None else x3:
        ws = _wgrad_workspace(dy.device, int(L.pd_gemm_wgrad_f32x3_ws_floats(N, K)))
        _lib.check(L.pd_gemm_wgrad_acc_f32x3_ws(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else None,
                                                ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
                                                M, N, K, dy.stride(0), x.stride(0), K, _stream()))
    else:
        _lib.check(L.pd_gemm_wgrad_acc_f32(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else None,
                                           M, N, K, dy.stride(0), x.stride(0), K, _stream()))
    if _TIMING["on"]:
        b.record()
        _TIMING["wgrad"].append((a, b, (2.0 * M * N * K, 4.0 * (M * N + M * K + N * K), "h2" if h2 else "x3")))


class _WgradDesc(ctypes.Structure):                                 # PdGemmWgradDesc (include/pd_gemm.h)
    _fields_ = [("dY", ctypes.c_void_p), ("X", ctypes.c_void_p), ("dW", ctypes.c_void_p), ("dB", ctypes.c_void_p)] + \
               [(n, ctypes.c_int32) for n in ("M", "N", "K", "ldy", "ldx", "ldw")] + [("y_amax", ctypes.c_void_p), ("x_amax", ctypes.c_void_p)]


class WgradQueue:
    """Weight gradients of one backward pass collected and run as ONE grouped launch (pd_gemm_wgrad_f32x3_grouped): nothing
    consumes a weight gradient before the optimizer, so the encoder queues its 30 per step (5 per layer) instead of launching each
    when its operands appear.  The queue keeps the operands alive until flush()."""
    MAXP = 256
    _ring = None
    _table_dev = {}
    _ws = {}

    def __init__(self, h2=False):
        """h2: the fp16 two-plane form (pd_gemm_wgrad_f16x2_grouped); add() then takes the operands' absolute row maxima"""
        self.items = []
        self.h2 = h2

    def add(self, dy, x, dw, db=None, y_amax=None, x_amax=None):
        M, N = dy.shape
        K = x.shape[1]
        assert dw.shape == (N, K) and dw.is_contiguous() and (db is None or db.numel() == N)
        assert dy.dtype == torch.float32 and x.dtype == torch.float32 and dy.stride(1) == 1 and x.stride(1) == 1
        for v in (y_amax, x_amax):
            assert v is None or (self.h2 and v.dtype == torch.float32 and v.is_contiguous() and v.numel() == M)
        self.items.append((dy, x, dw, db, y_amax, x_amax))

    def flush(self):
        items, self.items = self.items, []
        if not items:
            return
        L = _lib.load()
        dev = items[0][0].device
        if self.h2:
            # a grouped launch runs on 256 x 256 tiles only when all its problems suit them: the large outputs (the FFN's) and the
            # rest go as two launches
            wide = [it for it in items if L.pd_gemm_wgrad_f16x2_takes_wide_tiles(it[0].shape[1], it[1].shape[1])]
            rest = [it for it in items if not L.pd_gemm_wgrad_f16x2_takes_wide_tiles(it[0].shape[1], it[1].shape[1])]
            parts = [g[lo:lo + self.MAXP] for g in (wide, rest) for lo in range(0, len(g), self.MAXP)]
        else:
            parts = [items[lo:lo + self.MAXP] for lo in range(0, len(items), self.MAXP)]
        from .. import cmdbuf
        for part in parts:
            descs = (_WgradDesc * len(part))()
            flops = nbytes = 0.0
            for d, (dy, x, dw, db, ya, xa) in zip(descs, part):
                if cmdbuf.active() is not None:
                    for t_, nm in ((dy, "dY"), (x, "X"), (dw, "dW"), (db, "dB"), (ya, "y_amax"), (xa, "x_amax")):
                        if t_ is not None:
                            cmdbuf.require_stable(t_.data_ptr(), "grouped weight gradient operand " + nm)
                d.dY, d.X, d.dW, d.dB = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), (db.data_ptr() if db is not None else None)
                d.y_amax, d.x_amax = (ya.data_ptr() if ya is not None else None), (xa.data_ptr() if xa is not None else None)
                d.M, d.N, d.K, d.ldy, d.ldx, d.ldw = dy.shape[0], dy.shape[1], x.shape[1], dy.stride(0), x.stride(0), dw.shape[1]
                flops += 2.0 * dy.shape[0] * dy.shape[1] * x.shape[1]
                nbytes += 4.0 * (dy.shape[0] * (dy.shape[1] + x.shape[1]) + dy.shape[1] * x.shape[1])
            need = int((L.pd_gemm_wgrad_f16x2_grouped_ws_floats if self.h2 else L.pd_gemm_wgrad_f32x3_grouped_ws_floats)(ctypes.byref(descs), len(part)))
            if need < 0:
                raise RuntimeError("pd_gemm_wgrad_f32x3_grouped: a queued problem violates the alignment rules (N, K, strides % 4, 16-byte bases)")
            tbytes = int(L.pd_gemm_wgrad_f32x3_grouped_table_bytes(self.MAXP))
            if WgradQueue._ring is None:
                from .fused import PinnedRing
                with cmdbuf.host_ops():
                    WgradQueue._ring = PinnedRing(tbytes, torch.uint8, pin=True)
            if cmdbuf.active() is not None:                       # recorded region: scratch + table owned by the recording (its arena)
                ws = torch.empty(max(need, 1), dtype=torch.float32, device=dev)
                tdev = torch.empty(tbytes, dtype=torch.uint8, device=dev)
            else:
                key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
                ws = WgradQueue._ws.get(key)
                if ws is None or ws.numel() < need:
                    ws = WgradQueue._ws[key] = torch.empty(max(need, 1 << 24), dtype=torch.float32, device=dev)
                tdev = WgradQueue._table_dev.get(str(dev))
                if tdev is None:
                    tdev = WgradQueue._table_dev[str(dev)] = torch.empty(tbytes, dtype=torch.uint8, device=dev)
            host = WgradQueue._ring.acquire()
            if _TIMING["on"]:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
            with torch.cuda.device(dev):
                fn = L.pd_gemm_wgrad_f16x2_grouped if self.h2 else L.pd_gemm_wgrad_f32x3_grouped
                rc = fn(ctypes.byref(descs), len(part), host.data_ptr(), tdev.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
            WgradQueue._ring.release()
            _lib.check(rc)
            if _TIMING["on"]:
                b.record()
                _TIMING["wgrad"].append((a, b, (flops, nbytes, "h2 grouped" if self.h2 else "x3 grouped")))


def gemm_wgrad(dy, x, with_bias=False):
    """dy [M,N], x [M,K] -> dy.T @ x  [N,K]  (and dy.sum(0) [N] from the same pass when with_bias)."""
    if not dy.is_cuda:
        raise RuntimeError("pd_gemm_wgrad_f32 runs on the GPU only (no CPU fallback in partdistillation_amd)")
    M, N = dy.shape
    K = x.shape[1]
    if WGRAD_X3 and N % 4 == 0 and K % 4 == 0:
        buf = torch.zeros(N * K + (N if with_bias else 0), dtype=torch.float32, device=dy.device)      # one fill for both
        dw, db = buf[:N * K].view(N, K), (buf[N * K:] if with_bias else None)
        with torch.cuda.device(dy.device):
            gemm_wgrad_acc(dy, x, dw, db, x3=True)
        return (dw, db) if with_bias else dw
    dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    db = torch.empty((N,), dtype=torch.float32, device=dy.device) if with_bias else None
    with torch.cuda.device(dy.device):
        rc = _lib.load().pd_gemm_wgrad_f32(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr() if with_bias else None,
                                           M, N, K, dy.stride(0), x.stride(0), K, _stream())
    _lib.check(rc)
    return (dw, db) if with_bias else dw


# Measured on MI355X (tools/bench_gemm.py, M = 43008 tokens): the library's heuristic is good for the forward and
# input-gradient shapes (100-125 TFLOP/s fp32) but picks ~44 TFLOP/s kernels for the weight gradient (contraction
# over the 43008 tokens); pd_gemm_wgrad_f32 runs those at 105-114 TFLOP/s.  `ALL_MFMA` routes all three through
# the hand-written kernels (used by the tests; 65-97 TFLOP/s on fwd/dgrad today).
ALL_MFMA = False


class LinearF32(Function):
    """y = x W^T + b on [*, K] fp32 inputs; optional fused ReLU (its mask is recovered from y > 0 in backward)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(1) != 1 or x2.stride(0) % 4:
            x2 = x2.contiguous()
        if ALL_MFMA:
            y = gemm_tn(x2, weight, bias, relu)
        else:
            y = torch.nn.functional.linear(x2, weight, bias)
            if relu:
                y = torch.relu_(y)
        ctx.relu = relu
        ctx.save_for_backward(x2, weight, y if relu else None)
        ctx.has_bias = bias is not None
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, weight, y = ctx.saved_tensors
        g2 = gy.reshape(-1, gy.shape[-1])
        if ctx.relu:
            g2 = torch.ops.aten.threshold_backward(g2.contiguous(), y, 0.0)        # one pass: g * [y > 0]
        elif g2.stride(1) != 1 or g2.stride(0) % 4:
            g2 = g2.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = (gemm_tn(g2, weight.t().contiguous()) if ALL_MFMA else g2 @ weight).view(*gy.shape[:-1], weight.shape[1])
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            gw = gemm_wgrad(g2, x2, with_bias=want_b)
            if want_b:
                gw, gb = gw
        elif want_b:
            gb = g2.sum(0)
        return gx, gw, gb, None


def linear_f32(x, weight, bias=None, relu=False):
    return LinearF32.apply(x, weight, bias, relu)
