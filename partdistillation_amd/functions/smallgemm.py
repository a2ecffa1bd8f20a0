"""ctypes faces of the skinny-activation bf16 GEMMs (include/pd_smallgemm.h).  Raw functions, no autograd: building
blocks of the hand-written decoder backward (functions/decoder_core.py).  GPU only; there is no fallback."""
import ctypes

import torch

from .. import lib as _lib


def _stream():
    return _lib.current_stream()


def _chk2d(*ts):
    for t in ts:
        assert t.dtype == torch.bfloat16 and t.dim() == 2 and t.stride(1) == 1, (t.dtype, t.shape, t.stride())


def supported(K_tn=None, N_nn=None):
    return (K_tn is None or K_tn % 64 == 0) and (N_nn is None or N_nn % 64 == 0)


SPLIT = __import__("os").environ.get("PD_SGEMM_SPLIT", "1") != "0"   # long contractions (>= 512, % 256 == 0) over few rows as 256-wide slices
_SPLIT_WS = {}


def _split_ws(dev, floats, tickets):
    """fp32 partial-tile workspace + zeroed ticket counters per device and stream (launches on one stream run in order)"""
    from .. import cmdbuf
    if cmdbuf.active() is not None:                          # recorded region: scratch + (zero-filled at every replay) tickets in its arena
        return (torch.empty(max(floats, 1), dtype=torch.float32, device=dev), torch.zeros(max(tickets, 1), dtype=torch.int32, device=dev))
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
    e = _SPLIT_WS.get(key)
    if e is None or e[0].numel() < floats or e[1].numel() < tickets:
        e = _SPLIT_WS[key] = (torch.empty(max(floats, 1 << 20), dtype=torch.float32, device=dev),
                              torch.zeros(max(tickets, 256), dtype=torch.int32, device=dev))
    return e


def linear(x, w, b=None, relu=False, out=None):
    """x [M,K] @ w[N,K].T (+ b) (ReLU) -> [M,N]   (bf16; K % 64 == 0)"""
    _chk2d(x, w)
    M, K = x.shape
    N = w.shape[0]
    y = out if out is not None else torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    if SPLIT and K >= 512 and K % 256 == 0 and M <= 1024:
        L = _lib.load()
        ws, tk = _split_ws(x.device, int(L.pd_sgemm_split_workspace_floats(M, N, K)), int(L.pd_sgemm_split_tickets(M, N)))
        _lib.check(L.pd_sgemm_tn_splitk_bf16(x.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(), ws.data_ptr(),
                                             ws.numel(), tk.data_ptr(), M, N, K, x.stride(0), w.stride(0), y.stride(0), int(relu), _stream()))
        return y
    _lib.check(_lib.load().pd_sgemm_tn_bf16(x.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(),
                                            M, N, K, x.stride(0), w.stride(0), y.stride(0), int(relu), _stream()))
    return y


def bmm_tn(x, w):
    """x [Bt, M, K] @ w [Bt, N, K]^T -> [Bt, M, N] (bf16, K <= 256, % 64 == 0): ONE launch (pd_sgemm_tn_batched_bf16)"""
    assert x.dtype == w.dtype == torch.bfloat16 and x.dim() == 3 and w.dim() == 3 and x.is_contiguous() and w.is_contiguous() and x.shape[0] == w.shape[0]
    Bt, M, K = x.shape
    N = w.shape[1]
    y = torch.empty((Bt, M, N), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.load().pd_sgemm_tn_batched_bf16(x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, K, K, N, Bt, M * K, N * K, M * N, _stream()))
    return y


def bmm_tn_supported(x, w):
    return (x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.dim() == 3 and w.dim() == 3 and x.is_contiguous() and w.is_contiguous()
            and x.shape[2] <= 256 and x.shape[2] % 64 == 0 and w.shape[1] % 4 == 0 and (x.shape[1] * x.shape[2]) % 8 == 0 and (w.shape[1] * w.shape[2]) % 8 == 0
            and (x.shape[1] * w.shape[1]) % 4 == 0)


class _TnDesc(ctypes.Structure):                                     # PdSgemmTnDesc (include/pd_smallgemm.h)
    _fields_ = [("X", ctypes.c_void_p), ("W", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("Y", ctypes.c_void_p)] + \
               [(n, ctypes.c_int32) for n in ("M", "N", "ldx", "ldw", "ldy")]


def linear_multi(problems):
    """[(x [M_i,K], w [N_i,K], b | None)] (<= 4, one K <= 256) -> [y_i] as ONE launch (pd_sgemm_tn_multi_bf16)"""
    K = problems[0][0].shape[1]
    descs = (_TnDesc * len(problems))()
    outs = []
    from .. import cmdbuf
    for d, (x, w, b) in zip(descs, problems):
        _chk2d(x, w)
        assert x.shape[1] == K and w.shape[1] == K
        y = torch.empty((x.shape[0], w.shape[0]), dtype=torch.bfloat16, device=x.device)
        if cmdbuf.active() is not None:                      # the problem table is host memory the replay re-reads
            for t_, nm in ((x, "X"), (w, "W"), (b, "bias")):
                if t_ is not None:
                    cmdbuf.require_stable(t_.data_ptr(), "pd_sgemm_tn_multi_bf16 operand " + nm)
        d.X, d.W, d.bias, d.Y = x.data_ptr(), w.data_ptr(), (b.data_ptr() if b is not None else None), y.data_ptr()
        d.M, d.N, d.ldx, d.ldw, d.ldy = x.shape[0], w.shape[0], x.stride(0), w.stride(0), y.stride(0)
        outs.append(y)
    _lib.check(_lib.load().pd_sgemm_tn_multi_bf16(ctypes.byref(descs), len(problems), K, _stream()))
    return outs


def dgrad(dy, w, relu_ref=None, out=None, accumulate=False):
    """dy [M,N] @ w [N,K] -> [M,K], optionally accumulated into `out` and masked by relu_ref > 0   (N % 64 == 0)"""
    _chk2d(dy, w)
    M, N = dy.shape
    K = w.shape[1]
    dx = out if out is not None else torch.empty((M, K), dtype=torch.bfloat16, device=dy.device)
    assert relu_ref is None or (relu_ref.shape == dx.shape and relu_ref.stride() == dx.stride())
    if SPLIT and N >= 512 and N % 256 == 0 and M <= 1024:
        L = _lib.load()
        ws, tk = _split_ws(dy.device, int(L.pd_sgemm_split_workspace_floats(M, K, N)), int(L.pd_sgemm_split_tickets(M, K)))
        _lib.check(L.pd_sgemm_nn_splitn_bf16(dy.data_ptr(), w.data_ptr(), relu_ref.data_ptr() if relu_ref is not None else None, dx.data_ptr(),
                                             ws.data_ptr(), ws.numel(), tk.data_ptr(), M, N, K, dy.stride(0), w.stride(0), dx.stride(0),
                                             int(accumulate), _stream()))
        return dx
    _lib.check(_lib.load().pd_sgemm_nn_bf16(dy.data_ptr(), w.data_ptr(), relu_ref.data_ptr() if relu_ref is not None else None,
                                            dx.data_ptr(), M, N, K, dy.stride(0), w.stride(0), dx.stride(0), int(accumulate), _stream()))
    return dx


def wgrad(dy, x, out=None, bias_out=None):
    """dy [M,N].T @ x [M,K] -> [N,K] (bf16);  bias_out (fp32 [N], optional) = dy.sum(0)"""
    _chk2d(dy, x)
    M, N = dy.shape
    K = x.shape[1]
    dw = out if out is not None else torch.empty((N, K), dtype=torch.bfloat16, device=dy.device)
    assert dw.stride(1) == 1 and (bias_out is None or (bias_out.dtype == torch.float32 and bias_out.numel() == N))
    _lib.check(_lib.load().pd_sgemm_wgrad_bf16(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), bias_out.data_ptr() if bias_out is not None else None,
                                               M, N, K, dy.stride(0), x.stride(0), dw.stride(0), _stream()))
    return dw


_WGRAD_WS = {}


def wgrad_split(dy, x, want_bias=True):
    """dy [M,N].T @ x [M,K] for many rows (pd_sgemm_wgrad_split_bf16) -> (dW bf16 [N,K], dB fp32 [N] or None)"""
    _chk2d(dy, x)
    M, N = dy.shape
    K = x.shape[1]
    L = _lib.load()
    need = int(L.pd_sgemm_wgrad_split_workspace(M, N, K))
    key = str(dy.device)
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _WGRAD_WS[key] = torch.empty(need, dtype=torch.float32, device=dy.device)
    dw = torch.empty((N, K), dtype=torch.bfloat16, device=dy.device)
    db = torch.empty(N, dtype=torch.float32, device=dy.device) if want_bias else None
    _lib.check(L.pd_sgemm_wgrad_split_bf16(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr() if want_bias else None, ws.data_ptr(),
                                           M, N, K, dy.stride(0), x.stride(0), dw.stride(0), _stream()))
    return dw, db



class _WgradDesc(ctypes.Structure):                                  # PdSgemmWgradDesc (include/pd_smallgemm.h)
    _fields_ = [("dY", ctypes.c_void_p), ("X", ctypes.c_void_p), ("dW", ctypes.c_void_p), ("dB", ctypes.c_void_p)] + \
               [(n, ctypes.c_int32) for n in ("M", "N", "K", "ldy", "ldx", "ldw")]


class WgradQueue:
    """weight gradients of one backward pass collected and run as ONE launch (pd_sgemm_wgrad_grouped_bf16).  add() allocates the
    output and keeps the operands alive; run() launches and forgets them.  Same arithmetic as wgrad()."""
    MAXP = 128
    _ring = None
    _table_dev = {}

    def __init__(self):
        self.items = []

    def add(self, dy, x, out=None, bias_out=None):
        _chk2d(dy, x)
        M, N = dy.shape
        K = x.shape[1]
        dw = out if out is not None else torch.empty((N, K), dtype=torch.bfloat16, device=dy.device)
        assert dw.stride(1) == 1 and (bias_out is None or (bias_out.dtype == torch.float32 and bias_out.numel() == N))
        self.items.append((dy, x, dw, bias_out))
        return dw

    def run(self):
        items, self.items = self.items, []
        if not items:
            return
        L = _lib.load()
        dev = items[0][0].device
        cls = WgradQueue
        for lo in range(0, len(items), cls.MAXP):
            part = items[lo:lo + cls.MAXP]
            descs = (_WgradDesc * len(part))()
            for d, (dy, x, dw, db) in zip(descs, part):
                d.dY, d.X, d.dW, d.dB = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), (db.data_ptr() if db is not None else None)
                d.M, d.N, d.K = dy.shape[0], dy.shape[1], x.shape[1]
                d.ldy, d.ldx, d.ldw = dy.stride(0), x.stride(0), dw.stride(0)
            from .. import cmdbuf
            tbytes = int(L.pd_sgemm_wgrad_grouped_table_bytes(cls.MAXP))
            if cls._ring is None:
                from .fused import PinnedRing
                with cmdbuf.host_ops():
                    cls._ring = PinnedRing(tbytes, torch.uint8, pin=True)
            if cmdbuf.active() is not None:
                for dy, x, dw, db in part:
                    for t_, nm in ((dy, "dY"), (x, "X"), (dw, "dW"), (db, "dB")):
                        if t_ is not None:
                            cmdbuf.require_stable(t_.data_ptr(), "grouped skinny weight gradient operand " + nm)
                tab = torch.empty(tbytes, dtype=torch.uint8, device=dev)
            else:
                tab = cls._table_dev.get(str(dev))
                if tab is None:
                    tab = cls._table_dev[str(dev)] = torch.empty(tbytes, dtype=torch.uint8, device=dev)
            host = cls._ring.acquire()
            with torch.cuda.device(dev):
                rc = L.pd_sgemm_wgrad_grouped_bf16(ctypes.byref(descs), len(part), host.data_ptr(), tab.data_ptr(), _stream())
            cls._ring.release()
            _lib.check(rc)
