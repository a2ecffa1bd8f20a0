"""Python face of the bf16 implicit-GEMM family (include/pd_igemm.h, csrc/igemm_bf16.hip): convolution / Linear forward and input
gradient with the affine, residual, activation and gate in the epilogue.  GPU only; no fallback."""
import ctypes

import torch

from .. import lib as _lib

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
GATE_NONE, GATE_RELU, GATE_GELU = 0, 1, 2
RES_DENSE, RES_UP2 = 0, 1


class PdIgemm(ctypes.Structure):                                     # include/pd_igemm.h
    _fields_ = [(n, ctypes.c_void_p) for n in ("src", "w", "scale", "bias", "res", "res2", "gate", "out", "out_pre")] + \
               [(n, ctypes.c_int32) for n in ("batch", "hs", "ws", "cs", "ho", "wo", "n", "k", "stride", "pad", "dgrad", "act", "gate_mode", "res_mode", "bias_bf16", "out_col_slab")]


class PdFilterTranspose(ctypes.Structure):                           # include/pd_igemm.h
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("co", ctypes.c_int32), ("taps", ctypes.c_int32),
                ("ci", ctypes.c_int32)]


class PdWgrad(ctypes.Structure):                                     # include/pd_igemm.h
    _fields_ = [("dy", ctypes.c_void_p), ("x", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("db", ctypes.c_void_p), ("row_scale", ctypes.c_void_p)] + \
               [(n, ctypes.c_int32) for n in ("m", "n", "k", "ldy", "ldx", "ldw", "dw_f32")]


_WS = {}


def workspace(dev, nbytes):
    """zero-initialised scratch for the split-K tickets + slabs of one stream (the kernel leaves the tickets zero)"""
    from .. import cmdbuf
    rec = cmdbuf.active()
    if rec is not None:
        # recorded region: ONE scratch in its arena, shared by the region's launches like the per-stream buffer below (stream order keeps
        # them apart); the ticket words at its head are cleared ONCE, now (pd_igemm_bf16 leaves them zero, pd_wgrad_bf16 never touches them)
        ws = getattr(rec, "_ig_scratch", None)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(max(int(nbytes), 40 << 20), dtype=torch.uint8, device=dev)
            _lib.check(_lib.real().pd_memset_async(ws.data_ptr(), 0, 16384, _lib.current_stream()))
            rec._ig_scratch = ws
        return ws
    key = (str(dev), _lib.current_stream())
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _WS[key] = torch.zeros(max(int(nbytes), 64 << 20), dtype=torch.uint8, device=dev)
    return ws


def supported(cs, n, k=1, stride=1, pad=0):
    return bool(_lib.load().pd_igemm_bf16_supported(int(cs), int(n), int(k), int(stride), int(pad)))


def _p(t):
    return t.data_ptr() if t is not None else None


def run(src, w, out, *, batch, hs, ws, cs, ho, wo, n, k=1, stride=1, pad=0, dgrad=False, scale=None, bias=None, res=None, gate=None,
        out_pre=None, res2=None, act=ACT_NONE, gate_mode=GATE_NONE, res_mode=RES_DENSE, out_col_slab=0):
    """raw launch: every tensor is a contiguous bf16 buffer in the layout pd_igemm.h names (scale / bias fp32)"""
    if not src.is_cuda:
        raise RuntimeError("pd_igemm_bf16: CUDA tensors required (partdistillation_amd has no CPU fallback)")
    d = PdIgemm(_p(src), _p(w), _p(scale), _p(bias), _p(res), _p(res2), _p(gate), _p(out), _p(out_pre), batch, hs, ws, cs, ho, wo, n, k, stride, pad,
                int(dgrad), act, gate_mode if gate is not None else GATE_NONE, res_mode, int(bias is not None and bias.dtype == torch.bfloat16),
                int(out_col_slab))
    from .. import cmdbuf
    if cmdbuf.active() is not None:                          # the problem struct is host memory the replay re-reads
        for t_, nm in ((src, "src"), (w, "w"), (scale, "scale"), (bias, "bias"), (res, "res"), (res2, "res2"), (gate, "gate"), (out, "out"), (out_pre, "out_pre")):
            if t_ is not None:
                cmdbuf.require_stable(t_.data_ptr(), "pd_igemm_bf16 operand " + nm)
    L = _lib.load()
    need = int(L.pd_igemm_bf16_workspace_bytes(ctypes.byref(d)))
    if need < 0:
        _lib.check(-1)
    wsb = workspace(src.device, need) if need else None
    _lib.check(L.pd_igemm_bf16(ctypes.byref(d), _p(wsb), wsb.numel() if wsb is not None else 0, _lib.current_stream()))
    return out


def linear(x, w, bias=None, act=ACT_NONE, res=None, gate=None, gate_mode=GATE_NONE, want_pre=False, scale=None, out_col_slab=0):
    """x [M, K] bf16 row-major, w [N, K] bf16 -> act(x w^T * scale + bias + res) * gate'  [M, N] bf16 (and the pre-activation).
    out_col_slab S (a multiple of 128 dividing N; no res / gate / pre): the result is [N / S, M, S] — several Linears over the same rows run as
    ONE product, each one's [M, S] result a dense matrix of its own."""
    M, K = x.shape
    N = w.shape[0]
    assert x.dtype == w.dtype == torch.bfloat16 and x.is_contiguous() and w.is_contiguous() and w.shape[1] == K
    if out_col_slab:
        assert N % out_col_slab == 0 and res is None and gate is None and not want_pre
        out = torch.empty((N // out_col_slab, M, out_col_slab), dtype=torch.bfloat16, device=x.device)
        return run(x, w, out, batch=1, hs=M, ws=1, cs=K, ho=M, wo=1, n=N, scale=scale, bias=bias, act=act, out_col_slab=out_col_slab)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    pre = torch.empty((M, N), dtype=torch.bfloat16, device=x.device) if want_pre else None
    run(x, w, out, batch=1, hs=M, ws=1, cs=K, ho=M, wo=1, n=N, scale=scale, bias=bias, res=res, gate=gate, out_pre=pre, act=act, gate_mode=gate_mode)
    return (out, pre) if want_pre else out


def conv_nhwc(x, w, *, k, stride=1, pad=0, scale=None, bias=None, res=None, act=ACT_NONE, gate=None, gate_mode=GATE_NONE, res_mode=RES_DENSE):
    """x [B, H, W, Ci] bf16 contiguous (NHWC), w [Co, k, k, Ci] bf16 -> [B, Ho, Wo, Co]"""
    B, H, W, Ci = x.shape
    Co = w.shape[0]
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.empty((B, Ho, Wo, Co), dtype=torch.bfloat16, device=x.device)
    return run(x, w, out, batch=B, hs=H, ws=W, cs=Ci, ho=Ho, wo=Wo, n=Co, k=k, stride=stride, pad=pad, scale=scale, bias=bias, res=res, gate=gate,
               act=act, gate_mode=gate_mode, res_mode=res_mode)


def conv_dgrad_nhwc(dz, wt, in_hw, *, k, stride=1, pad=0, res=None, gate=None, gate_mode=GATE_NONE, res_mode=RES_DENSE):
    """dz [B, Ho, Wo, Co] bf16, wt [Ci, k, k, Co] (the transposed filter) -> dx [B, H, W, Ci] (+ res) (* gate mask)"""
    B, Ho, Wo, Co = dz.shape
    Ci = wt.shape[0]
    H, W = in_hw
    out = torch.empty((B, H, W, Ci), dtype=torch.bfloat16, device=dz.device)
    return run(dz, wt, out, batch=B, hs=Ho, ws=Wo, cs=Co, ho=H, wo=W, n=Ci, k=k, stride=stride, pad=pad, dgrad=True, res=res, gate=gate,
               gate_mode=gate_mode, res_mode=res_mode)


_TR = {}


def transposed(weights):
    """[N_i, K_i] bf16 row-major weights -> their [K_i, N_i] transposes (the "weight" operand of the input-gradient GEMMs), all in ONE
    grouped launch into one fresh buffer; the descriptor table is cached per list of weight addresses"""
    from .fused import PinnedRing
    # (address AND shape: the allocator hands a freed model's addresses to the next one, whose layers may be shaped differently — offsets and
    #  buffer size cached under the bare addresses would then be another model's: out-of-bounds transposes)
    key = tuple((w.data_ptr(), tuple(w.shape)) for w in weights)
    L = _lib.load()
    dev = weights[0].device
    hit = _TR.get(key)
    if hit is None:
        descs = (PdFilterTranspose * len(weights))()
        offs, total = [], 0
        for w in weights:
            assert w.dtype == torch.bfloat16 and w.is_contiguous() and w.dim() == 2
            offs.append(total)
            total += (w.numel() + 127) // 128 * 128
        tb = int(L.pd_filter_transpose_table_bytes(len(weights)))
        hit = _TR[key] = (descs, offs, total, PinnedRing(tb, torch.uint8, pin=True), torch.empty(tb, dtype=torch.uint8, device=dev))
        if len(_TR) > 64:
            _TR.pop(next(iter(_TR)))
    descs, offs, total, ring, tdev = hit
    from .. import cmdbuf
    if cmdbuf.active() is not None:                          # recorded region: its own descriptor array + device table (arena)
        descs = (PdFilterTranspose * len(weights))()
        tdev = torch.empty(tdev.numel(), dtype=torch.uint8, device=dev)
        for w in weights:
            cmdbuf.require_stable(w.data_ptr(), "weight to transpose")
    buf = torch.empty(total, dtype=torch.bfloat16, device=dev)
    base = buf.data_ptr()
    for d, w, o in zip(descs, weights, offs):
        d.src, d.dst, d.scale, d.co, d.taps, d.ci = w.data_ptr(), base + 2 * o, None, w.shape[0], 1, w.shape[1]
    host = ring.acquire()
    rc = L.pd_filter_transpose_grouped(descs, len(weights), host.data_ptr(), tdev.data_ptr(), _lib.current_stream())
    ring.release()
    _lib.check(rc)
    return [buf[o:o + w.numel()].view(w.shape[1], w.shape[0]) for w, o in zip(weights, offs)]


class OwnLinear(torch.autograd.Function):
    """y = x w^T (+ b) on pd_igemm_bf16 with own input / weight gradients (nn.Linear under bf16 autocast: F.linear would cast x and run
    the library GEMM).  x [..., K] any float dtype (cast to bf16 like autocast does), w [N, K] bf16, b bf16 / fp32 or None."""

    @staticmethod
    def forward(ctx, x, w, b):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        x2 = x2 if x2.dtype == torch.bfloat16 else x2.to(torch.bfloat16)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        y = linear(x2, w, b)
        ctx.save_for_backward(x2, w)
        ctx.has_b, ctx.shp, ctx.xdt = b is not None, shp, x.dtype
        return y.view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        from . import conv_bf16
        from . import rowwise as rw
        x2, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, w.shape[0])
        dy2 = dy2 if dy2.dtype == torch.bfloat16 else dy2.to(torch.bfloat16)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = linear(dy2, transposed([w])[0]).view(ctx.shp)
            dx = dx if dx.dtype == ctx.xdt else dx.to(ctx.xdt)
        want_db = ctx.has_b and ctx.needs_input_grad[2]
        if want_db:
            db = torch.zeros(w.shape[0], dtype=torch.float32, device=dy.device)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            if wgrad_supported(dy2, x2):
                wgrad(dy2, x2, dw, db)                      # the column sums of dy ride along
                want_db = False
            else:
                conv_bf16.run_now([conv_bf16.rows_entry(dy2, x2, dw)])
        if want_db:
            rw.colsum_acc(dy2, db)
        return dx, dw, db


def own_linear_supported(x, w):
    return (x.is_cuda and w.dtype == torch.bfloat16 and w.is_contiguous() and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0
            and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)


def wgrad_supported(dy, x):
    return (dy.is_cuda and dy.dtype == x.dtype == torch.bfloat16 and dy.dim() == x.dim() == 2 and dy.stride(1) == x.stride(1) == 1 and dy.shape[1] % 8 == 0
            and x.shape[1] % 8 == 0 and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0)


def wgrad_prefers_library(M, N, K):
    """few rows under a large [N, K] result: hundreds of 128 x 128 tiles with a few dozen 64-row stages each — the library's GEMM measured
    40 us against 45 (Swin-B stage 4 fc1, 2 592 x 1 024 -> 4 096) and 115 against 155 (Swin-L), tools/bench_swin_wgrad.py"""
    return N * K >= 3_000_000 and M <= 8192


def wgrad(dy, x, dw=None, db=None, row_scale=None, out_dtype=torch.bfloat16):
    """dy [M, N], x [M, K] bf16 -> dw [N, K] (bf16 or fp32: dw's dtype, else out_dtype) = dy^T x (pd_wgrad_bf16); db (fp32 [N], optional) += dy.sum(0)"""
    M, N = dy.shape
    K = x.shape[1]
    if dw is None:
        dw = torch.empty((N, K), dtype=out_dtype, device=dy.device)
    assert dw.dtype in (torch.bfloat16, torch.float32) and dw.stride(1) == 1 and (db is None or (db.dtype == torch.float32 and db.numel() == N))
    d = PdWgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), _p(db), _p(row_scale), M, N, K, dy.stride(0), x.stride(0), dw.stride(0),
                1 if dw.dtype == torch.float32 else 0)
    from .. import cmdbuf
    if cmdbuf.active() is not None:
        for t_, nm in ((dy, "dy"), (x, "x"), (dw, "dw"), (db, "db"), (row_scale, "row_scale")):
            if t_ is not None:
                cmdbuf.require_stable(t_.data_ptr(), "pd_wgrad_bf16 operand " + nm)
    L = _lib.load()
    need = int(L.pd_wgrad_bf16_workspace_bytes(ctypes.byref(d)))
    if need < 0:
        _lib.check(-1)
    wsb = workspace(dy.device, need) if need else None
    _lib.check(L.pd_wgrad_bf16(ctypes.byref(d), _p(wsb), wsb.numel() if wsb is not None else 0, _lib.current_stream()))
    return dw


WGRAD_SEQ_MAX = 8                                                    # PD_WGRAD_SEQ_MAX of include/pd_igemm.h


def wgrad_seq(items):
    """items: [(dy [M, N], x [M, K], dw [N, K] bf16 / fp32, db fp32 [N] or None)], at most 8 -> the weight gradients of all of them by ONE call
    (pd_wgrad_bf16_seq: the main launches back to back, one launch for all their slice sums); db += dy.sum(0)"""
    assert 0 < len(items) <= WGRAD_SEQ_MAX
    descs = (PdWgrad * len(items))()
    from .. import cmdbuf
    rec = cmdbuf.active() is not None
    for d, (dy, x, dw, db) in zip(descs, items):
        assert wgrad_supported(dy, x) and dw.dtype in (torch.bfloat16, torch.float32) and dw.stride(1) == 1 and (db is None or (db.dtype == torch.float32 and db.numel() == dy.shape[1]))
        d.dy, d.x, d.dw, d.db, d.row_scale = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), _p(db), None
        d.m, d.n, d.k, d.ldy, d.ldx, d.ldw, d.dw_f32 = dy.shape[0], dy.shape[1], x.shape[1], dy.stride(0), x.stride(0), dw.stride(0), int(dw.dtype == torch.float32)
        if rec:
            for t_, nm in ((dy, "dy"), (x, "x"), (dw, "dw"), (db, "db")):
                if t_ is not None:
                    cmdbuf.require_stable(t_.data_ptr(), "pd_wgrad_bf16_seq operand " + nm)
    L = _lib.load()
    need = int(L.pd_wgrad_bf16_seq_workspace_bytes(descs, len(items)))
    if need < 0:
        _lib.check(-1)
    wsb = workspace(items[0][0].device, need) if need else None
    _lib.check(L.pd_wgrad_bf16_seq(descs, len(items), _p(wsb), wsb.numel() if wsb is not None else 0, _lib.current_stream()))
