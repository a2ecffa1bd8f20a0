"""Python face of the bf16 NHWC backbone convolutions (include/pd_conv.h, csrc/conv_bf16.hip): one launch = convolution +
frozen-BN affine + residual add + ReLU; the input gradient = transposed convolution + the gradient arriving over the
shortcut.  GPU only; no fallback."""
import contextlib
import ctypes
import weakref

import os

import torch
from torch.autograd import Function

from .. import lib as _lib


def _stream():
    return _lib.current_stream()


def supported(x, weight, stride, padding, dilation=1, groups=1):
    co, ci, kh, kw = weight.shape
    return (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and kh == kw and groups == 1 and dilation == 1
            and bool(_lib.load().pd_conv_bf16_supported(ci, co, kh, stride, padding)))


def _nhwc(t):
    return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)


def conv_fwd(x, w, scale=None, bias=None, residual=None, relu=False, stride=1, pad=0):
    """x [B,ci,H,W], w [co,ci,k,k], residual [B,co,Ho,Wo]: bf16, channels_last storage; scale / bias fp32 [co] -> y (channels_last)"""
    B, ci, H, W = x.shape
    co, _, k, _ = w.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty((B, co, Ho, Wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = _lib.load().pd_conv_bf16_fwd(x.data_ptr(), w.data_ptr(), scale.data_ptr() if scale is not None else None,
                                          bias.data_ptr() if bias is not None else None,
                                          residual.data_ptr() if residual is not None else None, y.data_ptr(),
                                          B, H, W, ci, Ho, Wo, co, k, stride, pad, int(relu), _stream())
    _lib.check(rc)
    return y


def conv_dgrad(dz, wt, x_shape, k, stride=1, pad=0, addend=None):
    """dz [B,co,Ho,Wo] (channels_last), wt = filter stored [ci][k][k][co] -> dx [B,ci,H,W] (+ addend)"""
    B, ci, H, W = x_shape
    co, Ho, Wo = dz.shape[1], dz.shape[2], dz.shape[3]
    dx = torch.empty((B, ci, H, W), dtype=torch.bfloat16, device=dz.device, memory_format=torch.channels_last)
    with torch.cuda.device(dz.device):
        rc = _lib.load().pd_conv_bf16_dgrad(dz.data_ptr(), wt.data_ptr(), addend.data_ptr() if addend is not None else None, dx.data_ptr(),
                                            B, H, W, ci, Ho, Wo, co, k, stride, pad, _stream())
    _lib.check(rc)
    return dx


def transposed_filter(w):
    """[co,ci,k,k] (channels_last storage = [co][k][k][ci]) -> a tensor whose storage is [ci][k][k][co]"""
    return w.permute(1, 2, 3, 0).contiguous()


_WS = {}


def conv_wgrad(dz, x, k, stride=1, pad=0, like=None):
    """dz [B,co,Ho,Wo], x [B,ci,H,W] (bf16, channels_last) -> dw [co,ci,k,k] with channels_last strides (storage [co][k][k][ci])"""
    B, ci, H, W = x.shape
    co, Ho, Wo = dz.shape[1], dz.shape[2], dz.shape[3]
    L = _lib.load()
    need = int(L.pd_conv_bf16_wgrad_workspace_floats(B, Ho, Wo, ci, co, k))
    ws = _WS.get(str(x.device))
    if ws is None or ws.numel() < need:
        ws = _WS[str(x.device)] = torch.empty(max(need, 1 << 24), dtype=torch.float32, device=x.device)
    dwk = torch.empty((co, k, k, ci), dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        rc = L.pd_conv_bf16_wgrad(dz.data_ptr(), x.data_ptr(), dwk.data_ptr(), ws.data_ptr(), ws.numel(), B, H, W, ci, Ho, Wo, co, k, stride, pad,
                                  _stream())
    _lib.check(rc)
    dw = dwk.permute(0, 3, 1, 2)
    if like is not None and dw.stride() != like.stride():            # size-1 dimensions (1 x 1 filters): same storage, the filter's strides
        dw = dw.as_strided(like.shape, like.stride())
    return dw


class _Desc(ctypes.Structure):                                      # PdConvWgradDesc (include/pd_conv.h)
    _fields_ = [("dz", ctypes.c_void_p), ("x", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("db", ctypes.c_void_p)] + \
               [(n, ctypes.c_int32) for n in ("batch", "hi", "wi", "ci", "ho", "wo", "co", "k", "stride", "pad")] + [("scale", ctypes.c_void_p)]


class _Deferred:
    """filter gradients queued by Conv2dOwnWgrad.backward while `deferred_wgrads()` is active: ((dz, x) kept alive, their addresses, the address of dw, geometry)"""
    active = False
    queue = []
    adopt = []                                               # (weakref(filter), address handed to autograd): verify_adopted()
    queued = set()                                           # id(filter) of every filter with a gradient in the queue: a second use computes now
    ring = None
    table_dev = None
    MAXP = 256


@contextlib.contextmanager
def deferred_wgrads():
    """Inside this context the filter gradients of Conv2dOwnWgrad are NOT computed when autograd asks for them: backward() hands
    autograd an allocated-but-unwritten tensor and queues the problem; flush() — called on exit, and by the data-parallel reducer
    before it reads a bucket's gradients — runs everything queued as ONE grouped launch (pd_conv_bf16_wgrad_grouped).  Only for
    callers that do not look at .grad of the convolution filters before the flush (engine/trainer.py)."""
    prev = _Deferred.active
    _Deferred.active = True
    ok = False
    try:
        yield
        ok = True
    finally:
        _Deferred.active = prev
        if ok:
            flush()
            verify_adopted()
        else:                                                # backward itself failed: do not replace its exception with ours
            _Deferred.queue, _Deferred.adopt = [], []
            _Deferred.queued.clear()


def verify_adopted():
    """every unwritten tensor Conv2dOwnWgrad.backward handed to autograd must have become its filter's .grad (the grouped launch wrote
    through the raw address): if AccumulateGrad cloned it or added it into an existing .grad — gradient accumulation without
    zero_grad, a tensor hook on the weight, retain_graph — the filter would silently train on uninitialised memory.  Checked after
    the backward pass; backward() avoids the known cases by computing such gradients immediately."""
    pending, _Deferred.adopt = _Deferred.adopt, []
    for wref, ptr in pending:
        w = wref()
        if w is None:
            continue
        if w.grad is None or w.grad.data_ptr() != ptr:
            raise RuntimeError("deferred filter gradient was not adopted as .grad of its weight (shape %s): the grouped launch wrote "
                               "into an orphaned buffer — run this backward outside deferred_wgrads()" % (tuple(w.shape),))


def rows_entry(dy, x, dw, bias_acc=None):
    """queue entry for dw[N,K] = dy[M,N]^T x[M,K] (bf16 row-major, N % 8 == K % 8 == 0, dw rows K apart): a 1 x 1 "convolution" over
    M pixels — the decoder's key / value projection weight gradients (M = 10^3..10^5 memory tokens) ride in the same grouped launch"""
    M, N = dy.shape
    K = x.shape[1]
    assert dy.dtype == x.dtype == dw.dtype == torch.bfloat16 and dy.is_contiguous() and x.is_contiguous() and dw.stride(0) == K
    if bias_acc is not None:                                  # fp32 [N] accumulator: += dy.sum(0) in the same pass
        assert bias_acc.dtype == torch.float32 and bias_acc.numel() == N
        return ((dy, x, bias_acc), dy.data_ptr(), x.data_ptr(), dw.data_ptr(), (1, M, 1, K, M, 1, N, 1, 1, 0), bias_acc.data_ptr())
    return ((dy, x), dy.data_ptr(), x.data_ptr(), dw.data_ptr(), (1, M, 1, K, M, 1, N, 1, 1, 0))


def submit(entries):
    """run these filter-gradient problems: with the deferred queue active they join it, otherwise one grouped launch now"""
    if not entries:
        return
    if _Deferred.active:
        _Deferred.queue.extend(entries)
    else:
        _run(entries)


def run_now(entries):
    """one grouped launch for these problems, whatever the deferred queue's state"""
    _run(list(entries))


def flush():
    q, _Deferred.queue = _Deferred.queue, []
    _Deferred.queued.clear()
    _run(q)


def _run(q):
    if not q:
        return
    dev = q[0][0][0].device
    L = _lib.load()
    for lo in range(0, len(q), _Deferred.MAXP):
        part = q[lo:lo + _Deferred.MAXP]
        descs = (_Desc * len(part))()
        for d, entry in zip(descs, part):
            _keep, dzp, xp, dwp, geom = entry[:5]
            d.dz, d.x, d.dw = dzp, xp, dwp
            d.db = entry[5] if len(entry) > 5 else None
            d.scale = entry[6] if len(entry) > 6 else None
            d.batch, d.hi, d.wi, d.ci, d.ho, d.wo, d.co, d.k, d.stride, d.pad = geom
        from .. import cmdbuf
        need = int(L.pd_conv_bf16_wgrad_grouped_workspace_floats(ctypes.byref(descs), len(part)))
        tbytes = int(L.pd_conv_bf16_wgrad_grouped_table_bytes(_Deferred.MAXP))
        if _Deferred.ring is None:
            from .fused import PinnedRing
            with cmdbuf.host_ops():
                _Deferred.ring = PinnedRing(tbytes, torch.uint8, pin=True)
        if cmdbuf.active() is not None:                          # recorded region: scratch + table in its arena, operands must be stable
            for entry in part:
                for ptr, nm in ((entry[1], "dz"), (entry[2], "x"), (entry[3], "dw"), (entry[5] if len(entry) > 5 else None, "db")):
                    if ptr:
                        cmdbuf.require_stable(ptr, "grouped filter gradient operand " + nm)
            ws = torch.empty(max(need, 1), dtype=torch.float32, device=dev)
            table_dev = torch.empty(tbytes, dtype=torch.uint8, device=dev)
        else:
            ws = _WS.get(str(dev))
            if ws is None or ws.numel() < need:
                ws = _WS[str(dev)] = torch.empty(max(need, 1 << 24), dtype=torch.float32, device=dev)
            if _Deferred.table_dev is None or _Deferred.table_dev.device != dev:
                _Deferred.table_dev = torch.empty(tbytes, dtype=torch.uint8, device=dev)
            table_dev = _Deferred.table_dev
        host = _Deferred.ring.acquire()
        with torch.cuda.device(dev):
            rc = L.pd_conv_bf16_wgrad_grouped(ctypes.byref(descs), len(part), host.data_ptr(), table_dev.data_ptr(), ws.data_ptr(),
                                              ws.numel(), _stream())
        _Deferred.ring.release()
        _lib.check(rc)


GEMM_1X1_FWD = os.environ.get("PD_GEMM_1X1_FWD", "0") != "0"
GEMM_1X1_DGRAD = os.environ.get("PD_GEMM_1X1_DGRAD", "1") != "0"


def _is_plain_1x1(x, weight, stride, padding):
    return (weight.shape[2] == 1 and weight.shape[3] == 1 and stride == 1 and padding == 0 and x.is_contiguous(memory_format=torch.channels_last)
            and weight.is_contiguous())


class Conv2dOwnWgrad(Function):
    """bias-free bf16 NHWC convolution whose FILTER gradient is pd_conv_bf16_wgrad (forward and input gradient: the library's).
    Measured on the 52 such convolutions of R50 at 2 x 1024^2 (tools/bench_r50_convs.py): filter gradients 1.55 ms against
    MIOpen's 2.16; the hand-written forward / input-gradient kernels of this file are still behind MIOpen's and stay off."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding):
        ctx.save_for_backward(x, weight)
        ctx.stride, ctx.padding = stride, padding
        if GEMM_1X1_FWD and _is_plain_1x1(x, weight, stride, padding):
            B, ci, H, W = x.shape
            y = torch.mm(x.permute(0, 2, 3, 1).reshape(-1, ci), weight.view(weight.shape[0], ci).t())
            return y.view(B, H, W, weight.shape[0]).permute(0, 3, 1, 2)
        return torch.ops.aten.convolution(x, weight, None, [stride, stride], [padding, padding], [1, 1], False, [0, 0], 1)

    @staticmethod
    def backward(ctx, dz):
        x, weight = ctx.saved_tensors
        s, p = ctx.stride, ctx.padding
        dz = _nhwc(dz)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if GEMM_1X1_DGRAD and _is_plain_1x1(x, weight, s, p):
                # a 1 x 1, stride-1 convolution on NHWC rows is a plain GEMM: dX[M, Ci] = dZ[M, Co] W[Co, Ci].  The library's GEMM needs no
                # zero-filled output (MIOpen's split-K input-gradient kernels do: a SubTensorOpWithScalar1d launch each)
                B, ci, H, W = x.shape
                dx = torch.mm(dz.permute(0, 2, 3, 1).reshape(-1, dz.shape[1]), weight.view(weight.shape[0], ci))
                dx = dx.view(B, H, W, ci).permute(0, 3, 1, 2)
            else:
                dx = torch.ops.aten.convolution_backward(dz, x, weight, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            # deferred only when AccumulateGrad will ADOPT the tensor we return: no existing .grad to add into, no tensor hooks on the
            # weight, a leaf parameter; anything else gets its gradient computed now
            # a filter used TWICE in one graph: autograd sums the two returned tensors in place before AccumulateGrad sees them, the
            # pointer check of verify_adopted() would pass and the grouped launch overwrite the sum with one use's gradient
            adoptable = (weight.grad is None and weight.is_leaf and not weight._backward_hooks and torch.is_grad_enabled() is False
                         and id(weight) not in _Deferred.queued)
            if _Deferred.active and id(weight) in _Deferred.queued:
                flush()                                      # the first use's tensor is written BEFORE autograd adds this use's gradient to it
                _Deferred.adopt = [(r, p_) for r, p_ in _Deferred.adopt if r() is not weight]   # the sum need not live at that address
            if _Deferred.active and adoptable:
                _Deferred.queued.add(id(weight))
                dw = torch.empty_strided(weight.shape, weight.stride(), dtype=torch.bfloat16, device=x.device)   # written by flush()
                # only the ADDRESS is kept: with a second reference alive autograd's AccumulateGrad would not adopt this tensor
                # as .grad but clone it (unwritten) — the adopted tensor keeps the storage alive until the flush
                B, ci, H, W = x.shape
                _Deferred.queue.append(((dz, x), dz.data_ptr(), x.data_ptr(), dw.data_ptr(),
                                        (B, H, W, ci, dz.shape[2], dz.shape[3], dz.shape[1], weight.shape[2], s, p)))
                _Deferred.adopt.append((weakref.ref(weight), dw.data_ptr()))
            else:
                dw = conv_wgrad(dz, x, weight.shape[2], s, p, like=weight)
        return dx, dw, None, None


def own_wgrad_supported(x, weight, stride, padding, dilation=(1, 1), groups=1):
    co, ci, kh, kw = weight.shape
    return (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and kh == kw and groups == 1
            and tuple(dilation) == (1, 1) and stride[0] == stride[1] and padding[0] == padding[1] and ci % 8 == 0 and co % 8 == 0
            and x.is_contiguous(memory_format=torch.channels_last)
            and (weight.is_contiguous(memory_format=torch.channels_last) or kh == 1))


def conv2d_own_wgrad(x, weight, stride, padding):
    return Conv2dOwnWgrad.apply(x, weight, int(stride[0]), int(padding[0]))
