"""Python faces of the fused bandwidth-bound kernels (include/pd_fused.h)."""
import numpy as np
import torch
from torch.autograd import Function

from . import amax_cache

from .. import lib as _lib


def _stream():
    return _lib.current_stream()


def _nhwc(t):
    return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)


class MaxPool3x3S2(Function):
    """F.max_pool2d(x, 3, stride=2, padding=1) on bf16 channels-last maps (the R50 stem) with hand-written kernels: the forward records the
    window position of the maximum in one byte per element, the backward gathers (pd_maxpool3s2_{fwd,bwd}_bf16)."""

    @staticmethod
    def forward(ctx, x):
        x = _nhwc(x)
        B, C, H, W = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((B, C, OH, OW), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        arg = torch.empty(B * OH * OW * C, dtype=torch.uint8, device=x.device)
        _lib.check(_lib.load().pd_maxpool3s2_fwd_bf16(x.data_ptr(), y.data_ptr(), arg.data_ptr(), B, H, W, C, _stream()))
        ctx.save_for_backward(arg)
        ctx.dims = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        (arg,) = ctx.saved_tensors
        B, C, H, W = ctx.dims
        gy = _nhwc(gy)
        gx = torch.empty((B, C, H, W), dtype=gy.dtype, device=gy.device, memory_format=torch.channels_last)
        _lib.check(_lib.load().pd_maxpool3s2_bwd_bf16(gy.data_ptr(), arg.data_ptr(), gx.data_ptr(), B, H, W, C, _stream()))
        return gx


def max_pool3x3s2_supported(x):
    return x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] % 8 == 0 and x.is_contiguous(memory_format=torch.channels_last)


def max_pool3x3s2(x):
    return MaxPool3x3S2.apply(x)


class AffineAct(Function):
    """y = act(x * scale[c] + bias[c] (+ residual)) on bf16 NCHW-shaped, channels-last-stored tensors; scale/bias are
    constants (frozen BatchNorm), so only x and residual receive gradients."""

    @staticmethod
    def forward(ctx, x, scale, bias, residual, relu):
        x = _nhwc(x)
        res = _nhwc(residual) if residual is not None else None
        y = torch.empty_like(x, memory_format=torch.channels_last)
        with torch.cuda.device(x.device):
            rc = _lib.load().pd_affine_act_fwd_bf16(x.data_ptr(), res.data_ptr() if res is not None else None,
                                                    scale.data_ptr(), bias.data_ptr(), y.data_ptr(), x.numel(), x.shape[1],
                                                    int(relu), _stream())
        _lib.check(rc)
        ctx.relu, ctx.has_res = relu, residual is not None
        ctx.save_for_backward(y if relu else None, scale)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, scale = ctx.saved_tensors
        gy = _nhwc(gy)
        gx = torch.empty_like(gy, memory_format=torch.channels_last)
        gres = torch.empty_like(gy, memory_format=torch.channels_last) if ctx.has_res else None
        with torch.cuda.device(gy.device):
            rc = _lib.load().pd_affine_act_bwd_bf16(gy.data_ptr(), y.data_ptr() if y is not None else None, scale.data_ptr(),
                                                    gx.data_ptr(), gres.data_ptr() if gres is not None else None, gy.numel(),
                                                    gy.shape[1], int(ctx.relu), _stream())
        _lib.check(rc)
        return gx, None, None, gres, None


class AffineActFork(Function):
    """AffineAct whose result is handed out TWICE (the tensor and an alias of it) for an activation with two consumers — a bottleneck
    block's output feeds the next block's first convolution and its shortcut.  Autograd then delivers one gradient per consumer to
    backward(), and pd_affine_act_bwd2_bf16 sums them on the fly instead of autograd launching an add kernel in between."""

    @staticmethod
    def forward(ctx, x, scale, bias, residual, relu):
        x = _nhwc(x)
        res = _nhwc(residual) if residual is not None else None
        y = torch.empty_like(x, memory_format=torch.channels_last)
        with torch.cuda.device(x.device):
            rc = _lib.load().pd_affine_act_fwd_bf16(x.data_ptr(), res.data_ptr() if res is not None else None,
                                                    scale.data_ptr(), bias.data_ptr(), y.data_ptr(), x.numel(), x.shape[1],
                                                    int(relu), _stream())
        _lib.check(rc)
        ctx.relu, ctx.has_res = relu, residual is not None
        ctx.save_for_backward(y if relu else None, scale)
        return y, y.view_as(y)

    @staticmethod
    def backward(ctx, g1, g2):
        y, scale = ctx.saved_tensors
        if g1 is None:
            g1, g2 = g2, None
        g1 = _nhwc(g1)
        g2 = _nhwc(g2) if g2 is not None else None
        gx = torch.empty_like(g1, memory_format=torch.channels_last)
        gres = torch.empty_like(g1, memory_format=torch.channels_last) if ctx.has_res else None
        with torch.cuda.device(g1.device):
            rc = _lib.load().pd_affine_act_bwd2_bf16(g1.data_ptr(), g2.data_ptr() if g2 is not None else None,
                                                     y.data_ptr() if y is not None else None, scale.data_ptr(), gx.data_ptr(),
                                                     gres.data_ptr() if gres is not None else None, g1.numel(), g1.shape[1],
                                                     int(ctx.relu), _stream())
        _lib.check(rc)
        return gx, None, None, gres, None


def affine_act(x, scale, bias, residual=None, relu=True, fork=False):
    """fork=True -> (y, alias of y): give one to each of the two consumers (see AffineActFork)"""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] % 8 == 0):
        raise RuntimeError("pd_affine_act: bf16 CUDA NCHW tensor with channels % 8 == 0 required (no fallback here)")
    fn = AffineActFork if fork else AffineAct
    return fn.apply(x, scale.float().contiguous(), bias.float().contiguous(), residual, relu)


class PinnedRing:
    """Ring of pinned host staging buffers for small host->device tables that are rewritten every step.  A slot is
    rewritten only after the async copy that read it has executed: the host may run steps ahead of the device without a
    later step's table overwriting one still waiting to be copied.  One event per BLOCK of 8 slots (recorded after the
    block's last slot, waited for before its first slot is handed out again a lap later) instead of one per upload:
    the step makes ~45 such uploads and a hipEventRecord costs the host 10-14 us each.

    While a hipGraph is being captured the copy becomes a memcpy NODE that re-reads its host buffer at every replay,
    so a ring slot (recycled by later eager uploads) must never back it: acquire() then hands out a DEDICATED pinned
    buffer that is never written again.  Pinned memory cannot be allocated while a stream is capturing, so those
    buffers are reserved beforehand: every ring counts its acquisitions, TrainStep.capture() measures the count of one
    (eager) rehearsal step and calls reserve_all() with it."""
    _all = []                                               # weak references to every ring (for reserve_all / counters)

    BLOCK = 8

    def __init__(self, shape, dtype, pin, slots=32):
        import weakref
        assert slots % self.BLOCK == 0 and slots >= 2 * self.BLOCK
        self.bufs = [torch.zeros(shape, dtype=dtype, pin_memory=pin) for _ in range(slots)]
        self.events = [None] * (slots // self.BLOCK)
        self.pin, self.i = pin, -1
        self._block_stream = None                            # raw stream the current block's uploads were enqueued on
        self.reserved, self.captured = [], []                # dedicated buffers: waiting for / baked into captured graphs
        self.count = 0                                       # acquisitions so far
        self._in_capture = False
        PinnedRing._all.append(weakref.ref(self))

    @classmethod
    def counters(cls):
        cls._all = [r for r in cls._all if r() is not None]
        return {id(r()): r().count for r in cls._all}

    @classmethod
    def reserve_all(cls, before, after, margin=2):
        """reserve, in every ring, as many dedicated pinned buffers as it handed out between the two counters() snapshots"""
        for r in cls._all:
            ring = r()
            if ring is None or not ring.pin:
                continue
            n = after.get(id(ring), 0) - before.get(id(ring), 0)
            for _ in range(max(0, n + margin - len(ring.reserved)) if n > 0 else 0):
                ring.reserved.append(torch.zeros_like(ring.bufs[0]).pin_memory())

    @classmethod
    def release_captured(cls):
        """drop the dedicated pinned buffers of released graphs (TrainStep.release_graph): nothing replays them any more"""
        for r in cls._all:
            ring = r()
            if ring is not None:
                ring.captured.clear()
                ring.reserved.clear()

    def acquire(self):
        self.count += 1
        from .. import cmdbuf
        rec = cmdbuf.active()
        if rec is not None and self.pin:
            # inside a recorded region (cmdbuf.py) the copy out of this buffer is replayed every step from the SAME address: a
            # dedicated pinned buffer owned by the recording (its content — a launch table — is identical at every replay)
            with cmdbuf.host_ops():
                buf = torch.zeros_like(self.bufs[0]).pin_memory()
            rec.host_static(buf)
            self._in_capture = True
            return buf
        if self.pin and torch.cuda.is_current_stream_capturing():
            if not self.reserved:
                raise RuntimeError("PinnedRing: a host->device upload inside a hipGraph capture needs a reserved pinned buffer "
                                   "(TrainStep.capture() reserves them from a rehearsal step; this upload did not occur in it)")
            buf = self.reserved.pop()
            self.captured.append(buf)
            self._in_capture = True
            return buf
        self._in_capture = False
        self.i = (self.i + 1) % len(self.bufs)
        if self.i % self.BLOCK == 0:
            evs = self.events[self.i // self.BLOCK]
            if evs is not None:
                for ev in evs:
                    ev.synchronize()                          # the copies out of this block's slots, one lap ago, have executed
                self.events[self.i // self.BLOCK] = None
            self._block_stream = None
        return self.bufs[self.i]

    def release(self):
        """call after enqueueing the copy out of the buffer acquire() returned.  One completion event covers a block of BLOCK slots
        as long as their copies were enqueued on ONE stream (the event of the block's last slot is recorded on it); a slot whose copy
        went to another stream (the data-parallel side stream, a warm-up stream) gets an event of its own on that stream."""
        if not self.pin or self._in_capture:
            return
        b = self.i // self.BLOCK
        st = _lib.current_stream()
        if self._block_stream is None:
            self._block_stream = st
        last = self.i % self.BLOCK == self.BLOCK - 1
        if st != self._block_stream or last:
            ev = torch.cuda.Event()
            ev.record()
            if self.events[b] is None:
                self.events[b] = []
            self.events[b].append(ev)
            if st != self._block_stream and not last:
                return
            if st != self._block_stream:                      # the block's last slot on a foreign stream: the home stream's slots need theirs too
                with torch.cuda.stream(torch.cuda.ExternalStream(self._block_stream)):
                    ev2 = torch.cuda.Event()
                    ev2.record()
                self.events[b].append(ev2)


_UPLOAD_RINGS = {}


def upload_small(values, dtype, device):
    """python numbers -> device tensor WITHOUT stalling the host: torch.as_tensor(list, device=cuda) copies from pageable
    memory, which on ROCm waits for everything queued on the stream (measured: 22 ms per step in the part-distillation
    decoder, where it sat behind the backbone forward) and ends the host's run-ahead.  Staged through a ring of pinned
    buffers with an asynchronous copy instead."""
    device = torch.device(device)
    n = len(values)
    if device.type != "cuda":
        return torch.tensor(values, dtype=dtype, device=device)
    cap = 1 << max(4, (max(n, 1) - 1).bit_length())           # rings by power-of-two capacity: table lengths vary from batch to batch
    key = (cap, dtype, str(device))
    ring = _UPLOAD_RINGS.get(key)
    if ring is None:
        ring = _UPLOAD_RINGS[key] = PinnedRing(cap, dtype, pin=True)
    buf = ring.acquire()
    buf[:n] = torch.tensor(values, dtype=dtype)
    out = torch.empty(n, dtype=dtype, device=device)
    out.copy_(buf[:n], non_blocking=True)
    ring.release()
    return out


class GatherPlan:
    """static block table for pd_multi_gather_sumsq over a list of (numel, dst_offset) tensors."""
    CHUNK = 16384

    def __init__(self, numels, dst_offsets, device):
        bt, bs, bd, bl, first = [], [], [], [], []
        for t, (n, off) in enumerate(zip(numels, dst_offsets)):
            first.append(len(bt))
            for s in range(0, max(n, 1), self.CHUNK):
                bt.append(t), bs.append(s), bd.append(off + s), bl.append(min(self.CHUNK, n - s))
        first.append(len(bt))
        self.first_block = first                                   # tensor t owns blocks [first[t], first[t+1])
        self.nblocks, self.ntensors = len(bt), len(numels)
        self.blk_tensor = torch.tensor(bt, dtype=torch.int32, device=device)
        self.blk_start = torch.tensor(bs, dtype=torch.int64, device=device)
        self.blk_dst = torch.tensor(bd, dtype=torch.int64, device=device)
        self.blk_len = torch.tensor(bl, dtype=torch.int32, device=device)
        pin = device.type == "cuda"
        self._host_ptrs = PinnedRing(self.ntensors, torch.int64, pin)
        self._host_bf16 = PinnedRing(self.ntensors, torch.int32, pin)
        self.src_ptrs = torch.zeros(self.ntensors, dtype=torch.int64, device=device)
        self.src_bf16 = torch.zeros(self.ntensors, dtype=torch.int32, device=device)
        self._last = None                                          # host copy of what the device tables hold

    def upload(self, grads, t_begin=0):
        """grads: tensors or None (-> zeros) of tensors [t_begin, t_begin+len(grads)); records their addresses and
        element types for the next gather (pinned staging, async copy of just that slice)."""
        tp, tb = self._host_ptrs.acquire(), self._host_bf16.acquire()
        hp, hb = tp.numpy(), tb.numpy()
        for i, g in enumerate(grads, start=t_begin):
            if g is None:
                hp[i], hb[i] = 0, 0
            else:
                hp[i], hb[i] = g.data_ptr(), 1 if g.dtype == torch.bfloat16 else 0
        t_end = t_begin + len(grads)
        # the gradients of a step mostly sit where the last step's did (arena memory of the recorded regions, the allocator handing the
        # same blocks back): the device table already holds these words then, and the two host -> device copies (a launch each, 8 per
        # step over the four groups) are skipped
        last = self._last
        capturing = self.src_ptrs.is_cuda and torch.cuda.is_current_stream_capturing()
        if last is not None and not capturing and (last[0][t_begin:t_end] == hp[t_begin:t_end]).all() and (last[1][t_begin:t_end] == hb[t_begin:t_end]).all():
            self._host_ptrs.release(), self._host_bf16.release()
            return
        if last is None:
            last = self._last = (hp.copy(), hb.copy())
            last[0][:] = -1
        last[0][t_begin:t_end] = hp[t_begin:t_end]
        last[1][t_begin:t_end] = hb[t_begin:t_end]
        self.src_ptrs[t_begin:t_end].copy_(tp[t_begin:t_end], non_blocking=True)
        self.src_bf16[t_begin:t_end].copy_(tb[t_begin:t_end], non_blocking=True)
        self._host_ptrs.release(), self._host_bf16.release()

    def gather(self, dst, sumsq=None, t_begin=0, t_end=None):
        t_end = self.ntensors if t_end is None else t_end
        b0, b1 = self.first_block[t_begin], self.first_block[t_end]
        with torch.cuda.device(dst.device):
            rc = _lib.load().pd_multi_gather_sumsq(self.src_ptrs.data_ptr(), self.src_bf16.data_ptr(), self.blk_tensor.data_ptr(),
                                                   self.blk_start.data_ptr(), self.blk_dst.data_ptr(), self.blk_len.data_ptr(),
                                                   dst.data_ptr(), sumsq.data_ptr() if sumsq is not None else None, b0, b1,
                                                   _stream())
        _lib.check(rc)


class GroupNormNHWC(Function):
    """GroupNorm (+ optional ReLU) of an fp32 NCHW-shaped, channels-last-stored map, with three streaming HIP kernels
    per direction (pd_nc_sums / pd_nc_affine / pd_nc_affine2) and one O(N*C) coefficient kernel (pd_gn_coeffs_*) between them."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, relu):
        N, C, H, W = x.shape
        P = H * W
        lib = _lib.load()
        st = _stream()
        sums = torch.empty((N, C, 2), dtype=torch.float64, device=x.device)
        coef = torch.empty((5, N, C), dtype=torch.float32, device=x.device)          # a, b, mean_c, rstd_c, xb
        _lib.check(lib.pd_nc_sums_f32(x.data_ptr(), None, None, None, None, sums.data_ptr(), N, P, C, 0, 0, st))
        _lib.check(lib.pd_gn_coeffs_fwd(sums.data_ptr(), weight.data_ptr(), bias.data_ptr(), N, C, groups, P, float(eps),
                                        coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(),
                                        coef[4].data_ptr(), st))
        y = torch.empty_like(x, memory_format=torch.channels_last)
        if C == 256:            # + the pixel maxima the fp16 two-plane convolutions that read y scale their rows with (functions/amax_cache.py)
            am = torch.empty(N * P, dtype=torch.float32, device=x.device)
            _lib.check(lib.pd_nc_affine_amax_f32(x.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(), y.data_ptr(), am.data_ptr(), N, P, C, int(relu), st))
            ctx.y_am = am
        else:
            _lib.check(lib.pd_nc_affine_f32(x.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(), y.data_ptr(), N, P, C, int(relu), st))
            ctx.y_am = None
        ctx.save_for_backward(x, y if relu else None, weight, coef)
        ctx.groups, ctx.relu = groups, relu
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, weight, coef = ctx.saved_tensors
        N, C, H, W = x.shape
        P, G = H * W, ctx.groups
        gy = _nhwc(gy)
        lib = _lib.load()
        st = _stream()
        mean_c, rstd_c, xb = coef[2], coef[3], coef[4]                                # x_hat = x*rstd_c + xb
        sums = torch.empty((N, C, 2), dtype=torch.float64, device=x.device)
        _lib.check(lib.pd_nc_sums_f32(x.data_ptr(), gy.data_ptr(), y.data_ptr() if y is not None else None, rstd_c.data_ptr(),
                                      xb.data_ptr(), sums.data_ptr(), N, P, C, 1, int(ctx.relu), st))
        out = torch.empty((3 * N * C + 2 * C,), dtype=torch.float32, device=x.device)    # a, p, r [N,C] | gw, gb [C]
        a, pc, rc = out[:N * C], out[N * C:2 * N * C], out[2 * N * C:3 * N * C]
        gw, gb = out[3 * N * C:3 * N * C + C], out[3 * N * C + C:]
        _lib.check(lib.pd_gn_coeffs_bwd(sums.data_ptr(), weight.data_ptr(), mean_c.data_ptr(), rstd_c.data_ptr(), N, C, G, P,
                                        a.data_ptr(), pc.data_ptr(), rc.data_ptr(), gw.data_ptr(), gb.data_ptr(), st))
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        if C == 256:
            am = torch.empty(N * P, dtype=torch.float32, device=x.device)
            _lib.check(lib.pd_nc_affine2_amax_f32(gy.data_ptr(), x.data_ptr(), y.data_ptr() if y is not None else None, a.data_ptr(),
                                                  pc.data_ptr(), rc.data_ptr(), dx.data_ptr(), am.data_ptr(), N, P, C, int(ctx.relu), st))
            amax_cache.put(dx, am)
        else:
            _lib.check(lib.pd_nc_affine2_f32(gy.data_ptr(), x.data_ptr(), y.data_ptr() if y is not None else None, a.data_ptr(),
                                             pc.data_ptr(), rc.data_ptr(), dx.data_ptr(), N, P, C, int(ctx.relu), st))
        return dx, gw, gb, None, None, None


def group_norm_nhwc_supported(x, groups):
    C = x.shape[1] if x.dim() == 4 else 0
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and C % 4 == 0 and C <= 256 and 256 % (C // 4) == 0
            and C % groups == 0 and not torch.is_autocast_enabled())


def group_norm_nhwc(x, weight, bias, groups=32, eps=1e-5, relu=False):
    if not group_norm_nhwc_supported(x, groups):
        raise RuntimeError("pd group norm: fp32 CUDA NCHW-shaped tensor with C = 4*2^k <= 256 required (no fallback here)")
    y = GroupNormNHWC.apply(_nhwc(x), weight, bias, groups, eps, relu)
    am = getattr(y.grad_fn, "y_am", None) if y.grad_fn is not None else None
    if am is not None:
        amax_cache.put(y, am)
    return y
