"""Data-parallel gradient exchange: one process per GPU, the model replicated, gradients averaged
with bucketed all-reduces over RCCL/xGMI that are issued from autograd hooks while backward is still running (the
process group runs them on its own stream; finish() makes the compute stream wait for them).  A bucket is a run of consecutive parameters of one flat group: when the last of its gradients has
been produced, one multi-tensor kernel gathers them into the bucket's contiguous SLICE of the flat gradient buffer
(engine/flat_params.py) and the slice is all-reduced in place — no bucket copy-in / copy-out.

The reference gets this from detectron2's ``create_ddp_model`` (torch DDP over NCCL); the only other collective on
the path is the scalar ``num_masks`` all-reduce of criterion.py:252-254, kept in modeling/criterion.py.
Works with any torch.distributed backend ("nccl" = RCCL on ROCm; "gloo" in the CPU tests)."""
import contextlib
import os
from typing import List

import torch
import torch.distributed as dist

_CHECK_SPARSE_ROWS = bool(int(os.environ.get("PD_DDP_CHECK_SPARSE", "0")))   # assert that a row-sparse group's gradient is zero outside its rows

from ..functions.conv_bf16 import flush as _flush_deferred_wgrads


class _Bucket:
    __slots__ = ("group", "t_begin", "t_end", "start", "end", "pending", "total", "work", "index")

    def __init__(self, group, t_begin, t_end, start, end):
        self.group, self.t_begin, self.t_end, self.start, self.end = group, t_begin, t_end, start, end
        self.total = t_end - t_begin
        self.pending, self.work = self.total, None


class _NoWork:
    def wait(self):
        return True


class _WeakPublish:
    def __init__(self, reducer):
        import weakref
        self.ref = weakref.ref(reducer)

    def __call__(self, p, grad):
        r = self.ref()
        return bool(r is not None and r.publish(p, grad))


class BucketedGradReducer:
    def __init__(self, flat, bucket_mb: float = 25.0, process_group=None, overlap: bool = True, optimizer=None, sparse_rows_cap: int = 256):
        self.flat, self.pg, self.optimizer = flat, process_group, optimizer
        self.sparse_rows_cap = int(sparse_rows_cap)       # rows per rank of the static row-sparse exchange (_exchange_rows); same on every rank
        self._overflow = []
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        from ..utils.misc import collectives_active
        self.active = self.world > 1 or collectives_active()       # (one rank with PD_DDP_FORCE=1: the collectives run with a single participant)
        self.buckets: List[_Bucket] = []
        self._param_bucket = {}
        cap_bytes = int(bucket_mb * 1024 * 1024)
        # row-sparse groups (the float64 part-distillation class head, [N_obj * K + 1, 256]: 16 MB at 1 000 object classes,
        # 360 MB at the shipped 22 000; part_distillation_transformer_decoder.py:107): a step touches only the K + 1 rows of each
        # image's object class, every other row's gradient is exactly zero on every rank.  They get no buckets: finish()
        # exchanges just the touched rows — ~35 KB per rank instead of the dense buffer.
        self.sparse_groups = [gi for gi, g in enumerate(flat.groups)
                              if self.active and all(getattr(p, "_pd_row_sparse", False) for p in g.params)]
        for gi, g in enumerate(flat.groups):
            if gi in self.sparse_groups:
                continue
            cap = max(1, cap_bytes // g.grad.element_size())
            t0 = 0
            for t, (p, off) in enumerate(zip(g.params, g.offsets)):
                end = g.offsets[t + 1] if t + 1 < len(g.params) else g.numel
                if end - g.offsets[t0] >= cap or t + 1 == len(g.params):
                    b = _Bucket(gi, t0, t + 1, g.offsets[t0], end)
                    self.buckets.append(b)
                    for q in g.params[t0:t + 1]:
                        self._param_bucket[q] = b
                    t0 = t + 1
        for i, b in enumerate(self.buckets):
            b.index = i
        self._next = 0                    # collectives are issued in bucket-index order on every rank (see _on_grad)
        self._use_avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        dev = flat.groups[0].grad.device
        # The collectives overlap with backward either way: torch's NCCL (= RCCL) process group runs them on ITS stream and `work.wait()`
        # (finish()) is a stream-level wait.  A private side stream in front of that (rounds 2-3; PD_DDP_SIDE_STREAM=1) doubles the
        # cross-stream events per bucket: measured with the collectives forced on one GPU (PD_DDP_FORCE=1, 25 MB buckets) 24.00 ms
        # per step with it, 23.15 without, 22.40 with no data parallelism at all.
        overlap = overlap and os.environ.get("PD_DDP_SIDE_STREAM", "0") == "1"
        self._side = torch.cuda.Stream(device=dev) if (dev.type == "cuda" and overlap) else None
        self._skip_reduce = os.environ.get("PD_DDP_SKIP_REDUCE", "0") == "1"        # (development: everything but the collective itself)
        self._hooks = []
        self._early = False
        if self.active:
            # the fused ResNet body (one autograd node: its filter gradients would all arrive when its backward returns) hands a stage's
            # gradients over as soon as that stage is done: modeling/backbone/resnet_core.py calls publish(parameter, gradient)
            from ..modeling.backbone import resnet_core
            resnet_core.PUBLISH = _WeakPublish(self)        # (weak: a finished trainer's reducer is not kept alive by the module global)
            self._early = True
            for gi, g in enumerate(flat.groups):
                if gi in self.sparse_groups:
                    continue
                for p in g.params:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # ------------------------------------------------------------------ hooks
    def publish(self, p, grad):
        """a gradient that did NOT come through autograd's AccumulateGrad (a fused node publishing per stage): install it as p.grad and
        count the parameter as complete, exactly as the post-accumulate hook would.  -> False when p is not one of this reducer's
        parameters (the caller then returns the gradient to autograd as usual)"""
        if p not in self._param_bucket:
            return False
        p.grad = grad
        self._on_grad(p)
        return True

    def _on_grad(self, p):
        """a bucket is all-reduced once ITS gradients are complete AND every bucket before it has been issued: the ranks
        then issue the same collectives in the same order even when the set of parameters that received gradients
        differs between them in a step (data-dependent branches: empty targets, the per-head criterion loop) — the
        ordering rule torch DDP enforces.  Buckets held back by an unused parameter go out in finish()."""
        b = self._param_bucket[p]
        b.pending -= 1
        while self._next < len(self.buckets) and self.buckets[self._next].pending <= 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def _launch(self, b):
        g = self.flat.groups[b.group]
        _flush_deferred_wgrads()                                 # filter gradients the backbone queued (functions/conv_bf16.py) are due now
        g.gather(None, b.t_begin, b.t_end)                       # compute stream: p.grad tensors -> flat slice
        buf = g.grad[b.start:b.end]
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream(buf.device))
            with torch.cuda.stream(self._side):
                self._reduce(b, buf)
        else:
            self._reduce(b, buf)

    def _reduce(self, b, buf):
        op = dist.ReduceOp.AVG if self._use_avg else dist.ReduceOp.SUM
        if self._skip_reduce:
            b.work = _NoWork()
            return
        b.work = dist.all_reduce(buf, op=op, group=self.pg, async_op=True)

    def _exchange_rows(self, gi):
        """mean over the ranks of a row-sparse group's gradient from the touched rows only.  Every parameter of the group is
        [rows, ...] and carries `_pd_rows` (LongTensor [R], the rows this rank's step used: images per GPU x (K + 1)).  Equal to
        the dense all-reduce: untouched rows are exact zeros everywhere.
        The exchange has a STATIC shape — `sparse_rows_cap` rows per rank (a config constant, identical on every rank), a rank's
        list padded with the sentinel row -1 whose payload is zero — so the ranks need no agreement round and the host reads
        nothing back: the whole exchange is enqueued on the side stream behind the bucket all-reduces while the GPU is still in
        backward (round 3 agreed on the size with a 16-byte all-reduce and two blocking host reads per step).  A rank whose step
        kept no row record sends the non-zero rows of its local gradient, found on the device (top-`cap` rows by "has a non-zero
        entry"); a rank with no gradient at all sends only sentinels.  More touched rows than the cap is a configuration error
        (TrainStep validates the cap against images per rank x (K + 1) at construction): the rank it happens on — host-known, a record
        longer than the cap, or device-found — sends its first `cap` rows and an OVERFLOW FLAG in an extra element of its row vector;
        every rank reads the gathered flags of all ranks and all of them raise at the same point (their next step's check), instead
        of one rank raising before its collective and the others hanging in theirs."""
        g = self.flat.groups[gi]
        g.gather(None)                                            # local dense gradients -> flat buffer (compute stream)
        rows = getattr(g.params[0], "_pd_rows", None)
        for p in g.params:
            if hasattr(p, "_pd_rows"):
                p._pd_rows = None                                 # consumed: a stale list is never exchanged for a later step
        dev = g.grad.device
        cap = self.sparse_rows_cap
        # (a record longer than the cap is NOT raised here: this rank alone would leave the others blocked in all_gather until the
        # process group's timeout.  The overflow travels with the gathered rows instead — every rank sees every rank's flag — and all
        # ranks raise together at their next _check_overflow(); see _exchange_rows_on_stream)
        self._check_overflow()
        side = self._side
        if side is not None:                                      # the exchange itself runs on the side stream, behind the bucket all-reduces
            side.wait_stream(torch.cuda.current_stream(dev))
        ctx = torch.cuda.stream(side) if side is not None else contextlib.nullcontext()
        with ctx:
            self._exchange_rows_on_stream(g, rows, cap, dev)
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)

    def _check_overflow(self):
        """raise if an EARLIER step's device-side row search found more touched rows than the cap (read without blocking: the flag
        of a step is looked at once its copy to pinned memory has completed)"""
        keep = []
        for flag, ev in self._overflow:
            if ev is None or ev.query():
                if int(flag[0]) != 0:
                    raise RuntimeError(f"row-sparse gradient group: a rank's step touched more than MODEL.AMD.DDP_SPARSE_ROWS_CAP = "
                                       f"{self.sparse_rows_cap} rows; that step's gradient was truncated (raised on every rank)")
            else:
                keep.append((flag, ev))
        self._overflow = keep

    def _exchange_rows_on_stream(self, g, rows, cap, dev):
        views = [g._view(g.grad, p, off).reshape(p.shape[0], -1) for p, off in zip(g.params, g.offsets)]
        over = None
        no_record, beyond = rows is None, None
        if rows is None:
            # no record: the touched rows are the rows with a non-zero entry — found on the device in a fixed-size form
            nz = views[0].ne(0).any(1)
            for v in views[1:]:
                nz |= v.ne(0).any(1)
            k = min(cap, nz.numel())
            val, idx = torch.topk(nz.to(torch.float32), k)
            rows = torch.where(val > 0, idx, idx.new_full((), -1))
            over = (nz.sum() > cap).to(torch.int64).reshape(1)
        rows = rows.reshape(-1).to(dev)
        if over is None:
            over = torch.full((1,), int(rows.numel() > cap), dtype=torch.int64, device=dev)
            beyond = rows[cap:] if rows.numel() > cap else None  # host-known overflow: these rows' local gradient is never exchanged
            rows = rows[:cap]                                     # (flagged: every rank raises at its next check)
        if rows.numel() < cap:                                    # pad to the static size: sentinel rows carry zeros
            rows = torch.cat([rows, rows.new_full((cap - rows.numel(),), -1)])
        valid = rows >= 0
        src = rows.clamp_min(0)
        if _CHECK_SPARSE_ROWS:                                    # debugging aid: the dense gradient must vanish outside `rows`
            for v in views:
                mask = torch.ones(v.shape[0], dtype=torch.bool, device=dev)
                mask[src[valid]] = False
                assert not bool(v[mask].any()), "row-sparse gradient group has non-zero rows outside its row record"
        # a row listed twice (two images of one class, the shared no-object row) holds the SUM already: send it once
        first = ~((rows[:, None] == rows[None, :]).tril(-1).any(1)) & valid
        pack = torch.cat([v[src] for v in views], dim=1) * first[:, None].to(g.grad.dtype)
        rows_flag = torch.cat([rows, over.to(rows.dtype)])        # [cap + 1]: the rows and this rank's overflow flag
        all_pack = [torch.empty_like(pack) for _ in range(self.world)]
        all_rows = [torch.empty_like(rows_flag) for _ in range(self.world)]
        dist.all_gather(all_pack, pack.contiguous(), group=self.pg)
        dist.all_gather(all_rows, rows_flag.contiguous(), group=self.pg)
        any_over = torch.stack([r[cap] for r in all_rows]).max().reshape(1)        # identical on every rank
        if dev.type == "cuda":
            pinned = torch.zeros(1, dtype=torch.int64, pin_memory=True)
            pinned.copy_(any_over, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            self._overflow.append((pinned, ev))
        else:
            self._overflow.append((any_over, None))
        all_rows = [r[:cap] for r in all_rows]
        # A truncated exchange is never APPLIED: when any rank's flag is set the group's gradient of this step is zero on every
        # rank — decided on the device from the gathered flags, identical everywhere, no host read — and the error surfaces at the next
        # _check_overflow() (the next step, or check_now() before a checkpoint / at the end of training).
        ok = (any_over == 0).to(g.grad.dtype)
        # sentinel rows arrive with an all-zero payload: clamped to row 0 they zero it (its gradient is rebuilt from the payloads
        # of the ranks that touched it, or is an exact zero anyway) and add zeros — no host branch, no compaction
        rows_all, pack_all = torch.cat(all_rows).clamp_min(0), torch.cat(all_pack) * (ok / self.world)
        col = 0
        for v in views:
            w = v.shape[1]
            v[rows_all] = 0                                       # (v is a view of the flat gradient buffer)
            v.index_add_(0, rows_all, pack_all[:, col:col + w])
            if beyond is not None:
                v[beyond] = 0                                     # rows this rank could not send
            elif no_record:
                v.mul_(ok)                                        # (rows beyond the device-side search's cap, if any: the dense pass of this path)
            col += w

    def finish(self):
        """call after backward: reduce buckets whose hooks did not all fire (unused params), wait for the
        collectives and make the compute stream wait for the side stream."""
        if not self.active:
            return
        for b in self.buckets[self._next:]:                      # in index order, like the hooks
            self._launch(b)
        self._next = 0
        for b in self.buckets:
            buf = self.flat.groups[b.group].grad[b.start:b.end]
            if self._side is not None:
                with torch.cuda.stream(self._side):              # the SIDE stream waits for the collective ...
                    b.work.wait()
                    if not self._use_avg:
                        buf.div_(self.world)                     # ... so the division is ordered after its result
            else:
                b.work.wait()
                if not self._use_avg:
                    buf.div_(self.world)
            b.work, b.pending = None, b.total
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        for gi in self.sparse_groups:
            self._exchange_rows(gi)
        if self.optimizer is not None:
            self.optimizer.grads_ready = True                    # step() must not gather again

    def check_now(self):
        """wait for the overflow flags of the steps issued so far and raise if one is set: before a checkpoint is written and after the
        last step (the per-step check is non-blocking and looks at EARLIER steps only)"""
        for flag, ev in self._overflow:
            if ev is not None:
                ev.synchronize()
        self._check_overflow()

    def remove(self):
        for h in self._hooks:
            h.remove()
        if self._early:
            from ..modeling.backbone import resnet_core
            if isinstance(resnet_core.PUBLISH, _WeakPublish) and resnet_core.PUBLISH.ref() is self:
                resnet_core.PUBLISH = None


def broadcast_parameters(flat, src=0, process_group=None):
    """one broadcast per flat group so every rank starts from rank `src`'s weights."""
    from ..utils.misc import collectives_active
    if dist.is_initialized() and (dist.get_world_size(process_group) > 1 or collectives_active()):
        for g in flat.groups:
            dist.broadcast(g.param, src=src, group=process_group)
            if g.shadow is not None:
                g.shadow.copy_(g.param)
