"""Data-parallel gradient exchange: one process per GPU, the model replicated, gradients averaged
with bucketed all-reduces over RCCL/xGMI that are issued from autograd hooks on a side HIP stream while backward is
still running.  A bucket is a run of consecutive parameters of one flat group: when the last of its gradients has
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


class BucketedGradReducer:
    def __init__(self, flat, bucket_mb: float = 25.0, process_group=None, overlap: bool = True, optimizer=None):
        self.flat, self.pg, self.optimizer = flat, process_group, optimizer
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.buckets: List[_Bucket] = []
        self._param_bucket = {}
        cap_bytes = int(bucket_mb * 1024 * 1024)
        # row-sparse groups (the float64 part-distillation class head, [N_obj * K + 1, 256]: 16 MB at 1 000 object classes,
        # 360 MB at the shipped 22 000; part_distillation_transformer_decoder.py:107): a step touches only the K + 1 rows of each
        # image's object class, every other row's gradient is exactly zero on every rank.  They get no buckets: finish()
        # exchanges just the touched rows — ~35 KB per rank instead of the dense buffer.
        self.sparse_groups = [gi for gi, g in enumerate(flat.groups)
                              if self.world > 1 and all(getattr(p, "_pd_row_sparse", False) for p in g.params)]
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
        self._row_meta = {}               # sparse group -> (rows, [R max, any rank without a record], event)
        self._use_avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        dev = flat.groups[0].grad.device
        self._side = torch.cuda.Stream(device=dev) if (dev.type == "cuda" and overlap) else None
        self._hooks = []
        if self.world > 1:
            for gi, g in enumerate(flat.groups):
                if gi in self.sparse_groups:
                    continue
                for p in g.params:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # ------------------------------------------------------------------ hooks
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
        b.work = dist.all_reduce(buf, op=op, group=self.pg, async_op=True)

    def _exchange_rows(self, gi):
        """mean over the ranks of a row-sparse group's gradient from the touched rows only.  Every parameter of the group is
        [rows, ...] and carries `_pd_rows` (LongTensor [R], the rows this rank's step used: images per GPU x (K + 1)).  Equal to
        the dense all-reduce: untouched rows are exact zeros everywhere.
        The ranks first agree on ONE path (a MAX all-reduce of [R, no-record flag], 16 bytes): if any rank has no row record this
        step, or the ranks' R differ (uneven last batch), a rank that went on to all_gather R-sized buffers would hang the others or
        scatter garbage — so R is padded to the maximum with a sentinel row (-1, dropped on arrival) and a missing record sends
        everyone down the dense all-reduce.  The record is consumed: a stale list is never exchanged for a later step."""
        g = self.flat.groups[gi]
        g.gather(None)                                            # local dense gradients -> flat buffer (compute stream)
        rows, meta_host, ev = self._row_meta.pop(gi)
        dev = g.grad.device
        if ev is not None:
            ev.synchronize()                                      # waits for the 16-byte agreement only (side stream), not for the compute stream
        r_max, any_missing = int(meta_host[0]), int(meta_host[1])
        side = self._side
        if side is not None:                                      # the exchange itself runs on the side stream, behind the bucket all-reduces
            side.wait_stream(torch.cuda.current_stream(dev))
        ctx = torch.cuda.stream(side) if side is not None else contextlib.nullcontext()
        with ctx:
            self._exchange_rows_on_stream(g, rows, r_max, any_missing, dev)
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)

    def _agree_rows(self, gi):
        """start the ranks' agreement on the row-sparse path of group gi (MAX all-reduce of [R, no-record flag]); issued by
        finish() right behind the last bucket all-reduce, on the side stream, read back through a pinned buffer"""
        g = self.flat.groups[gi]
        rows = getattr(g.params[0], "_pd_rows", None)
        for p in g.params:
            if hasattr(p, "_pd_rows"):
                p._pd_rows = None                                 # consumed: a stale list is never exchanged for a later step
        dev = g.grad.device
        host = torch.tensor([0 if rows is None else int(rows.numel()), 1 if rows is None else 0], dtype=torch.int64)
        if dev.type != "cuda":
            dist.all_reduce(host, op=dist.ReduceOp.MAX, group=self.pg)
            self._row_meta[gi] = (rows, host, None)
            return
        pinned = torch.empty(2, dtype=torch.int64, pin_memory=True)
        pinned.copy_(host)
        side = self._side if self._side is not None else torch.cuda.current_stream(dev)
        with torch.cuda.stream(side):
            meta = pinned.to(dev, non_blocking=True)
            dist.all_reduce(meta, op=dist.ReduceOp.MAX, group=self.pg)
            pinned.copy_(meta, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        self._row_meta[gi] = (rows, pinned, ev)

    def _exchange_rows_on_stream(self, g, rows, r_max, any_missing, dev):
        if any_missing or r_max == 0:                             # dense path, on EVERY rank
            dist.all_reduce(g.grad, group=self.pg)
            g.grad.div_(self.world)
            return
        rows = rows.reshape(-1).to(dev)
        if _CHECK_SPARSE_ROWS:                                    # debugging aid: the dense gradient must vanish outside `rows`
            views_chk = [g._view(g.grad, p, off).reshape(p.shape[0], -1) for p, off in zip(g.params, g.offsets)]
            for v in views_chk:
                mask = torch.ones(v.shape[0], dtype=torch.bool, device=dev)
                mask[rows] = False
                assert not bool(v[mask].any()), "row-sparse gradient group has non-zero rows outside its row record"
        # a row listed twice (two images of one class, the shared no-object row) holds the SUM already: send it once
        first = ~(rows[:, None] == rows[None, :]).tril(-1).any(1)
        views = [g._view(g.grad, p, off).reshape(p.shape[0], -1) for p, off in zip(g.params, g.offsets)]
        pack = torch.cat([v[rows] for v in views], dim=1) * first[:, None].to(g.grad.dtype)
        if rows.numel() < r_max:                                  # pad to the agreed size: sentinel rows carry zeros
            pad = r_max - rows.numel()
            rows = torch.cat([rows, rows.new_full((pad,), -1)])
            pack = torch.cat([pack, pack.new_zeros((pad, pack.shape[1]))])
        all_pack = [torch.empty_like(pack) for _ in range(self.world)]
        all_rows = [torch.empty_like(rows) for _ in range(self.world)]
        dist.all_gather(all_pack, pack.contiguous(), group=self.pg)
        dist.all_gather(all_rows, rows.contiguous(), group=self.pg)
        rows_all, pack_all = torch.cat(all_rows), torch.cat(all_pack) / self.world
        keep = rows_all >= 0
        if not bool(keep.all()):
            rows_all, pack_all = rows_all[keep], pack_all[keep]
        col = 0
        for v in views:
            w = v.shape[1]
            v[rows_all] = 0                                       # (v is a view of the flat gradient buffer)
            v.index_add_(0, rows_all, pack_all[:, col:col + w])
            col += w

    def finish(self):
        """call after backward: reduce buckets whose hooks did not all fire (unused params), wait for the
        collectives and make the compute stream wait for the side stream."""
        if self.world == 1:
            return
        for b in self.buckets[self._next:]:                      # in index order, like the hooks
            self._launch(b)
        self._next = 0
        for gi in self.sparse_groups:                            # after the LAST bucket on every rank: one collective order for all
            self._agree_rows(gi)
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

    def remove(self):
        for h in self._hooks:
            h.remove()


def broadcast_parameters(flat, src=0, process_group=None):
    """one broadcast per flat group so every rank starts from rank `src`'s weights."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        for g in flat.groups:
            dist.broadcast(g.param, src=src, group=process_group)
            if g.shadow is not None:
                g.shadow.copy_(g.param)
