"""Data-parallel gradient exchange: one process per GPU, the model
replicated, gradients averaged with bucketed all-reduces over RCCL/xGMI that
are issued from autograd hooks on a side HIP stream while backward is still
running.  The buckets are contiguous SLICES of the flat gradient buffers
(flat_params.py), so a bucket is reduced in place — no copy-in / copy-out.

The reference gets this from detectron2's ``create_ddp_model`` (torch DDP,
NCCL); the only other collective on the path is the scalar ``num_masks``
all-reduce of criterion.py:252-254, kept in modeling/criterion.py.
Works with any torch.distributed backend ("nccl" = RCCL on ROCm; "gloo" in the
CPU tests)."""
from typing import List

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("group", "start", "end", "pending", "total", "work")

    def __init__(self, group, start, end, total):
        self.group, self.start, self.end, self.total = group, start, end, total
        self.pending, self.work = total, None


class BucketedGradReducer:
    def __init__(self, flat, bucket_mb: float = 25.0, process_group=None, overlap: bool = True):
        self.flat, self.pg = flat, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.overlap = overlap
        self.buckets: List[_Bucket] = []
        self._param_bucket = {}
        cap_bytes = int(bucket_mb * 1024 * 1024)
        for gi, g in enumerate(flat.groups):
            cap = max(1, cap_bytes // g.grad.element_size())
            start, count = 0, 0
            members = []
            for p, off in zip(g.params, g.offsets):
                end = off + (p.numel() + 3) // 4 * 4
                members.append(p)
                count += 1
                if end - start >= cap:
                    self._add(gi, start, end, members)
                    start, members, count = end, [], 0
            if members:
                self._add(gi, start, g.numel, members)
        self._use_avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        dev = flat.groups[0].grad.device
        self._side = torch.cuda.Stream(device=dev) if (dev.type == "cuda" and overlap) else None
        self._hooks = []
        if self.world > 1:
            for g in flat.groups:
                for p in g.params:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _add(self, gi, start, end, members):
        b = _Bucket(gi, start, end, len(members))
        self.buckets.append(b)
        for p in members:
            self._param_bucket[p] = b

    # ------------------------------------------------------------------ hooks
    def _on_grad(self, p):
        b = self._param_bucket[p]
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b):
        g = self.flat.groups[b.group]
        buf = g.grad[b.start:b.end]
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream(buf.device))    # grads of this bucket are complete
            with torch.cuda.stream(self._side):
                self._reduce(b, buf)
        else:
            self._reduce(b, buf)

    def _reduce(self, b, buf):
        if self._use_avg:
            b.work = dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.pg, async_op=True)
        else:
            b.work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def finish(self):
        """call after backward: reduce buckets whose hooks did not all fire (unused params), wait for the
        collectives and make the compute stream wait for the side stream."""
        if self.world == 1:
            return
        for b in self.buckets:
            if b.work is None:
                self._launch(b)
        for b in self.buckets:
            b.work.wait()
            if not self._use_avg:
                buf = self.flat.groups[b.group].grad[b.start:b.end]
                if self._side is not None:
                    with torch.cuda.stream(self._side):
                        buf.div_(self.world)
                else:
                    buf.div_(self.world)
            b.work, b.pending = None, b.total
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    def remove(self):
        for h in self._hooks:
            h.remove()


def broadcast_parameters(flat, src=0, process_group=None):
    """one broadcast per flat group so every rank starts from rank `src`'s weights."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        for g in flat.groups:
            dist.broadcast(g.param, src=src, group=process_group)
