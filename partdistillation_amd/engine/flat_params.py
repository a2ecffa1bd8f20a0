"""Flat parameter / gradient storage.

Every trainable parameter of a hyper-parameter group (same lr, weight decay, master dtype, shadow flag) lives in
ONE contiguous fp32 "master" buffer; the group's gradients are collected into a second flat buffer.  That turns the
reference's per-tensor optimizer loop (base_trainer.py:118-133 + torch.optim.AdamW) into two bandwidth-bound kernel
launches per group and lets the data-parallel reducer all-reduce contiguous slices in place.

* ``shadow`` groups (weights that autocast would re-cast to bf16 on every use: backbone convolutions, decoder
  linears) additionally own a flat bf16 copy; the ``nn.Parameter``s are views of THAT copy, so the modules read
  bf16 weights with no per-step cast kernels and autograd produces bf16 gradients; the optimizer kernel updates the
  fp32 master and refreshes the bf16 copy in the same pass.
* gradients are not accumulated in place by autograd: ``zero_grad`` drops ``.grad``; after backward one
  multi-tensor kernel gathers every gradient tensor autograd produced into the flat buffer (bf16 -> fp32 where
  needed) and accumulates the global sum of squares for clipping on the way (functions/fused.py).
"""
from typing import Dict, List

import torch

from ..functions.fused import GatherPlan


def _aligned(n):
    return (n + 7) // 8 * 8                                       # every tensor 16-byte aligned in the bf16 copy too


class FlatGroup:
    def __init__(self, params: List[torch.nn.Parameter], names: List[str], hyper: Dict, shadow: bool):
        self.params, self.names, self.hyper, self.shadow_on = params, names, dict(hyper), shadow
        dev, dt = params[0].device, params[0].dtype
        self.offsets, total = [], 0
        for p in params:
            self.offsets.append(total)
            total += _aligned(p.numel())
        self.numel = total
        self.param = torch.zeros(total, dtype=dt, device=dev)       # master
        self.grad = torch.zeros(total, dtype=dt, device=dev)
        self.shadow = torch.zeros(total, dtype=torch.bfloat16, device=dev) if shadow else None
        for p, off in zip(params, self.offsets):
            view = self._view(self.param, p, off)
            view.copy_(p.data)
            if shadow:
                sview = self._view(self.shadow, p, off)
                sview.copy_(p.data)
                p.data = sview
            else:
                p.data = view
            p.grad = None
        self.plan = GatherPlan([p.numel() for p in params], self.offsets, dev) if (dev.type == "cuda" and dt == torch.float32) else None

    @staticmethod
    def _view(flat, p, off):
        """conv weights (4-d) are stored channels-last inside the flat buffer so MIOpen sees NHWC filters."""
        seg = flat[off:off + p.numel()]
        if p.dim() == 4:
            o, i, kh, kw = p.shape
            return seg.view(o, kh, kw, i).permute(0, 3, 1, 2)
        return seg.view(p.shape)

    def master_view(self, i):
        return self._view(self.param, self.params[i], self.offsets[i])

    def _grad_sources(self, t_begin, t_end):
        out = []
        for p in self.params[t_begin:t_end]:
            g = p.grad
            if g is not None and g.stride() != p.stride():          # autograd keeps the layout contract; be safe anyway
                g = torch.empty_like(p).copy_(g)
                p.grad = g
            out.append(g)
        return out

    def gather(self, sumsq=None, t_begin=0, t_end=None):
        """collect p.grad of params [t_begin, t_end) into the flat gradient buffer (missing gradients -> zeros; their
        indices are remembered in `self.nograd` so the optimizer can leave those parameters untouched, as torch.optim.AdamW
        skips parameters whose .grad is None)."""
        t_end = len(self.params) if t_end is None else t_end
        if t_begin == 0:
            self.nograd = []
        self.nograd = getattr(self, "nograd", []) + [i for i in range(t_begin, t_end) if self.params[i].grad is None]
        if self.plan is not None:
            self.plan.upload(self._grad_sources(t_begin, t_end), t_begin)
            self.plan.gather(self.grad, sumsq, t_begin, t_end)
            return
        # non-CUDA tensors (the gloo CPU tests of the reducer's host logic) and the rare fp64 group
        for p, off in zip(self.params[t_begin:t_end], self.offsets[t_begin:t_end]):
            dst = self._view(self.grad, p, off)
            if p.grad is None:
                dst.zero_()
            else:
                dst.copy_(p.grad)
        if sumsq is not None and t_end > t_begin:
            a, b = self.offsets[t_begin], (self.offsets[t_end] if t_end < len(self.params) else self.numel)
            sumsq += self.grad[a:b].double().pow(2).sum()


class FlatParams:
    """groups: list of {"params": [...], "names": [...], "shadow": bool, **hyper}; order inside a group = order given."""

    def __init__(self, groups: List[Dict]):
        self.groups: List[FlatGroup] = []
        for g in groups:
            hyper = {k: v for k, v in g.items() if k not in ("params", "names", "shadow")}
            by_dtype: Dict[torch.dtype, List[int]] = {}
            for i, p in enumerate(g["params"]):
                by_dtype.setdefault(p.dtype, []).append(i)
            for dt, idx in by_dtype.items():
                self.groups.append(FlatGroup([g["params"][i] for i in idx], [g["names"][i] for i in idx], hyper,
                                             bool(g.get("shadow", False)) and dt == torch.float32))

    def zero_grad(self):
        for g in self.groups:
            for p in g.params:
                p.grad = None

    def numel(self):
        return sum(g.numel for g in self.groups)

    def master_state(self):
        """{parameter name: fp32 master tensor} (what a checkpoint stores; shadowed modules hold bf16 copies)."""
        return {n: g.master_view(i) for g in self.groups for i, n in enumerate(g.names)}
