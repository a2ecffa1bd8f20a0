"""Flat parameter / gradient storage.

Every trainable parameter of a hyper-parameter group (same lr, weight decay and
dtype) becomes a view into ONE contiguous buffer, and its ``.grad`` a view into
a second one.  That turns the reference's per-tensor optimizer loop
(base_trainer.py:118-133 + torch.optim.AdamW) into two bandwidth-bound kernel
launches per group and lets the data-parallel reducer all-reduce contiguous
slices in place (no bucket copy-in / copy-out)."""
from typing import Dict, List

import torch


class FlatGroup:
    def __init__(self, params: List[torch.nn.Parameter], names: List[str], hyper: Dict):
        self.params, self.names, self.hyper = params, names, dict(hyper)
        dev, dt = params[0].device, params[0].dtype
        self.offsets, total = [], 0
        for p in params:
            self.offsets.append(total)
            total += (p.numel() + 3) // 4 * 4                       # keep every tensor 16-byte aligned (fp32)
        self.numel = total
        self.param = torch.zeros(total, dtype=dt, device=dev)
        self.grad = torch.zeros(total, dtype=dt, device=dev)
        for p, off in zip(params, self.offsets):
            view = self._view(self.param, p, off)
            view.copy_(p.data)
            p.data = view
            p.grad = self._view(self.grad, p, off)

    @staticmethod
    def _view(flat, p, off):
        """conv weights (4-d) are stored channels-last inside the flat buffer so MIOpen sees NHWC filters."""
        seg = flat[off:off + p.numel()]
        if p.dim() == 4:
            o, i, kh, kw = p.shape
            return seg.view(o, kh, kw, i).permute(0, 3, 1, 2)
        return seg.view(p.shape)

    def rebind_grads(self):
        """autograd may have replaced .grad (e.g. after zero_grad(set_to_none=True)): point it back."""
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.grad[off:].data_ptr():
                p.grad = self._view(self.grad, p, off)


class FlatParams:
    """groups: list of {"params": [...], "names": [...], **hyper}; order inside a group = order given."""

    def __init__(self, groups: List[Dict]):
        self.groups = []
        for g in groups:
            hyper = {k: v for k, v in g.items() if k not in ("params", "names")}
            by_dtype: Dict[torch.dtype, List[int]] = {}
            for i, p in enumerate(g["params"]):
                by_dtype.setdefault(p.dtype, []).append(i)
            for dt, idx in by_dtype.items():
                self.groups.append(FlatGroup([g["params"][i] for i in idx], [g["names"][i] for i in idx], hyper))

    def zero_grad(self):
        for g in self.groups:
            g.grad.zero_()
            g.rebind_grads()

    def numel(self):
        return sum(g.numel for g in self.groups)
