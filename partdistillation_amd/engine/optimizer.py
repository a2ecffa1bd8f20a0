"""Optimizer construction of the training step (reference base_trainer.py:64-148):
per-parameter hyper-parameters (backbone lr x BACKBONE_MULTIPLIER, no weight
decay on norm layers / embeddings / relative-position tables, FREEZE_KEYS),
full-model gradient-norm clipping folded into AdamW, and the WarmupMultiStepLR
schedule detectron2's ``build_lr_scheduler`` gives the reference drivers."""
import bisect
from typing import Dict, List

import torch
from torch import nn

from ..compat.layers import FrozenBatchNorm2d
from ..functions import optim as optim_op
from ..functions.fused import PinnedRing
from .flat_params import FlatParams

_NORM_TYPES = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.SyncBatchNorm, nn.GroupNorm, nn.InstanceNorm1d,
               nn.InstanceNorm2d, nn.InstanceNorm3d, nn.LayerNorm, nn.LocalResponseNorm, FrozenBatchNorm2d)


def _wants_bf16_shadow(module_name, module, pname, value):
    """weights that torch.autocast would cast to bf16 at every use: conv / linear / attention projection weights and
    biases of the autocast regions (backbone, transformer predictor).  Norm layers, embeddings, everything in the
    fp32 pixel decoder and non-fp32 parameters keep fp32 storage."""
    if value.dtype != torch.float32 or "pixel_decoder" in module_name or "criterion" in module_name:
        return False
    if isinstance(module, _NORM_TYPES) or isinstance(module, nn.Embedding):
        return False
    if isinstance(module, (nn.Conv2d, nn.Linear)):
        return True
    return pname in ("in_proj_weight", "in_proj_bias")            # attention projections (_MHAParams)


def param_hyperparams(cfg, model) -> List[Dict]:
    """one entry per trainable parameter: {"param", "name", "lr", "weight_decay"} — the loop of
    base_trainer.py:88-116 (including its side effect: parameters whose module name contains a
    FREEZE_KEYS entry get requires_grad=False)."""
    base_lr, base_wd = cfg.SOLVER.BASE_LR, cfg.SOLVER.WEIGHT_DECAY
    out, memo = [], set()
    for module_name, module in model.named_modules():
        for pname, value in module.named_parameters(recurse=False):
            if not value.requires_grad or value in memo:
                continue
            if any(k in module_name for k in cfg.MODEL.MASK_FORMER.FREEZE_KEYS):
                value.requires_grad = False
                continue
            memo.add(value)
            lr, wd = base_lr, base_wd
            if "backbone" in module_name:
                lr = lr * cfg.SOLVER.BACKBONE_MULTIPLIER
            if "relative_position_bias_table" in pname or "absolute_pos_embed" in pname:
                wd = 0.0
            if isinstance(module, _NORM_TYPES):
                wd = cfg.SOLVER.WEIGHT_DECAY_NORM
            if isinstance(module, nn.Embedding):
                wd = cfg.SOLVER.WEIGHT_DECAY_EMBED
            out.append({"param": value, "name": f"{module_name}.{pname}" if module_name else pname, "lr": lr,
                        "weight_decay": wd, "shadow": _wants_bf16_shadow(module_name, module, pname, value)})
    return out


class FlatClippedAdamW:
    """AdamW with full-model L2 clipping over flat buffers (HIP kernels pd_sumsq_accumulate + pd_adamw_clipped).
    ``param_groups`` exposes lr / weight_decay per flat group like a torch optimizer (for LR schedulers)."""

    def __init__(self, entries: List[Dict], betas=(0.9, 0.999), eps=1e-8, clip_norm=0.0, bf16_shadow=False):
        groups: Dict = {}
        # reverse registration order ~ the order gradients become ready in backward (DDP buckets fill front to back)
        for e in reversed(entries):
            shadow = bool(bf16_shadow and e.get("shadow", False))
            key = (e["lr"], e["weight_decay"], shadow)
            g = groups.setdefault(key, {"params": [], "names": [], "lr": e["lr"], "initial_lr": e["lr"],
                                        "weight_decay": e["weight_decay"], "shadow": shadow})
            g["params"].append(e["param"])
            g["names"].append(e["name"])
        self.flat = FlatParams(list(groups.values()))
        self.param_groups = [dict(g.hyper, params=g.params) for g in self.flat.groups]
        self.betas, self.eps, self.clip_norm = betas, eps, clip_norm
        self.exp_avg = [torch.zeros_like(g.param) for g in self.flat.groups]
        self.exp_avg_sq = [torch.zeros_like(g.param) for g in self.flat.groups]
        dev = self.flat.groups[0].param.device
        self._sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        self.steps = 0
        self.grads_ready = False          # set by the data-parallel reducer when it has already gathered + reduced
        # per-group {lr, 1-beta1^t, sqrt(1-beta2^t)} live on the device so that a hipGraph-captured step keeps
        # following the LR schedule / bias corrections when replayed (host refreshes them before every step)
        ng = len(self.flat.groups)
        self._dyn_host = PinnedRing((ng, 4), torch.float32, dev.type == "cuda")
        self._dyn_dev = torch.zeros((ng, 4), dtype=torch.float32, device=dev)

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def grad_norm(self):
        """device scalar: global L2 norm of the (unclipped) gradients of the last step."""
        return self._sumsq.sqrt()

    def prepare_step(self):
        """host side of a step: advance the step count, publish lr / bias corrections to the device."""
        self.steps += 1
        b1, b2 = self.betas
        host = self._dyn_host.acquire()
        h = host.numpy()
        for i, pg in enumerate(self.param_groups):
            h[i, 0], h[i, 1], h[i, 2] = pg["lr"], 1.0 - b1 ** self.steps, (1.0 - b2 ** self.steps) ** 0.5
        self._dyn_dev.copy_(host, non_blocking=True)
        self._dyn_host.release()

    @torch.no_grad()
    def step(self):
        self.prepare_step()
        self.launch_step()

    @torch.no_grad()
    def launch_step(self):
        """device side of a step (capturable in a hipGraph): gather gradients, global norm, clipped AdamW."""
        self._sumsq.zero_()
        if self.grads_ready:                               # flat gradients already gathered and all-reduced
            if self.clip_norm > 0:
                for g in self.flat.groups:
                    optim_op.sumsq_accumulate(g.grad, self._sumsq)
        else:                                              # single process: gather + sum of squares in one pass
            for g in self.flat.groups:
                g.gather(self._sumsq if self.clip_norm > 0 else None)
        single = not self.grads_ready                      # (data-parallel runs: an unused parameter is an error in the reference's DDP)
        self.grads_ready = False
        for i, (g, pg, m, v) in enumerate(zip(self.flat.groups, self.param_groups, self.exp_avg, self.exp_avg_sq)):
            # parameters that received NO gradient this step are left exactly as they are (weights, moments, bf16 copy):
            # torch.optim.AdamW — the reference's optimizer — skips p.grad is None, while the flat kernel would still apply
            # weight decay and decay the moments on their zero-filled slots.  Rare (data-dependent branches: empty targets),
            # so the flat pass stays as it is and the few affected tensors are saved before it and put back after it.
            keep = []
            if single:
                for t in getattr(g, "nograd", ()):
                    views = [g._view(buf, g.params[t], g.offsets[t]) for buf in (g.param, m, v) + ((g.shadow,) if g.shadow is not None else ())]
                    keep.append((views, [x.clone() for x in views]))
            optim_op.adamw_clipped_(g.param, g.grad, m, v, lr=pg["lr"], betas=self.betas, eps=self.eps,
                                    weight_decay=pg["weight_decay"], step=max(self.steps, 1),
                                    grad_sumsq=self._sumsq if self.clip_norm > 0 else None, max_norm=self.clip_norm,
                                    shadow=g.shadow, dyn=self._dyn_dev[i])
            for views, saved in keep:
                for x, y in zip(views, saved):
                    x.copy_(y)

    def state_dict(self):
        return {"steps": self.steps, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "lr": [pg["lr"] for pg in self.param_groups]}

    def load_state_dict(self, sd):
        self.steps = sd["steps"]
        for a, b in zip(self.exp_avg, sd["exp_avg"]):
            a.copy_(b)
        for a, b in zip(self.exp_avg_sq, sd["exp_avg_sq"]):
            a.copy_(b)
        for pg, lr in zip(self.param_groups, sd["lr"]):
            pg["lr"] = lr


def build_optimizer(cfg, model):
    entries = param_hyperparams(cfg, model)
    shadow = bool(cfg.SOLVER.AMP.ENABLED) and next(model.parameters()).is_cuda
    if cfg.SOLVER.OPTIMIZER != "ADAMW":
        raise NotImplementedError(f"no optimizer type {cfg.SOLVER.OPTIMIZER} on the MI355X path (ADAMW only)")
    cg = cfg.SOLVER.CLIP_GRADIENTS
    clip = cg.CLIP_VALUE if (cg.ENABLED and cg.CLIP_TYPE == "full_model" and cg.CLIP_VALUE > 0.0) else 0.0
    return FlatClippedAdamW(entries, clip_norm=clip, bf16_shadow=shadow)


class WarmupMultiStepLR:
    """detectron2 WarmupMultiStepLR (what projects.deeplab.build_lr_scheduler returns for the shipped configs):
    lr = base * warmup(iter) * gamma ** #(milestones <= iter)."""

    def __init__(self, optimizer, milestones, gamma=0.1, warmup_factor=0.001, warmup_iters=1000, warmup_method="linear"):
        self.opt, self.milestones, self.gamma = optimizer, sorted(milestones), gamma
        self.warmup_factor, self.warmup_iters, self.warmup_method = warmup_factor, warmup_iters, warmup_method
        self.base_lrs = [pg.get("initial_lr", pg["lr"]) for pg in optimizer.param_groups]
        self.last_iter = 0
        self._apply()

    def _factor(self, it):
        w = 1.0
        if it < self.warmup_iters:
            if self.warmup_method == "constant":
                w = self.warmup_factor
            else:
                alpha = it / self.warmup_iters
                w = self.warmup_factor * (1 - alpha) + alpha
        return w * self.gamma ** bisect.bisect_right(self.milestones, it)

    def _apply(self):
        f = self._factor(self.last_iter)
        for pg, b in zip(self.opt.param_groups, self.base_lrs):
            pg["lr"] = b * f

    def step(self):
        self.last_iter += 1
        self._apply()


def build_lr_scheduler(cfg, optimizer):
    s = cfg.SOLVER
    if s.LR_SCHEDULER_NAME != "WarmupMultiStepLR":
        raise NotImplementedError(s.LR_SCHEDULER_NAME)
    return WarmupMultiStepLR(optimizer, [x for x in s.STEPS if x <= s.MAX_ITER], s.GAMMA, s.WARMUP_FACTOR, s.WARMUP_ITERS,
                             s.WARMUP_METHOD)
