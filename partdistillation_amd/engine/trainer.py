"""The training step of the reference drivers (detectron2 AMPTrainer.run_step
as used by part_proposal_train_net.py / part_distillation_train_net.py through
base_trainer.BaseTrainer): autocast forward -> summed weighted losses ->
backward (gradient all-reduce overlapped) -> clipped AdamW -> LR schedule.

bf16 autocast needs no GradScaler (the reference's fp16 AMP on V100 does); the
pixel decoder and the matcher costs stay fp32 as in the reference."""
import torch
import torch.distributed as dist

from ..compat import build_model
from .ddp import BucketedGradReducer, broadcast_parameters
from .optimizer import build_lr_scheduler, build_optimizer


class TrainStep:
    def __init__(self, cfg, model=None, process_group=None):
        import partdistillation_amd.modeling  # noqa: F401  (registers the classes)
        import partdistillation_amd.part_distillation_model  # noqa: F401
        import partdistillation_amd.proposal_model  # noqa: F401
        self.cfg = cfg
        self.model = model if model is not None else build_model(cfg)
        self.model.train()
        self.optimizer = build_optimizer(cfg, self.model)
        self.scheduler = build_lr_scheduler(cfg, self.optimizer)
        self.amp = bool(cfg.SOLVER.AMP.ENABLED)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        broadcast_parameters(self.optimizer.flat, 0, process_group)
        self.reducer = BucketedGradReducer(self.optimizer.flat, cfg.MODEL.AMD.DDP_BUCKET_MB, process_group,
                                           optimizer=self.optimizer)
        self.iter = 0

    def __call__(self, batched_inputs):
        """one optimisation step; returns the dict of weighted losses (device scalars, no host sync)."""
        dev_type = self.model.device.type
        self.optimizer.zero_grad()
        with torch.autocast(device_type=dev_type, dtype=torch.bfloat16, enabled=self.amp):
            loss_dict = self.model(batched_inputs)
            total = getattr(loss_dict, "total", None)
            if total is None:
                total = sum(loss_dict.values())
        total.backward()
        self.reducer.finish()
        self.optimizer.step()
        self.scheduler.step()
        self.iter += 1
        return loss_dict

    def state_dict(self):
        """checkpoint with the reference's key names: fp32 master weights (modules hold bf16 copies of some), buffers,
        optimizer moments and the iteration."""
        sd = {k: v for k, v in self.model.state_dict().items()}
        sd.update(self.optimizer.flat.master_state())
        return {"model": sd, "optimizer": self.optimizer.state_dict(), "iteration": self.iter}

    def load_model_state(self, model_sd, strict=False):
        """load reference-format weights: into the modules (bf16 copies) AND the fp32 masters."""
        missing = self.model.load_state_dict(model_sd, strict=strict)
        masters = self.optimizer.flat.master_state()
        for k, m in masters.items():
            if k in model_sd:
                m.copy_(model_sd[k])
        return missing
