"""The training step of the reference drivers (detectron2 AMPTrainer.run_step as used by part_proposal_train_net.py /
part_distillation_train_net.py through base_trainer.BaseTrainer): autocast forward -> summed weighted losses ->
backward (gradient all-reduce overlapped) -> clipped AdamW -> LR schedule.

bf16 autocast needs no GradScaler (the reference's fp16 AMP on V100 does); the pixel decoder and the matcher costs
stay fp32 as in the reference.

The step issues ~3 000 kernels; PyTorch's eager dispatch of them costs more host time than the GPU needs to run
them, so on a single GPU the whole step (forward, criterion, backward, gradient gather, optimizer) is captured ONCE
into a hipGraph and replayed: the step has no host synchronisation and no data-dependent shapes (device-side
Hungarian matching, host-known index tables), learning rate and Adam bias corrections are read from device memory,
and the batch is copied into static input buffers before each replay."""
import torch
import torch.distributed as dist

from ..compat import build_model
from ..compat.structures import BitMasks, Instances
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
        self._graph = None
        self._replay_done = None

    # ------------------------------------------------------------------ eager
    def _forward_backward(self, batched_inputs):
        dev_type = self.model.device.type
        self.optimizer.zero_grad()
        with torch.autocast(device_type=dev_type, dtype=torch.bfloat16, enabled=self.amp):
            loss_dict = self.model(batched_inputs)
            total = getattr(loss_dict, "total", None)
            if total is None:
                total = sum(loss_dict.values())
        total.backward()
        return loss_dict

    def __call__(self, batched_inputs):
        """one optimisation step; returns the dict of weighted losses (device scalars, no host sync)."""
        if self._graph is not None and self._signature(batched_inputs) == self._graph_sig:
            return self._replay(batched_inputs)
        loss_dict = self._forward_backward(batched_inputs)
        self.reducer.finish()
        self.optimizer.step()
        self.scheduler.step()
        self.iter += 1
        return loss_dict

    # ------------------------------------------------------------------ hipGraph
    @staticmethod
    def _signature(batch):
        """everything the captured step read on the HOST (and therefore baked into the graph): tensor shapes and the
        image's object class (the part-distillation decoder selects class-head rows from it on the host)."""
        return tuple((tuple(x["image"].shape), tuple(x["instances"].gt_masks.tensor.shape), int(x.get("gt_object_class", -1)))
                     for x in batch)

    def _flat_state(self):
        opt = self.optimizer
        return ([g.param for g in opt.flat.groups] + [g.shadow for g in opt.flat.groups if g.shadow is not None]
                + list(opt.exp_avg) + list(opt.exp_avg_sq))

    def capture(self, example_batch, warmup=3):
        """capture the whole step for batches shaped like `example_batch` (single process only).  The warm-up runs real
        steps (allocator growth, MIOpen/BLAS lazy initialisation need the full kernel sequence) on a SNAPSHOT of the
        weights, bf16 shadows and Adam moments that is restored afterwards, so capturing does not move the training
        trajectory (weights, moments and step count are exactly what they were before the call)."""
        if self.world > 1:
            raise RuntimeError("hipGraph capture of the step is for the single-GPU path (collectives stay eager)")
        static = []
        for x in example_batch:
            inst = Instances(x["instances"].image_size)
            inst.gt_masks = BitMasks(x["instances"].gt_masks.tensor.clone())
            inst.gt_classes = x["instances"].gt_classes.clone()
            static.append({**{k: v for k, v in x.items() if k not in ("image", "instances")},
                           "image": x["image"].clone(), "instances": inst})
        # warm-up on the CURRENT stream (allocator growth, MIOpen/BLAS lazy initialisation).  NB: the usual
        # "warm up on a side stream" recipe makes the second replay of this graph fault on ROCm 7.2 (observed:
        # tools/debug_graph2.py cap0 vs DBG_NOSIDE), so no extra stream is created here.
        live = self._flat_state()
        snapshot = [t.clone() for t in live]
        steps0 = self.optimizer.steps
        for _ in range(warmup):
            self._forward_backward(static)
            self.optimizer.step()
        with torch.no_grad():
            for t, s in zip(live, snapshot):
                t.copy_(s)
        self.optimizer.steps = steps0
        del snapshot
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad()
        with torch.cuda.graph(graph):
            loss_dict = self._forward_backward(static)
            self.optimizer.launch_step()
        self._graph, self._static, self._static_losses = graph, static, loss_dict
        self._graph_sig = self._signature(example_batch)
        return self

    def _replay(self, batch):
        # ONE replay in flight, fenced by a DEVICE synchronize: on ROCm 7.2 back-to-back replays of this graph fault,
        # replays separated by torch.cuda.synchronize() never do, and an event recorded on the launch stream after
        # hipGraphLaunch is not enough (the graph's internal branch streams can still be running) — observed with
        # tools/debug_graph2.py and bench.py PD_DEBUG_GRAPH.  The wait is free while the step is GPU-bound.
        if self._replay_done is not None:
            torch.cuda.synchronize()
        for s, x in zip(self._static, batch):
            if s["image"] is not x["image"]:
                s["image"].copy_(x["image"], non_blocking=True)
                s["instances"].gt_masks.tensor.copy_(x["instances"].gt_masks.tensor, non_blocking=True)
                s["instances"].gt_classes.copy_(x["instances"].gt_classes, non_blocking=True)
        self.optimizer.prepare_step()
        self._graph.replay()
        if self._replay_done is None:
            self._replay_done = torch.cuda.Event()
        self._replay_done.record()
        self.scheduler.step()
        self.iter += 1
        return self._static_losses

    # ------------------------------------------------------------------ checkpoints
    def state_dict(self):
        """checkpoint with the reference's key names: fp32 master weights (modules hold bf16 copies of some), buffers,
        optimizer moments and the iteration."""
        sd = {k: v for k, v in self.model.state_dict().items()}
        sd.update(self.optimizer.flat.master_state())
        return {"model": sd, "optimizer": self.optimizer.state_dict(), "iteration": self.iter}

    def load_state_dict(self, sd, strict=False):
        """resume from state_dict(): weights (modules + fp32 masters), Adam moments and step count, the iteration and the
        position of the LR schedule (so a reloaded run does not restart the warm-up)."""
        missing = self.load_model_state(sd["model"], strict=strict)
        self.optimizer.load_state_dict(sd["optimizer"])
        self.iter = int(sd["iteration"])
        self.scheduler.last_iter = self.iter
        self.scheduler._apply()
        return missing

    def load_model_state(self, model_sd, strict=False):
        """load reference-format weights: into the modules (bf16 copies) AND the fp32 masters."""
        missing = self.model.load_state_dict(model_sd, strict=strict)
        masters = self.optimizer.flat.master_state()
        for k, m in masters.items():
            if k in model_sd:
                m.copy_(model_sd[k])
        return missing
