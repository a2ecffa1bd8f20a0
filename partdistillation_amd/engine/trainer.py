"""The training step of the reference drivers (detectron2 AMPTrainer.run_step as used by part_proposal_train_net.py /
part_distillation_train_net.py through base_trainer.BaseTrainer): autocast forward -> summed weighted losses ->
backward (gradient all-reduce overlapped) -> clipped AdamW -> LR schedule.

bf16 autocast needs no GradScaler (the reference's fp16 AMP on V100 does); the pixel decoder and the matcher costs
stay fp32 as in the reference.

The step is issued EAGERLY (the default and the only mode used by bench.py and the tests): it has no host
synchronisation and no data-dependent shapes (device-side Hungarian matching, host-known index tables; learning rate
and Adam bias corrections are read from device memory), so the host simply runs ahead of the GPU.  `capture()` can
record the whole step into one hipGraph, but on ROCm 7.2 the only SAFE replay mode (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0)
costs more host time than eager issue, so it is opt-in and refuses to run without that setting (DESIGN.md §5
"hipGraph")."""
import os as _os
import warnings

import torch
import torch.distributed as dist

from ..compat import build_model
from ..functions.conv_bf16 import deferred_wgrads as _deferred_wgrads
from ..compat.structures import BitMasks, Instances
from .ddp import BucketedGradReducer, broadcast_parameters
from .optimizer import build_lr_scheduler, build_optimizer

# the backward pass issued from the CALLING thread instead of autograd's device thread: its ~500 nodes are mostly Python callbacks (the
# hand-written backward passes), so the worker thread spent its time taking the GIL from the waiting main thread — host issue of the
# step 24.2 -> 21.1 ms where the host is the limiter (512 x 512), nothing changes where the GPU is.  One process drives one GPU here.
_SINGLE_THREAD_BWD = bool(int(_os.environ.get("PD_SINGLE_THREAD_BWD", "1")))
_GRAPH_SYNC = bool(int(_os.environ.get("PD_GRAPH_SYNC", "0")))          # debugging aid: device fence between two replays of the captured step
# read ONCE at import: the HIP runtime reads the variable when it initialises, so a value set later passes a check of os.environ
# but changes nothing (bench.py --graph 1 sets it before importing torch)
_PACKET_CAPTURE_AT_IMPORT = _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE")


def _detached(loss_dict):
    out = type(loss_dict)({k: v.detach() for k, v in loss_dict.items()})
    for name in ("vectors", "total", "indices", "points"):
        v = getattr(loss_dict, name, None)
        if isinstance(v, torch.Tensor):
            v = v.detach()
        elif isinstance(v, dict):
            v = {k: (x.detach() if isinstance(x, torch.Tensor) else x) for k, x in v.items()}
        if v is not None:
            setattr(out, name, v)
    return out


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
                                           optimizer=self.optimizer, sparse_rows_cap=int(cfg.MODEL.AMD.get("DDP_SPARSE_ROWS_CAP", 256)))
        self._rows_per_image = int(cfg.PART_DISTILLATION.NUM_PART_CLASSES) + 1 if self.reducer.sparse_groups else 0
        if self.reducer.sparse_groups:
            # the static row-sparse exchange sends `cap` rows per rank: a step touches (K + 1) class-head rows per image
            per_rank = -(-int(cfg.SOLVER.IMS_PER_BATCH) // max(self.world, 1))
            need = per_rank * (int(cfg.PART_DISTILLATION.NUM_PART_CLASSES) + 1)
            if need > self.reducer.sparse_rows_cap:
                raise ValueError(f"MODEL.AMD.DDP_SPARSE_ROWS_CAP = {self.reducer.sparse_rows_cap} is below images per rank x (parts + 1) = "
                                 f"{per_rank} x {int(cfg.PART_DISTILLATION.NUM_PART_CLASSES) + 1} = {need}: raise the cap (it is a static exchange size, the same on every rank)")
        if _os.environ.get("PD_CONV_GROUP_ROWS"):               # tools/ experiments
            from .. import lib as _l
            _l.load().pd_debug_set(b"conv_group_rows", int(_os.environ["PD_CONV_GROUP_ROWS"]))
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
        with _deferred_wgrads():                             # the backbone's filter gradients: one grouped launch after backward
            if _SINGLE_THREAD_BWD:
                with torch.autograd.set_multithreading_enabled(False):    
                    total.backward()
            else:
                total.backward()
        return loss_dict

    def __call__(self, batched_inputs):
        """one optimisation step; returns the dict of weighted losses (device scalars, no host sync)."""
        if self._graph is not None and self._signature(batched_inputs) == self._graph_sig:
            return self._replay(batched_inputs)
        if self.reducer.sparse_groups and len(batched_inputs) * self._rows_per_image > self.reducer.sparse_rows_cap:
            raise ValueError(f"this step's {len(batched_inputs)} images x (parts + 1) = {len(batched_inputs) * self._rows_per_image} class-head rows exceed "
                             f"MODEL.AMD.DDP_SPARSE_ROWS_CAP = {self.reducer.sparse_rows_cap} (every rank must be given at most cap // (parts + 1) images)")
        loss_dict = self._forward_backward(batched_inputs)
        self.reducer.finish()
        self.optimizer.step()
        self.scheduler.step()
        self.iter += 1
        return loss_dict

    # ------------------------------------------------------------------ hipGraph
    def release_graph(self):
        self._graph = self._static = self._static_losses = self._graph_sig = None
        self._replay_done = None
        from ..functions.fused import PinnedRing
        PinnedRing.release_captured()                      # the pinned buffers reserved for the uploads baked into the graph

    def _signature(self, batch):
        """everything the captured step read on the HOST (and therefore baked into the graph): tensor shapes and, for
        the part-distillation model, the image's object class (its decoder selects class-head rows from it on the host;
        the proposal model never looks at it)."""
        cls = (lambda x: int(x.get("gt_object_class", -1))) if getattr(self.model, "host_reads_object_class", False) else (lambda x: -1)
        return tuple((tuple(x["image"].shape), tuple(x["instances"].gt_masks.tensor.shape), cls(x)) for x in batch)

    def _flat_state(self):
        opt = self.optimizer
        return ([g.param for g in opt.flat.groups] + [g.shadow for g in opt.flat.groups if g.shadow is not None]
                + list(opt.exp_avg) + list(opt.exp_avg_sq))

    def _segment(self, batch, segment, rehearsal):
        """the captured sequence ("full"; "fb" / "fwd" are truncations used to bisect replay faults)"""
        if segment == "fwd":
            with torch.no_grad(), torch.autocast(device_type=self.model.device.type, dtype=torch.bfloat16, enabled=self.amp):
                return self.model(batch)
        loss_dict = self._forward_backward(batch)
        if segment == "full":
            self.optimizer.launch_step() if rehearsal else self.optimizer.step()
        return loss_dict

    def capture(self, example_batch, warmup=3, _segment="full"):
        """capture the whole step for batches shaped like `example_batch` (single process only).  The warm-up runs real
        steps (allocator growth, MIOpen/BLAS lazy initialisation need the full kernel sequence) on a SNAPSHOT of the
        weights, bf16 shadows and Adam moments that is restored afterwards, so capturing does not move the training
        trajectory (weights, moments and step count are exactly what they were before the call).
        Requires a process started with PD_CMDBUF=0 (after any command-buffer recording hipGraphInstantiate segfaults on ROCm 7.2;
        the call raises instead) and DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 — see INTEGRATION.md "hipGraph".  The command buffers and the
        fused ResNet body are switched off for the duration of the call and restored on every exit path."""
        if self.world > 1:
            raise RuntimeError("hipGraph capture of the step is for the single-GPU path (collectives stay eager)")
        if warmup < 1:
            raise ValueError("TrainStep.capture(): warmup must be >= 1 (the last warm-up step is the rehearsal that sizes the pinned buffers)")
        if _PACKET_CAPTURE_AT_IMPORT != "0":
            # ROCm 7.2 instantiates graphs with pre-built AQL packets ("packet capture").  Such a graph goes stale once a
            # few thousand EAGER launches have been issued since it was instantiated (the input copies between replays
            # count): its kernels then read wrong arguments — first silently (NaN gradient norm in the replayed
            # optimizer), later as a memory access fault (replay, 3 eager steps, replay -> 5/5 faults; bench.py --graph 1 --steps 2000 faults, --steps 300 does not).  None of
            # HIP_FORCE_DEV_KERNARG / DEBUG_HIP_KERNARG_COPY_OPT / DEBUG_HIP_FORCE_GRAPH_QUEUES / ... changes that;
            # DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 does (0 faults, replay == eager), at 60 ms of host CPU per replay of
            # this ~3 000-node graph, i.e. slower than eager issue (19 ms).  Silent corruption is not an option, so:
            raise RuntimeError("TrainStep.capture(): set DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment BEFORE the first "
                               "HIP call of the process (ROCm 7.2 packet-captured graphs go stale after eager launches; "
                               "see DESIGN.md §5 'hipGraph')")
        static = []
        for x in example_batch:
            inst = Instances(x["instances"].image_size)
            inst.gt_masks = BitMasks(x["instances"].gt_masks.tensor.clone())
            inst.gt_classes = x["instances"].gt_classes.clone()
            static.append({**{k: v for k, v in x.items() if k not in ("image", "instances")},
                           "image": x["image"].clone(), "instances": inst})
        # warm-up on the CURRENT stream (allocator growth, MIOpen/BLAS lazy initialisation); no extra stream is needed.
        live = self._flat_state()
        snapshot = [t.clone() for t in live]
        steps0 = self.optimizer.steps
        # the warm-up consumes random numbers (criterion sample points, DropPath): put both generators back afterwards
        rng_cpu, rng_dev = torch.get_rng_state(), (torch.cuda.get_rng_state() if torch.cuda.is_available() else None)
        from ..functions.fused import PinnedRing
        from .. import cmdbuf
        # the captured step is the EAGER launch sequence (a recording cannot be made or replayed while a stream captures): its
        # rehearsal must issue — and count the pinned uploads of — that same sequence, not the command-buffer replays
        if cmdbuf._EVER[0]:
            raise RuntimeError("TrainStep.capture(): command-buffer recordings (partdistillation_amd/cmdbuf.py) were made earlier in this "
                               "process; on ROCm 7.2 hipGraphInstantiate then segfaults on the captured step (bisected with "
                               "tests/test_graph_gpu.py: PD_CMDBUF=0 from process start passes).  Start the process with PD_CMDBUF=0 to use "
                               "the whole-step graph (bench.py --graph 1 does)")
        cmdbuf_was, cmdbuf.ENABLED = cmdbuf.ENABLED, False
        # ... and the module-by-module backbone: instantiating a graph that holds the fused ResNet body's launches segfaults inside
        # hipGraphInstantiate on ROCm 7.2 (the whole-step graph is opt-in and not the shipped mode: DESIGN.md 5 "hipGraph")
        from ..modeling.backbone import resnet_core
        r50_was, resnet_core.ENABLED = resnet_core.ENABLED, False
        try:                                                    # whatever happens below (out of memory, a refused graph): both switches come back
            before = PinnedRing.counters()
            for w in range(warmup):
                if w == warmup - 1:
                    before = PinnedRing.counters()                   # the last warm-up step is the rehearsal of the captured sequence
                self._segment(static, _segment, rehearsal=w == warmup - 1)
            PinnedRing.reserve_all(before, PinnedRing.counters())   # dedicated pinned buffers for the uploads the capture will bake in
            with torch.no_grad():
                for t, s in zip(live, snapshot):
                    t.copy_(s)
            self.optimizer.steps = steps0
            del snapshot
            torch.set_rng_state(rng_cpu)
            if rng_dev is not None:
                torch.cuda.set_rng_state(rng_dev)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            self.optimizer.zero_grad()
            with torch.cuda.graph(graph):
                loss_dict = self._segment(static, _segment, rehearsal=True)
            # keep only DETACHED views of the static losses: holding the captured autograd graph alive would also keep its
            # AccumulateGrad nodes, which are bound to the capture stream — every later EAGER step (a batch with another
            # signature) would then run its gradient accumulation on that stream, and the next replay faults on ROCm 7.2
            # (DESIGN.md §5 "hipGraph").
            loss_dict = _detached(loss_dict)
        finally:
            cmdbuf.ENABLED, resnet_core.ENABLED = cmdbuf_was, r50_was
        self._graph, self._static, self._static_losses = graph, static, loss_dict
        self._graph_sig = self._signature(example_batch)
        return self

    def _replay(self, batch):
        # Replays are stream-ordered like any other launch; no host-side fence.  (Round 1 put a device synchronize between
        # replays against an "intermittent" fault: that was the packet-capture bug capture() now refuses to run with.
        # PD_GRAPH_SYNC=1 restores the fence for debugging.)
        if self._replay_done is not None and _GRAPH_SYNC:
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
        self.reducer.check_now()                           # a truncated row-sparse exchange of a step already taken raises here, not after the save
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
