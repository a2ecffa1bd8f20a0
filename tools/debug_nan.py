"""Development tool: run N training steps of the bench workload, print the total loss per step and, at the first
non-finite value, which gradients / parameters are affected.  FUSED_DEC=0 / FUSED_ENC=0 switch the fused cores off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, "partdistillation_amd", "miopen_db"))
from partdistillation_amd import lib
lib.load()
from partdistillation_amd.config import setup_cfg
from partdistillation_amd.engine.synthetic import make_batch
from partdistillation_amd.engine.trainer import TrainStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd", "configs", "proposal_learning", "r50_mask2former.yaml"), ["INPUT.IMAGE_SIZE", "1024"])
torch.manual_seed(0)
step = TrainStep(cfg)
step.model.sem_seg_head.predictor.fused_core = os.environ.get("FUSED_DEC", "1") == "1"
step.model.sem_seg_head.pixel_decoder.transformer.encoder.fused_core = os.environ.get("FUSED_ENC", "1") == "1"
batches = [make_batch(2, 1024, seed=1234 + 1000 * i, device="cuda") for i in range(4)]
for i in range(steps):
    losses = step(batches[i % 4])
    tot = float(sum(v.float() for v in losses.values()))
    bad_g = [n for g in step.optimizer.flat.groups for n, p in zip(g.names, g.params) if p.grad is not None and not torch.isfinite(p.grad.float()).all()]
    bad_p = [n for g in step.optimizer.flat.groups for n, p in zip(g.names, g.params) if not torch.isfinite(p.float()).all()]
    print(i, f"{tot:.4f}", "bad grads:", len(bad_g), bad_g[:6], "bad params:", len(bad_p), bad_p[:4], flush=True)
    if bad_g or bad_p or tot != tot:
        print({k: float(v) for k, v in losses.items()})
        break
