"""development probe: config-2 full-size step vs the oracle — which (head, image) assignments differ and by how much the
oracle's own cost separates them."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
import common as C  # noqa: E402
from oracle import step_ref as R  # noqa: E402
from partdistillation_amd.config import setup_cfg  # noqa: E402
from partdistillation_amd.engine.synthetic import make_batch  # noqa: E402
from partdistillation_amd.engine.trainer import TrainStep  # noqa: E402

amp = len(sys.argv) > 1 and sys.argv[1] == "amp"
init = sys.argv[2] if len(sys.argv) > 2 else "ref"
cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd", "configs", "proposal_learning", "r50_mask2former.yaml"),
                ["INPUT.IMAGE_SIZE", "1024", "SOLVER.AMP.ENABLED", str(amp), "SOLVER.WARMUP_ITERS", "0"])
torch.manual_seed(0)
step = TrainStep(cfg)
if init == "seeded":
    table = {k: v for k, v in C.table_of(step.model.state_dict()).items() if not k.startswith("criterion.")}
    step.load_model_state(C.seeded_weights(table, 77))
sd = {k: v.detach().float().cpu().clone() for k, v in step.state_dict()["model"].items()}
batch = make_batch(1, 1024, seed=1234, device="cuda")
step.model.criterion.rand = C.ReplayRand(31337)
step.optimizer.step = lambda: None
losses = step(batch)
torch.set_num_threads(min(os.cpu_count() or 1, 32))
obatch = [{"image": b["image"].cpu(), "instances": {"gt_masks": b["instances"].gt_masks.tensor.cpu()}} for b in batch]
with torch.no_grad():
    olosses, oidx = R.proposal_model_losses(sd, obatch, C.ReplayRand(31337), return_indices=True)
rows, cols = (t.cpu() for t in losses.indices)
H = 10
for d in range(H):
    h = 0 if d == H - 1 else d + 1
    got = sorted(zip(rows[d, :4].tolist(), cols[d, :4].tolist()))
    want = sorted(zip(oidx[h][0][0].tolist(), oidx[h][0][1].tolist()))
    print("head", h, "same" if got == want else "DIFF", got, want)
for k in olosses:
    a, b = float(losses[k]), float(olosses[k])
    print(f"{k:14s} gpu {a:.6f} cpu {b:.6f} rel {abs(a - b) / max(abs(b), 1e-12):.2e}")
