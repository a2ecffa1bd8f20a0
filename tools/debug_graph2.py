import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(ROOT, 'partdistillation_amd', 'miopen_db'))
import torch
sys.path.insert(0, ROOT)
from partdistillation_amd.config import setup_cfg
from partdistillation_amd.engine.synthetic import make_batch
from partdistillation_amd.engine.trainer import TrainStep
mode = sys.argv[1]
cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd", "configs", "proposal_learning", "r50_mask2former.yaml"), [])
torch.manual_seed(0)
step = TrainStep(cfg)
m = step.model
batches = [make_batch(2, 1024, n_parts=4, seed=5 + i, device="cuda") for i in range(2)]
for i in range(3):
    step(batches[i % 2])
torch.cuda.synchronize(); print("eager ok", flush=True)
b = make_batch(2, 1024, n_parts=4, seed=77, device="cuda")     # fresh tensors, like the static clones


def run():
    if mode == "fb":
        return step._forward_backward(b).total
    if mode == "fbopt":
        ld = step._forward_backward(b)
        step.optimizer.launch_step()
        return ld.total
    if mode in ("model", "modelbwd"):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ld = m(b)
        if mode == "modelbwd":
            step.optimizer.zero_grad()
            ld.total.backward()
        return ld.total
    with torch.autocast("cuda", dtype=torch.bfloat16):
        images = m.preprocess(b)
        feats = m.backbone(images.tensor)
        if mode == "bb":
            return sum(f.float().mean() for f in feats.values())
        mf, _, ms = m.sem_seg_head.pixel_decoder.forward_features(feats)
        if mode == "pd":
            return mf.float().mean() + sum(x.float().mean() for x in ms)
        out = m.sem_seg_head.predictor(ms, mf, None)
        if mode == "dec":
            return out["all_masks"].float().mean() + out["pred_logits"].float().mean()
        targets = m._share_padded_masks(m.prepare_targets(b, images))
        losses = m.criterion(out, targets)
        tot = sum(v.sum() for v in losses.vectors.values())
        if mode == "critbwd":
            step.optimizer.zero_grad()
            tot.backward()
        return tot


if mode.startswith("cap"):
    step.capture(b, warmup=int(mode[3:]))
    torch.cuda.synchronize(); print("captured", mode, flush=True)
    for i in range(8):
        if os.environ.get("DBG_COPY"):
            ld = step(batches[i % 2])
        else:
            step._graph.replay()
        torch.cuda.synchronize(); print("replay", mode, i, float(step._static_losses.total.detach()), flush=True)
    sys.exit(0)
if mode != "fboptnowarm":
    run(); torch.cuda.synchronize()
else:
    mode = "fbopt"
g = torch.cuda.CUDAGraph()
step.optimizer.zero_grad()
with torch.cuda.graph(g):
    r = run()
torch.cuda.synchronize(); print("captured", mode, flush=True)
for i in range(3):
    g.replay(); torch.cuda.synchronize(); print("replay", mode, i, float(r.detach()), flush=True)
