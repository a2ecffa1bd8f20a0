import os, sys, time, json
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
import torch
from partdistillation_amd import lib; lib.load()
import partdistillation_amd.modeling, partdistillation_amd.proposal_generation_model as pgm
from partdistillation_amd.compat import BitMasks, Instances, build_model
from partdistillation_amd.config import setup_cfg
from partdistillation_amd.functions import kmeans as km
torch.backends.cudnn.benchmark = True
cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd", "configs", "proposal_generation", "r50.yaml"))
torch.manual_seed(0)
model = build_model(cfg).cuda().eval()
S = 1024
ys, xs = torch.meshgrid(torch.arange(S) / S, torch.arange(S) / S, indexing="ij")
mask = (((ys - 0.5) ** 2 / 0.13 + (xs - 0.5) ** 2 / 0.085) < 1.0)[None].float().cuda()
batch = []
for b in range(4):
    inst = Instances((S, S)); inst.gt_masks = BitMasks(mask)
    batch.append({"image": (torch.rand(3, S, S, device="cuda") * 255), "instances": inst, "file_name": f"{b}.pth", "class_code": "n0"})
model.kmeans_generator = torch.Generator(device="cuda").manual_seed(0)
T = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w
pgm.kmeans_lloyd_batched = timed("lloyd_batched(total)", pgm.kmeans_lloyd_batched)
km.kmeans_plusplus = timed("  kmeans++ (inside)", km.kmeans_plusplus)
model._result = timed("_result", model._result)
model._label_map = timed("_label_map", model._label_map)
model._scores = timed("_scores", model._scores)
model._prepare_features = timed("_prepare_features", model._prepare_features)
model.backbone.forward = timed("backbone", model.backbone.forward)
for it in range(6):
    if it == 2: T.clear()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model(batch)
        torch.cuda.synchronize(); T["TOTAL"] = T.get("TOTAL", 0.0) + time.perf_counter() - t0
for k, v in T.items(): print(f"{k:26s} {v / 4 * 1e3:7.2f} ms per batch")
