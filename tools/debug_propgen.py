import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from partdistillation_amd import lib; lib.load()
import partdistillation_amd.modeling, partdistillation_amd.proposal_generation_model as pg  # noqa
from partdistillation_amd.compat import BitMasks, Instances, build_model
from partdistillation_amd.config import setup_cfg
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
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); T[key] = T.get(key, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, g)
wrap(pg, "kmeans_lloyd_batched", "kmeans (incl. ++ init)")
wrap(model, "_label_map", "label map"); wrap(model, "_scores", "scores"); wrap(model, "_result", "result (RLE, read-backs)")
wrap(model, "_prepare_features", "prepare features"); wrap(model.backbone, "forward", "backbone")
from partdistillation_amd.functions import kmeans as km
wrap(km, "kmeans_plusplus", "  of which k-means++ init")
for _ in range(3): model(batch)
T.clear()
n = 5
t0 = time.perf_counter()
for _ in range(n): model(batch)
torch.cuda.synchronize()
print("batch of 4: %.1f ms" % ((time.perf_counter() - t0) / n * 1e3))
for k, v in T.items(): print("  %-32s %.2f ms per batch" % (k, v / n * 1e3))
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    model(batch)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60))
