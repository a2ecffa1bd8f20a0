"""Measurement of the pixel-grouping proposal generation (BASELINE config 4; SURVEY §8f-1 measurement spec):
B synthetic 1024^2 images, one elliptical object mask each (~35 % of the area), features res3 + res4, metric dot, K = 4.
Prints one JSON line: images/s for the whole stage, the split backbone / grouping, and the grouping's achieved GB/s on
its algorithmic bytes (read res3 + res4 once, K-means over the masked vectors, read the mask, write the label map)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="r50", choices=["r50", "swinl"])
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--amp", type=int, default=0, help="1: bf16 autocast for the backbone")
a = ap.parse_args()
from partdistillation_amd import lib
lib.load()
import partdistillation_amd.modeling, partdistillation_amd.proposal_generation_model  # noqa: F401,E401
from partdistillation_amd.compat import BitMasks, Instances, build_model
from partdistillation_amd.config import setup_cfg

torch.backends.cudnn.benchmark = True
cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd", "configs", "proposal_generation", a.backbone + ".yaml"))
torch.manual_seed(0)
model = build_model(cfg).cuda().eval()
S = a.size
ys, xs = torch.meshgrid(torch.arange(S) / S, torch.arange(S) / S, indexing="ij")
mask = (((ys - 0.5) ** 2 / 0.13 + (xs - 0.5) ** 2 / 0.085) < 1.0)[None].float().cuda()
batch = []
for b in range(a.batch):
    inst = Instances((S, S))
    inst.gt_masks = BitMasks(mask)
    batch.append({"image": (torch.rand(3, S, S, device="cuda") * 255), "instances": inst, "file_name": f"{b}.pth", "class_code": "n0"})
g = torch.Generator(device="cuda").manual_seed(0)
model.kmeans_generator = g


def run():
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bool(a.amp)):
        return model(batch)


for _ in range(3):
    res = run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    res = run()
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / a.iters
# backbone alone
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=bool(a.amp)):
    x = torch.stack([(b["image"] - model.pixel_mean) / model.pixel_std for b in batch])
    for _ in range(2):
        model.backbone(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        f = model.backbone(x)
    torch.cuda.synchronize()
    t_backbone = (time.perf_counter() - t0) / a.iters
c3, c4 = f["res3"].shape[1], f["res4"].shape[1]
h3, h4 = f["res3"].shape[-1], f["res4"].shape[-1]
n_obj = int(res[0]["kmeans_iterations"]) and int((torch.nn.functional.interpolate(mask[None], size=(h3, h3), mode="nearest") > 0).sum())
iters = sum(r["kmeans_iterations"] for r in res) / len(res)
alg = 4 * (c3 * h3 * h3 + c4 * h4 * h4) + iters * n_obj * (c3 + c4) * 4 + S * S + S * S     # bytes per image
t_group = max(total - t_backbone, 1e-9)
print(json.dumps({"workload": f"pixel-grouping proposal generation, {a.backbone}, {a.batch} x {S}^2, res3+res4 (C={c3 + c4}), dot, K=4",
                  "images_per_s": a.batch / total, "ms_per_batch": total * 1e3, "backbone_ms": t_backbone * 1e3,
                  "grouping_ms_per_image": t_group * 1e3 / a.batch, "kmeans_iterations_avg": iters, "object_feature_pixels": n_obj,
                  "grouping_alg_MB_per_image": alg / 1e6, "grouping_GBps": alg * a.batch / t_group / 1e9,
                  "reference_dense_bytes_per_image_GB": (c3 + c4) * S * S * 4 / 1e9, "labels_present": [r["present_labels"] for r in res]}))
