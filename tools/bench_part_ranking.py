"""Measurement of the part-ranking stage on one GPU: R50 part-proposal network, B synthetic 1024^2 images
with 4 part masks each; (1) the "cluster" pass (forward + proposal extraction with features), (2) the clustering of the
pooled features of `--classes` object classes into 8 centroids each on the device, (3) the labelling pass with the
nearest-centroid classifier.  Prints one JSON line."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--classes", type=int, default=100)
ap.add_argument("--per-class", type=int, default=400)
a = ap.parse_args()
from partdistillation_amd import lib
lib.load()
import partdistillation_amd.modeling, partdistillation_amd.part_ranking_model  # noqa: F401,E401
from partdistillation_amd.compat import META_ARCH_REGISTRY, BitMasks, Instances
from partdistillation_amd.config import setup_cfg
from partdistillation_amd.engine.synthetic import make_batch
from partdistillation_amd.evaluation import ClusteringModule

torch.backends.cudnn.benchmark = True
cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd", "configs", "proposal_learning", "r50_mask2former.yaml"),
                ["MODEL.META_ARCHITECTURE", "PartRankingModel", "INPUT.IMAGE_SIZE", str(a.size)])
torch.manual_seed(0)
model = META_ARCH_REGISTRY.get("PartRankingModel")(cfg).cuda().eval()
model.fg_score_threshold = -1.0                                     # random weights: every proposal counts
batch = make_batch(a.batch, a.size, n_parts=4, seed=3, device="cuda")
for x in batch:
    parts = x["instances"]
    obj = Instances(parts.image_size)
    obj.gt_masks, obj.gt_classes = BitMasks(parts.gt_masks.tensor.any(0, keepdim=True)), torch.tensor([7], device="cuda")
    parts.gt_classes = torch.arange(len(parts), device="cuda")
    x["part_instances"], x["instances"] = parts, obj


def timed(f, n):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, r


def forward():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return model(batch)


model.mode = "cluster"
t_cluster, res = timed(forward, a.iters)
cm = ClusteringModule(distributed=False, num_clusters=8)
g = torch.Generator(device="cuda").manual_seed(0)
feats = torch.nn.functional.normalize(torch.randn(a.classes * a.per_class, 256, device="cuda", generator=g), dim=-1)
labels = torch.arange(a.classes, device="cuda").repeat_interleave(a.per_class)
cm.process(None, [{"proposal_features": feats, "gt_label": labels}])
torch.cuda.synchronize()
t0 = time.perf_counter()
cents = cm.evaluate()
torch.cuda.synchronize()
t_kmeans = time.perf_counter() - t0
model.register_classifier({7: cents[0]})
model.mode = ""
t_label, res = timed(forward, a.iters)
print(json.dumps({"workload": f"part ranking, R50 part-proposal network, {a.batch} x {a.size}^2, 4 parts per image, bf16 autocast, 1 GPU",
                  "cluster_pass_images_per_s": a.batch / t_cluster, "labelling_pass_images_per_s": a.batch / t_label,
                  "kmeans": {"object_classes": a.classes, "features_per_class": a.per_class, "clusters": 8, "dim": 256,
                             "seconds": t_kmeans, "classes_per_s": a.classes / t_kmeans}}))
