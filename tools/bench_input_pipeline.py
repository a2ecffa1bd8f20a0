"""Measurement of the device input pipeline: ImageNet-sized decoded images (500 x 375, 4 pseudo-label
masks as COCO RLE) -> 1024^2 LSJ-augmented training inputs.  Device path = partdistillation_amd.data.DeviceProposalMapper
(host draws + tables, 3 kernels); CPU path = the same steps with Pillow (what detectron2's transforms call) and a numpy
RLE decode, one process.  Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from PIL import Image
from partdistillation_amd import lib
lib.load()
from partdistillation_amd.data import DeviceProposalMapper
from partdistillation_amd.utils import rle

S, H, W, N = 1024, 375, 500, 4
rng = np.random.RandomState(0)
imgs, annos = [], []
for i in range(16):
    img = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
    ys, xs = np.mgrid[0:H, 0:W]
    seeds = rng.rand(N, 2) * [H, W]
    lab = np.argmin((ys[None] - seeds[:, 0, None, None]) ** 2 + (xs[None] - seeds[:, 1, None, None]) ** 2, axis=0)
    inside = ((ys - H / 2) ** 2 / (0.17 * H * H) + (xs - W / 2) ** 2 / (0.12 * W * W)) < 1.0
    imgs.append(img)
    annos.append([{"segmentation": rle.encode((lab == k) & inside), "category_id": 0} for k in range(N)])
mapper = DeviceProposalMapper(S, 0.1, 2.0, "relative_range", (0.9, 0.9), rng=np.random.RandomState(1))


def device_pass(n):
    for i in range(n):
        out = mapper({"image": imgs[i % 16], "pseudo_annotations": annos[i % 16]})
    torch.cuda.synchronize()
    return out


def cpu_one(img, ann, p):
    masks = np.stack([rle.decode(a["segmentation"]) for a in ann]).astype(np.uint8)
    if p["flip"]:
        img, masks = img[:, ::-1], masks[:, :, ::-1]
    x0, y0, cw, ch = p["crop1"]
    img, masks = img[y0:y0 + ch, x0:x0 + cw], masks[:, y0:y0 + ch, x0:x0 + cw]
    rh, rw = p["resize"]
    img = np.asarray(Image.fromarray(np.ascontiguousarray(img)).resize((rw, rh), Image.BILINEAR))
    masks = np.stack([np.asarray(Image.fromarray(np.ascontiguousarray(m)).resize((rw, rh), Image.NEAREST)) for m in masks])
    ox, oy = p["crop2"]
    img, masks = img[oy:oy + S, ox:ox + S], masks[:, oy:oy + S, ox:ox + S]
    out = np.full((S, S, 3), 128, np.uint8)
    out[:img.shape[0], :img.shape[1]] = img
    om = np.zeros((N, S, S), bool)
    om[:, :masks.shape[1], :masks.shape[2]] = masks
    return torch.as_tensor(np.ascontiguousarray(out.transpose(2, 0, 1))), torch.as_tensor(om)


device_pass(8)
t0 = time.perf_counter(); device_pass(200); t_dev = (time.perf_counter() - t0) / 200
r2 = np.random.RandomState(1)
m2 = DeviceProposalMapper(S, 0.1, 2.0, "relative_range", (0.9, 0.9), device="cpu", rng=r2)
t0 = time.perf_counter()
for i in range(60):
    cpu_one(imgs[i % 16], annos[i % 16], m2.draw(H, W))
t_cpu = (time.perf_counter() - t0) / 60
print(json.dumps({"workload": f"input pipeline: {W}x{H} image + {N} RLE masks -> {S}^2 (flip, crop, scale 0.1-2.0, crop, pad)",
                  "device_images_per_s": 1 / t_dev, "device_ms_per_image": t_dev * 1e3,
                  "cpu_pillow_numpy_images_per_s_one_core": 1 / t_cpu, "cpu_ms_per_image": t_cpu * 1e3}))
