"""Seeded synthetic batches of the shape the training step consumes (SURVEY
§8a-1 / §8d): per image a uint8 ``[3,S,S]`` picture and ``n`` disjoint part
masks that partition a centred ellipse (~35 % of the image) by the Voronoi
cells of ``n`` random seeds — the look of the K-means (K=4) pseudo-labels the
reference trains on.  Generated directly on the target device."""
import math

import torch

from ..compat.structures import BitMasks, Instances


def make_batch(batch_size, size, *, n_parts=4, seed=1234, device="cuda", part_distillation=False, num_part_classes=8,
               num_object_classes=1000):
    g = torch.Generator(device="cpu").manual_seed(seed)
    dev = torch.device(device)
    ys, xs = torch.meshgrid(torch.arange(size, device=dev, dtype=torch.float32) / size,
                            torch.arange(size, device=dev, dtype=torch.float32) / size, indexing="ij")
    # ellipse of area ~0.35: pi*a*b = 0.35 with a/b = 1.3
    b_ax = math.sqrt(0.35 / (math.pi * 1.3))
    a_ax = 1.3 * b_ax
    inside = ((ys - 0.5) / b_ax) ** 2 + ((xs - 0.5) / a_ax) ** 2 < 1.0
    out = []
    for _ in range(batch_size):
        image = torch.randint(0, 256, (3, size, size), generator=g, dtype=torch.uint8).to(dev)
        ang = torch.rand(n_parts, generator=g) * 2 * math.pi
        rad = torch.rand(n_parts, generator=g).sqrt() * 0.8
        cy = (0.5 + rad * b_ax * torch.sin(ang)).to(dev)
        cx = (0.5 + rad * a_ax * torch.cos(ang)).to(dev)
        d = (ys[None] - cy[:, None, None]) ** 2 + (xs[None] - cx[:, None, None]) ** 2
        owner = d.argmin(0)
        masks = torch.stack([(owner == k) & inside for k in range(n_parts)])
        inst = Instances((size, size))
        inst.gt_masks = BitMasks(masks)
        if part_distillation:
            inst.gt_classes = torch.randperm(num_part_classes, generator=g)[:n_parts].to(dev)
        else:
            inst.gt_classes = torch.zeros(n_parts, dtype=torch.int64, device=dev)
        out.append({"image": image, "instances": inst, "height": size, "width": size,
                    "gt_object_class": int(torch.randint(0, num_object_classes, (1,), generator=g))})
    return out
