"""Helpers of reference utils/misc.py used on the hot path (:52-74, :110-115)."""
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import Tensor


class NestedTensor:
    def __init__(self, tensors, mask: Optional[Tensor]):
        self.tensors, self.mask = tensors, mask

    def to(self, device):
        return NestedTensor(self.tensors.to(device), None if self.mask is None else self.mask.to(device))

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


def nested_tensor_from_tensor_list(tensor_list: List[Tensor]) -> NestedTensor:
    """pad a list of [n_i,H_i,W_i] to [B, n_max, H_max, W_max] + bool padding mask."""
    if tensor_list[0].ndim != 3:
        raise ValueError("not supported")
    dims = [max(t.shape[d] for t in tensor_list) for d in range(3)]
    b = len(tensor_list)
    out = torch.zeros([b] + dims, dtype=tensor_list[0].dtype, device=tensor_list[0].device)
    mask = torch.ones((b, dims[1], dims[2]), dtype=torch.bool, device=tensor_list[0].device)
    for img, pad, m in zip(tensor_list, out, mask):
        pad[: img.shape[0], : img.shape[1], : img.shape[2]].copy_(img)
        m[: img.shape[1], : img.shape[2]] = False
    return NestedTensor(out, mask)


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def collectives_active():
    """do the data-parallel collectives run?  More than one rank — or ONE rank with PD_DDP_FORCE=1, which sends every collective of the step
    (bucket all-reduces, the row-sparse exchange, the num_masks all-reduce) through the backend with a single participant: how the RCCL
    call sequence is exercised on a one-GPU box (tests/test_ddp_gpu.py); the results equal the plain single-process step's"""
    import os
    return is_dist_avail_and_initialized() and (dist.get_world_size() > 1 or os.environ.get("PD_DDP_FORCE", "0") == "1")


def point_sample(input, point_coords, **kwargs):
    """detectron2 point_sample: coords in [0,1]x[0,1] (x,y)."""
    import torch.nn.functional as F
    add_dim = point_coords.dim() == 3
    if add_dim:
        point_coords = point_coords.unsqueeze(2)
    out = F.grid_sample(input, 2.0 * point_coords - 1.0, **kwargs)
    return out.squeeze(3) if add_dim else out


def get_uncertain_point_coords_with_randomness(coarse_logits, uncertainty_func, num_points, oversample_ratio,
                                               importance_sample_ratio, rand=None):
    """detectron2 PointRend sampler; ``rand(shape)`` overrides torch.rand for replay."""
    assert oversample_ratio >= 1 and 0 <= importance_sample_ratio <= 1
    rand = rand or (lambda shape: torch.rand(shape, device=coarse_logits.device))
    n = coarse_logits.shape[0]
    num_sampled = int(num_points * oversample_ratio)
    coords = rand((n, num_sampled, 2)).to(coarse_logits.device)
    unc = uncertainty_func(point_sample(coarse_logits, coords, align_corners=False))
    k = int(importance_sample_ratio * num_points)
    idx = torch.topk(unc[:, 0, :], k=k, dim=1)[1]
    idx = idx + num_sampled * torch.arange(n, dtype=torch.long, device=coarse_logits.device)[:, None]
    coords = coords.view(-1, 2)[idx.view(-1), :].view(n, k, 2)
    if num_points - k > 0:
        coords = torch.cat([coords, rand((n, num_points - k, 2)).to(coarse_logits.device)], dim=1)
    return coords
