"""MSDeformAttn module (reference ops/modules/ms_deform_attn.py:37-131): same
parameters / init / forward signature; the sampling core is the HIP operator
(the reference falls back to grid_sample because its import is commented out,
:28,121-127 — here the native op is the only path)."""
import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, xavier_uniform_

from .....functions.gemm import linear_f32
from ..functions.ms_deform_attn_func import MSDeformAttnFunction


def _is_power_of_2(n):
    if (not isinstance(n, int)) or (n < 0):
        raise ValueError(f"invalid input for _is_power_of_2: {n} (type: {type(n)})")
    return (n & (n - 1) == 0) and n != 0


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError(f"d_model must be divisible by n_heads, but got {d_model} and {n_heads}")
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("MSDeformAttn: head dim that is not a power of 2 takes the generic HIP path")
        self.im2col_step = 128
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        constant_(self.sampling_offsets.weight.data, 0.0)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2)
        grid = grid.repeat(1, self.n_levels, self.n_points, 1)
        grid = grid * torch.arange(1, self.n_points + 1, dtype=torch.float32).view(1, 1, -1, 1)
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid.reshape(-1))
        constant_(self.attention_weights.weight.data, 0.0)
        constant_(self.attention_weights.bias.data, 0.0)
        xavier_uniform_(self.value_proj.weight.data)
        constant_(self.value_proj.bias.data, 0.0)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.0)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        N, Len_q, _ = query.shape
        N, Len_in, _ = input_flatten.shape
        lin = self._linear
        value = lin(self.value_proj, input_flatten)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], float(0))
        value = value.view(N, Len_in, self.n_heads, self.d_model // self.n_heads)
        offsets = lin(self.sampling_offsets, query).view(N, Len_q, self.n_heads, self.n_levels, self.n_points, 2)
        weights = lin(self.attention_weights, query).view(N, Len_q, self.n_heads, self.n_levels * self.n_points)
        weights = F.softmax(weights, -1).view(N, Len_q, self.n_heads, self.n_levels, self.n_points)
        if reference_points.shape[-1] == 2:
            normalizer = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
            locations = reference_points[:, :, None, :, None, :] + offsets / normalizer[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            locations = reference_points[:, :, None, :, None, :2] \
                + offsets / self.n_points * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError(f"Last dim of reference_points must be 2 or 4, but get {reference_points.shape[-1]} instead.")
        output = MSDeformAttnFunction.apply(value.contiguous(), input_spatial_shapes, input_level_start_index,
                                            locations.contiguous(), weights.contiguous(), self.im2col_step)
        return lin(self.output_proj, output)

    @staticmethod
    def _linear(layer, x):
        """fp32 tokens x weights: hand-written MFMA weight-gradient GEMM (functions/gemm.py); other dtypes -> torch."""
        if x.dtype == torch.float32 and layer.weight.dtype == torch.float32 and x.is_cuda and not torch.is_autocast_enabled():
            return linear_f32(x, layer.weight, layer.bias)
        return layer(x)
