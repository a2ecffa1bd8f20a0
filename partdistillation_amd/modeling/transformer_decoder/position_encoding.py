"""Sine 2-D positional embedding (reference position_encoding.py:33-56).

With no padding mask (the only way the hot path calls it, msdeformattn.py:326,
mask2former_transformer_decoder.py:382) the embedding depends on (H, W) only,
so it is computed once per shape and cached on the device instead of being
re-derived with cumsum/sin/cos six times per step."""
import math

import torch
from torch import nn


def sine_embedding(h, w, num_pos_feats, temperature=10000.0, scale=2 * math.pi, device=None):
    """[2*num_pos_feats, h, w] fp32; normalize=True semantics (eps 1e-6)."""
    eps = 1e-6
    y = torch.arange(1, h + 1, dtype=torch.float32, device=device).view(h, 1).expand(h, w)
    x = torch.arange(1, w + 1, dtype=torch.float32, device=device).view(1, w).expand(h, w)
    y = y / (float(h) + eps) * scale
    x = x / (float(w) + eps) * scale
    i = torch.arange(num_pos_feats, dtype=torch.float32, device=device)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / num_pos_feats)
    px, py = x[:, :, None] / dim_t, y[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2).permute(2, 0, 1).contiguous()


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale
        self._cache = {}

    def table(self, h, w, device):
        """cached [C, h, w] table (normalize=True, no mask)."""
        key = (h, w, str(device))
        if key not in self._cache:
            self._cache[key] = sine_embedding(h, w, self.num_pos_feats, float(self.temperature), self.scale, device)
        return self._cache[key]

    def forward(self, x, mask=None):
        if mask is None and self.normalize:
            return self.table(x.size(2), x.size(3), x.device).unsqueeze(0).expand(x.size(0), -1, -1, -1)
        # general (masked / un-normalised) form, reference :33-56
        if mask is None:
            mask = torch.zeros((x.size(0), x.size(2), x.size(3)), device=x.device, dtype=torch.bool)
        not_mask = ~mask
        y_embed = not_mask.cumsum(1, dtype=torch.float32)
        x_embed = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            y_embed = y_embed / (y_embed[:, -1:, :] + 1e-6) * self.scale
            x_embed = x_embed / (x_embed[:, :, -1:] + 1e-6) * self.scale
        i = torch.arange(self.num_pos_feats, dtype=torch.float32, device=x.device)
        dim_t = self.temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / self.num_pos_feats)
        px, py = x_embed[..., None] / dim_t, y_embed[..., None] / dim_t
        px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
        py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
        return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)

    def __repr__(self, _repr_indent=4):
        return (f"Positional encoding {self.__class__.__name__}\n    num_pos_feats: {self.num_pos_feats}\n"
                f"    temperature: {self.temperature}\n    normalize: {self.normalize}\n    scale: {self.scale}")
