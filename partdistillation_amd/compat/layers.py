"""detectron2.layers / fvcore pieces used by the hot path:
Conv2d(conv -> norm -> activation), get_norm, FrozenBatchNorm2d, c2_xavier_fill."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm with fixed statistics and affine (buffers, never trained); eps 1e-5."""
    _version = 3

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

        self._cache = None

    def scale_bias(self):
        """(scale, bias) of the equivalent per-channel affine; the four buffers are frozen, so the pair is computed
        once and reused (invalidated when the buffers are re-loaded or moved)."""
        c = self._cache
        if c is None or c[0].device != self.weight.device or c[2] != self.weight._version + self.running_var._version:
            scale = self.weight * (self.running_var + self.eps).rsqrt()
            c = (scale, self.bias - self.running_mean * scale, self.weight._version + self.running_var._version)
            self._cache = c
        return c[0], c[1]

    def _load_from_state_dict(self, *args, **kwargs):
        self._cache = None
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, recurse=True):
        self._cache = None
        return super()._apply(fn, recurse)

    def forward(self, x):
        scale, bias = self.scale_bias()
        return x * scale.to(x.dtype).view(1, -1, 1, 1) + bias.to(x.dtype).view(1, -1, 1, 1)


def get_norm(norm, out_channels):
    if norm is None or (isinstance(norm, str) and len(norm) == 0):
        return None
    if isinstance(norm, str):
        return {"GN": lambda c: nn.GroupNorm(32, c), "FrozenBN": FrozenBatchNorm2d,
                "BN": nn.BatchNorm2d, "LN": lambda c: nn.GroupNorm(1, c)}[norm](out_channels)
    return norm(out_channels)


class Conv2d(nn.Conv2d):
    def __init__(self, *args, norm=None, activation=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def c2_xavier_fill(module):
    nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def c2_msra_fill(module):
    nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)
