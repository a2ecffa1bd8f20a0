"""``build_pixel_decoder`` (reference pixel_decoder/fpn.py:25-37) and ``BasePixelDecoder`` (:42-163), the plain FPN pixel
decoder that is the reference's config default (config.py:76) — BASELINE config 1 says "R50-FPN".

Module names / state_dict keys as the reference (``adapter_{1,2,3}``, ``layer_{1..4}``,
``mask_features`` with a 3 x 3 kernel; no conv bias when a norm is configured).  Differences to the MSDeformAttn decoder
that matter for parity: NEAREST top-down upsampling (:153), a 3 x 3 ``mask_features`` convolution, no transformer encoder
(``forward_features`` returns ``None`` in its place), and it runs under the caller's autocast (no fp32 pin).  On the GPU in
fp32 conv -> GroupNorm(32) -> ReLU goes through the channels-last HIP GroupNorm and the 3 x 3 convolutions through the
fp32-accurate bf16-matrix-core implicit GEMM of functions/conv_x3.py, exactly as in the MSDeformAttn decoder's FPN level."""
from typing import Callable, Dict, Optional, Union

import torch.nn.functional as F
from torch import nn

from ...compat import SEM_SEG_HEADS_REGISTRY, ShapeSpec, configurable
from ...compat.layers import Conv2d, c2_xavier_fill, get_norm


def build_pixel_decoder(cfg, input_shape):
    name = cfg.MODEL.SEM_SEG_HEAD.PIXEL_DECODER_NAME
    model = SEM_SEG_HEADS_REGISTRY.get(name)(cfg, input_shape)
    if not callable(getattr(model, "forward_features", None)):
        raise ValueError("Only SEM_SEG_HEADS with forward_features method can be used as pixel decoder. "
                         f"Please implement forward_features for {name} to only return mask features.")
    return model


def _conv_norm_act(conv, x):
    """detectron2 Conv2d (conv -> norm -> activation) with the HIP kernels where they apply."""
    from .msdeformattn import _conv_gn
    if isinstance(conv.norm, nn.GroupNorm) and x.is_cuda:
        return _conv_gn(conv, conv.norm, x, relu=conv.activation is F.relu)
    return conv(x)


@SEM_SEG_HEADS_REGISTRY.register()
class BasePixelDecoder(nn.Module):
    @configurable
    def __init__(self, input_shape: Dict[str, ShapeSpec], *, conv_dim: int, mask_dim: int,
                 norm: Optional[Union[str, Callable]] = None):
        super().__init__()
        shapes = sorted(input_shape.items(), key=lambda kv: kv[1].stride)
        self.in_features = [k for k, _ in shapes]                      # "res2" .. "res5"
        lateral_convs, output_convs = [], []
        use_bias = norm == ""
        for idx, (_, spec) in enumerate(shapes):
            last = idx == len(shapes) - 1
            out_in = spec.channels if last else conv_dim
            output_conv = Conv2d(out_in, conv_dim, kernel_size=3, stride=1, padding=1, bias=use_bias, norm=get_norm(norm, conv_dim),
                                 activation=F.relu)
            c2_xavier_fill(output_conv)
            lateral_conv = None
            if not last:
                lateral_conv = Conv2d(spec.channels, conv_dim, kernel_size=1, bias=use_bias, norm=get_norm(norm, conv_dim))
                c2_xavier_fill(lateral_conv)
                self.add_module(f"adapter_{idx + 1}", lateral_conv)
            self.add_module(f"layer_{idx + 1}", output_conv)
            lateral_convs.append(lateral_conv)
            output_convs.append(output_conv)
        # top-down order (low to high resolution)
        self.lateral_convs, self.output_convs = lateral_convs[::-1], output_convs[::-1]
        self.mask_dim = mask_dim
        self.mask_features = Conv2d(conv_dim, mask_dim, kernel_size=3, stride=1, padding=1)
        c2_xavier_fill(self.mask_features)
        self.maskformer_num_feature_levels = 3                          # always 3 scales

    @classmethod
    def from_config(cls, cfg, input_shape: Dict[str, ShapeSpec]):
        h = cfg.MODEL.SEM_SEG_HEAD
        return dict(input_shape={k: v for k, v in input_shape.items() if k in h.IN_FEATURES}, conv_dim=h.CONVS_DIM,
                    mask_dim=h.MASK_DIM, norm=h.NORM)

    def forward_features(self, features):
        multi_scale, y = [], None
        for idx, f in enumerate(self.in_features[::-1]):
            x = features[f]
            lateral, output = self.lateral_convs[idx], self.output_convs[idx]
            if lateral is None:
                y = _conv_norm_act(output, x)
            else:
                cur = _conv_norm_act(lateral, x)
                y = _conv_norm_act(output, cur + F.interpolate(y, size=cur.shape[-2:], mode="nearest"))   # FPN: nearest (:153)
            if len(multi_scale) < self.maskformer_num_feature_levels:
                multi_scale.append(y)
        return self.mask_features(y), None, multi_scale

    def forward(self, features, targets=None):
        return self.forward_features(features)
