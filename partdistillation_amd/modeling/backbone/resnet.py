"""ResNet backbone with the detectron2 0.6 ``build_resnet_backbone`` surface
(SURVEY §8a row a3, Appendix D): BasicStem 7x7/2 + maxpool, bottleneck stages
[3,4,6,3] (R50) / [3,4,23,3] (R101), stride on the 3x3 unless STRIDE_IN_1X1,
FrozenBN, outputs res2..res5; state_dict keys ``stem.conv1.*``,
``res{2..5}.{i}.{conv1,conv2,conv3,shortcut}.{weight,norm.*}``.

The reference takes this class from detectron2 (un-vendored): "parity unpinned"
by the reference's tests; pinned against oracle/step_ref.py::resnet50_forward.
Runs channels_last so MIOpen sees NHWC; the frozen-BN affine is folded into
the conv epilogue as (scale-folded weight, bias) — one conv call per layer.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...compat import BACKBONE_REGISTRY, ShapeSpec
from ...compat.layers import Conv2d, FrozenBatchNorm2d, c2_msra_fill, get_norm
from ...functions import conv_bf16
from ...functions import stem as _stem
from ...functions.fused import affine_act, max_pool3x3s2, max_pool3x3s2_supported

OWN_MAXPOOL = bool(int(__import__("os").environ.get("PD_OWN_MAXPOOL", "1")))    # the stem's max pooling on pd_maxpool3s2_{fwd,bwd}_bf16 (0: ATen)
OWN_STEM = bool(int(__import__("os").environ.get("PD_OWN_STEM", "1")))          # the 7 x 7 stem on pd_stem7x7_{fwd,wgrad} (0: MIOpen + pd_affine_act)
OWN_WGRAD = bool(int(__import__("os").environ.get("PD_CONV_OWN_WGRAD", "1")))   # 0: MIOpen's filter gradients (tools/ comparisons)


def _conv_bn_act(conv, x, residual=None, relu=True, fork=False):
    """conv -> norm (-> + residual) (-> ReLU).

    bf16 autocast on the GPU with a frozen norm (the training configuration): the convolution runs bias-free on the
    (bf16) filter and ONE HIP kernel applies the frozen-BN affine, the residual add and the ReLU to the NHWC output
    (functions/fused.py::affine_act) — instead of the fold-multiply, cast, bias pass, add pass and ReLU pass of the
    unfused chain.  Otherwise (fp32 parity runs, trainable norms): the same arithmetic with torch ops, FrozenBN folded
    into the filter."""
    norm = conv.norm
    frozen = isinstance(norm, FrozenBatchNorm2d)
    if frozen and x.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16 \
            and conv.out_channels % 8 == 0:
        if OWN_WGRAD and conv_bf16.own_wgrad_supported(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups):
            y = conv_bf16.conv2d_own_wgrad(x, conv.weight, conv.stride, conv.padding)     # filter gradient: csrc/conv_bf16.hip
        else:
            y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
        scale, bias = norm.scale_bias()
        return affine_act(y, scale, bias, residual, relu, fork=fork)
    if frozen:
        scale, bias = norm.scale_bias()
        w = conv.weight * scale.view(-1, 1, 1, 1).to(conv.weight.dtype)
        y = F.conv2d(x, w, bias.to(conv.weight.dtype), conv.stride, conv.padding, conv.dilation, conv.groups)
    else:
        y = F.conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation, conv.groups)
        y = norm(y) if norm is not None else y
    if residual is not None:
        y = y + residual
    y = F.relu_(y) if relu else y
    return (y, y) if fork else y


class BasicStem(nn.Module):
    def __init__(self, in_channels=3, out_channels=64, norm="FrozenBN"):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, 4
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=7, stride=2, padding=3, bias=False,
                            norm=get_norm(norm, out_channels))
        c2_msra_fill(self.conv1)

    def forward(self, x):
        if OWN_STEM and _stem.supported(x, self.conv1):
            x = _stem.stem_conv(x, self.conv1)                      # convolution + frozen-BN affine + ReLU: one launch (include/pd_stem.h)
        else:
            x = _conv_bn_act(self.conv1, x)
        if OWN_MAXPOOL and max_pool3x3s2_supported(x):
            return max_pool3x3s2(x)
        return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)


class BottleneckBlock(nn.Module):
    def __init__(self, in_channels, out_channels, *, bottleneck_channels, stride=1, num_groups=1, norm="FrozenBN",
                 stride_in_1x1=False, dilation=1):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        self.shortcut = None
        if in_channels != out_channels:
            self.shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=stride, bias=False,
                                   norm=get_norm(norm, out_channels))
        s1, s3 = (stride, 1) if stride_in_1x1 else (1, stride)
        self.conv1 = Conv2d(in_channels, bottleneck_channels, kernel_size=1, stride=s1, bias=False,
                            norm=get_norm(norm, bottleneck_channels))
        self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, kernel_size=3, stride=s3, padding=dilation,
                            bias=False, groups=num_groups, dilation=dilation, norm=get_norm(norm, bottleneck_channels))
        self.conv3 = Conv2d(bottleneck_channels, out_channels, kernel_size=1, bias=False,
                            norm=get_norm(norm, out_channels))
        for layer in (self.conv1, self.conv2, self.conv3, self.shortcut):
            if layer is not None:
                c2_msra_fill(layer)

    def forward(self, x, x_sc=None, fork=False):
        """x_sc: an alias of x for the shortcut branch (see functions/fused.py AffineActFork: the previous block hands its output out
        twice so that the two gradients meet inside its epilogue kernel); fork: hand THIS block's output out twice -> (out, alias)"""
        x_sc = x if x_sc is None else x_sc
        out = _conv_bn_act(self.conv1, x)
        out = _conv_bn_act(self.conv2, out)
        sc = _conv_bn_act(self.shortcut, x_sc, relu=False) if self.shortcut is not None else x_sc
        return _conv_bn_act(self.conv3, out, residual=sc, relu=True, fork=fork)


class ResNet(nn.Module):
    def __init__(self, stem, stages, out_features):
        super().__init__()
        self.stem = stem
        self._out_features = list(out_features)
        self._out_feature_strides, self._out_feature_channels = {"stem": stem.stride}, {"stem": stem.out_channels}
        self.stage_names = []
        cur = stem.stride
        for i, blocks in enumerate(stages):
            name = "res" + str(i + 2)
            self.add_module(name, nn.Sequential(*blocks))
            self.stage_names.append(name)
            cur = cur * blocks[0].stride
            self._out_feature_strides[name] = cur
            self._out_feature_channels[name] = blocks[-1].out_channels

    @property
    def size_divisibility(self):
        return 0

    def forward(self, x):
        assert x.dim() == 4, f"ResNet takes an input of shape (N, C, H, W). Got {x.shape} instead!"
        x = x.contiguous(memory_format=torch.channels_last)
        outputs = {}
        x = self.stem(x)
        if "stem" in self._out_features:
            outputs["stem"] = x
        blocks = [(name, blk) for name in self.stage_names for blk in getattr(self, name)]
        from . import resnet_core
        if resnet_core.supported([b for _, b in blocks], x):
            # res2 .. res5 as ONE autograd node on the fused implicit-GEMM kernels (resnet_core.py): 52 launches per direction from C++
            outputs.update(resnet_core.run_body(self, x, [b for _, b in blocks], [n for n, _ in blocks]))
            return {k: v for k, v in outputs.items() if k in self._out_features}
        x_sc = None
        for i, (name, blk) in enumerate(blocks):
            last = i + 1 == len(blocks)
            if isinstance(blk, BottleneckBlock):
                out = blk(x, x_sc, fork=not last)                    # the output's two consumers: next block's conv1 and shortcut
                x, x_sc = (out, None) if last else out
            else:
                x, x_sc = blk(x), None
            if (last or blocks[i + 1][0] != name) and name in self._out_features:
                outputs[name] = x
        return outputs

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n])
                for n in self._out_features}

    def freeze(self, freeze_at=0):
        if freeze_at >= 1:
            for p in self.stem.parameters():
                p.requires_grad = False
        for idx, name in enumerate(self.stage_names, start=2):
            if freeze_at >= idx:
                for p in getattr(self, name).parameters():
                    p.requires_grad = False
        return self


@BACKBONE_REGISTRY.register()
def build_resnet_backbone(cfg, input_shape):
    r = cfg.MODEL.RESNETS
    norm = r.NORM
    stem = BasicStem(in_channels=input_shape.channels, out_channels=r.STEM_OUT_CHANNELS, norm=norm)
    depth = r.DEPTH
    blocks_per_stage = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}[depth]
    out_features = r.OUT_FEATURES
    last = max({"res2": 2, "res3": 3, "res4": 4, "res5": 5}[f] for f in out_features)
    in_c, out_c, bott = r.STEM_OUT_CHANNELS, r.RES2_OUT_CHANNELS, r.NUM_GROUPS * r.WIDTH_PER_GROUP
    stages = []
    for idx, stage in enumerate(range(2, last + 1)):
        dilation = r.RES5_DILATION if stage == 5 else 1
        first_stride = 1 if idx == 0 or (stage == 5 and dilation == 2) else 2
        blocks = []
        for b in range(blocks_per_stage[idx]):
            blocks.append(BottleneckBlock(in_c, out_c, bottleneck_channels=bott, stride=first_stride if b == 0 else 1,
                                          num_groups=r.NUM_GROUPS, norm=norm, stride_in_1x1=r.STRIDE_IN_1X1,
                                          dilation=dilation))
            in_c = out_c
        stages.append(blocks)
        out_c, bott = out_c * 2, bott * 2
    return ResNet(stem, stages, out_features).freeze(cfg.MODEL.BACKBONE.FREEZE_AT)
