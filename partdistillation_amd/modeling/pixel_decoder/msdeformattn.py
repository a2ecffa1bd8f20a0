"""MSDeformAttn pixel decoder (reference pixel_decoder/msdeformattn.py:27-362):
1x1+GN input projections of res3-5, a 6-layer deformable-attention encoder over
the three flattened levels, one extra FPN level on res2 and the 1x1
``mask_features`` head.  Same class names, ctor/from_config arguments,
``forward_features`` contract and state_dict keys."""
from typing import Callable, Dict, List, Optional, Union

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import normal_

from ...compat import SEM_SEG_HEADS_REGISTRY, ShapeSpec, configurable
from ...compat.layers import Conv2d, c2_xavier_fill, get_norm
from ..transformer_decoder.position_encoding import PositionEmbeddingSine
from ...functions import conv_x3
from ...functions.fused import group_norm_nhwc, group_norm_nhwc_supported
from ...functions.gemm import linear_f32
from .ops.modules import MSDeformAttn
from ...functions.encoder_core import EncoderCore, EncoderSpec
from ...functions.rowwise import supports_width, upsample_add, upsample_add_supported


class MSDeformAttnTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        assert activation == "relu"
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        q = src if pos is None else src + pos
        attn = self.self_attn(q, reference_points, src, spatial_shapes, level_start_index, padding_mask)
        src = self.norm1(src + self.dropout1(attn))
        lin = MSDeformAttn._linear
        if self.dropout2.p == 0.0 and src.dtype == torch.float32 and src.is_cuda and not torch.is_autocast_enabled():
            hidden = linear_f32(src, self.linear1.weight, self.linear1.bias, True)      # bias + ReLU fused
        else:
            hidden = self.dropout2(F.relu(self.linear1(src)))
        ffn = lin(self.linear2, hidden)
        return self.norm2(src + self.dropout3(ffn))


class MSDeformAttnTransformerEncoder(nn.Module):
    def __init__(self, encoder_layer_factory, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([encoder_layer_factory() for _ in range(num_layers)])
        self.num_layers = num_layers
        self._ref_cache = {}

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        """pixel centres / (valid_ratio * size), reference :145-157."""
        refs = []
        for lvl, (H_, W_) in enumerate(spatial_shapes):
            H_, W_ = int(H_), int(W_)
            ry, rx = torch.meshgrid(torch.linspace(0.5, H_ - 0.5, H_, dtype=torch.float32, device=device),
                                    torch.linspace(0.5, W_ - 0.5, W_, dtype=torch.float32, device=device), indexing="ij")
            ry = ry.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H_)
            rx = rx.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W_)
            refs.append(torch.stack((rx, ry), -1))
        ref = torch.cat(refs, 1)
        return ref[:, :, None] * valid_ratios[:, None]

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None,
                shapes_host=None):
        key = (tuple(map(tuple, shapes_host)), src.shape[0], str(src.device)) if shapes_host is not None else None
        if key is not None and key in self._ref_cache:
            reference_points = self._ref_cache[key]
        else:
            reference_points = self.get_reference_points(shapes_host or spatial_shapes.tolist(), valid_ratios, src.device)
            if key is not None:                      # valid_ratios == 1 on this path: geometry only
                self._ref_cache[key] = reference_points
        if self._fused_ok(src, pos, reference_points, padding_mask):
            l0 = self.layers[0]
            spec = EncoderSpec(l0.self_attn.n_heads, l0.self_attn.n_levels, l0.self_attn.n_points, l0.norm1.eps,
                               l0.self_attn.im2col_step, reference_points, spatial_shapes, level_start_index)
            return EncoderCore.apply(spec, src, pos, *self._core_params())
        out = src
        for layer in self.layers:
            out = layer(out, pos, reference_points, spatial_shapes, level_start_index, padding_mask)
        return out

    fused_core = True          # GPU fp32: run the layer loop as one hand-written autograd node (functions/encoder_core.py)

    def _fused_ok(self, src, pos, reference_points, padding_mask):
        if not (self.fused_core and self.num_layers and src.is_cuda and src.dtype == torch.float32 and pos is not None
                and padding_mask is None and reference_points.shape[-1] == 2 and not torch.is_autocast_enabled("cuda")):
            return False
        l0 = self.layers[0]
        if self.training and any(d.p > 0 for l in self.layers for d in (l.dropout1, l.dropout2, l.dropout3)):
            return False
        return (supports_width(src.shape[-1]) and l0.linear1.out_features % 128 == 0
                and all(p.dtype == torch.float32 for p in l0.parameters()))

    def _core_params(self):
        p = []
        for l in self.layers:
            a = l.self_attn
            p += [a.sampling_offsets.weight, a.sampling_offsets.bias, a.attention_weights.weight, a.attention_weights.bias,
                  a.value_proj.weight, a.value_proj.bias, a.output_proj.weight, a.output_proj.bias, l.norm1.weight, l.norm1.bias,
                  l.linear1.weight, l.linear1.bias, l.linear2.weight, l.linear2.bias, l.norm2.weight, l.norm2.bias]
        return p


class LevelPos(torch.autograd.Function):
    """pos = cat_l(pos_embed_l + level_embed[l]) over the flattened levels (reference msdeformattn.py:80-84).  The sine
    tables carry no gradient; d level_embed[l] is the column sum of d_pos over level l's tokens (pd_colsum_acc) instead
    of a split + three strided reductions."""

    @staticmethod
    def forward(ctx, level_embed, *pos_embeds):
        ctx.sizes = [int(p.shape[2] * p.shape[3]) for p in pos_embeds]
        # every level's sum written straight into its rows of the result (three add launches; adds + torch.cat were 117 us for 44 MB)
        B, C = pos_embeds[0].shape[:2]
        out = torch.empty((B, sum(ctx.sizes), C), dtype=pos_embeds[0].dtype, device=pos_embeds[0].device)
        # the sine tables are per-shape constants broadcast over the batch (PositionEmbeddingSine.table): their token-major form
        # [S, C] is kept, so a step reads contiguous rows instead of transposing three NCHW tables
        flat = None
        if all(p.stride(0) == 0 for p in pos_embeds):
            key = tuple((p.data_ptr(), tuple(p.shape)) for p in pos_embeds)
            hit = LevelPos._flat.get(key)
            if hit is None:
                if len(LevelPos._flat) >= 8:
                    LevelPos._flat.clear()
                hit = LevelPos._flat[key] = (torch.cat([p[0].flatten(1).t() for p in pos_embeds]).contiguous(), pos_embeds)   # (+ the sources: the addresses stay theirs)
            flat = hit[0]
        o = 0
        for i, (p, n) in enumerate(zip(pos_embeds, ctx.sizes)):
            a = flat[o:o + n].unsqueeze(0).expand(B, -1, -1) if flat is not None else p.flatten(2).transpose(1, 2)
            torch.add(a, level_embed[i].view(1, 1, -1), out=out[:, o:o + n])
            o += n
        return out

    _flat = {}

    @staticmethod
    def backward(ctx, d_pos):
        from ...functions.rowwise import colsum_acc
        d_pos = d_pos if d_pos.is_contiguous() else d_pos.contiguous()
        B, S, C = d_pos.shape
        d_le = torch.zeros((len(ctx.sizes), C), dtype=torch.float32, device=d_pos.device)
        start = 0
        for l, n in enumerate(ctx.sizes):
            for b in range(B):
                colsum_acc(d_pos[b, start:start + n], d_le[l])
            start += n
        return (d_le,) + (None,) * len(ctx.sizes)


class CatLevels(torch.autograd.Function):
    """the levels' tokens side by side, [B, sum h_l w_l, C] from NCHW-shaped maps: three strided copies into one buffer (torch.cat's
    batched-copy kernel takes 76 us for these 44 MB).  Its own autograd node because the backward of slice assignments
    (CopySlices) clones the whole gradient once per level (78 us); the gradients of a concatenation are just views."""

    @staticmethod
    def forward(ctx, *srcs):
        ctx.shapes = [tuple(s.shape) for s in srcs]
        B, C = srcs[0].shape[:2]
        sizes = [s.shape[2] * s.shape[3] for s in srcs]
        out = torch.empty((B, sum(sizes), C), dtype=srcs[0].dtype, device=srcs[0].device)
        o = 0
        for s_, n in zip(srcs, sizes):
            out[:, o:o + n].copy_(s_.flatten(2).transpose(1, 2))
            o += n
        return out

    @staticmethod
    def backward(ctx, g):
        outs, o = [], 0
        for (B, C, h, w) in ctx.shapes:
            outs.append(g[:, o:o + h * w].transpose(1, 2).reshape(B, C, h, w))      # a view: NHWC storage inside g
            o += h * w
        return tuple(outs)


class MSDeformAttnTransformerEncoderOnly(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, dim_feedforward=1024, dropout=0.1,
                 activation="relu", num_feature_levels=4, enc_n_points=4):
        super().__init__()
        self.d_model, self.nhead = d_model, nhead
        self.encoder = MSDeformAttnTransformerEncoder(
            lambda: MSDeformAttnTransformerEncoderLayer(d_model, dim_feedforward, dropout, activation,
                                                        num_feature_levels, nhead, enc_n_points), num_encoder_layers)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        self._shape_cache = {}
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        normal_(self.level_embed)

    def forward(self, srcs, pos_embeds):
        """srcs / pos_embeds: lists of NCHW maps, low resolution first (reference :65-93; no padding masks)."""
        shapes_host = [(int(s.shape[2]), int(s.shape[3])) for s in srcs]
        key = (tuple(shapes_host), str(srcs[0].device))
        if key not in self._shape_cache:
            sh = torch.as_tensor(shapes_host, dtype=torch.long, device=srcs[0].device)
            self._shape_cache[key] = (sh, torch.cat((sh.new_zeros((1,)), sh.prod(1).cumsum(0)[:-1])))
        spatial_shapes, level_start_index = self._shape_cache[key]
        if srcs[0].is_cuda:
            src = CatLevels.apply(*srcs)
        else:
            src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
        if src.is_cuda and src.dtype == torch.float32 and self.level_embed.dtype == torch.float32 and self.d_model % 128 == 0:
            pos = LevelPos.apply(self.level_embed, *pos_embeds)          # same values; backward = 2 column-sum launches per level
        else:
            pos = torch.cat([p.flatten(2).transpose(1, 2) + self.level_embed[i].view(1, 1, -1)
                             for i, p in enumerate(pos_embeds)], 1)
        valid_ratios = src.new_ones((src.shape[0], len(srcs), 2))
        memory = self.encoder(src, spatial_shapes, level_start_index, valid_ratios, pos, None, shapes_host=shapes_host)
        return memory, spatial_shapes, level_start_index, shapes_host


CONV_X3 = bool(int(__import__("os").environ.get("PD_CONV_X3", "1")))   # functions/conv_x3.py for the 3 x 3 FPN convolution


def _as_fp32_input(x, conv):
    """the reference's `features[f].float()` (:324, 338).  A bf16 channels-last backbone map whose consumer is the own 1 x 1 convolution stays
    bf16 here: that node makes the fp32 copy together with the row maxima it needs anyway and returns a bf16 input gradient
    (functions/conv_x3.Conv1x1OwnWgrad) — the same values, two cast launches and one row-maxima launch less per level."""
    if x.dtype == torch.bfloat16 and CONV_X3 and conv_x3.conv1x1_supported(x, conv) and x.requires_grad:
        return x
    return x.float()


def _conv_gn(conv, gn, x, relu=False):
    """conv -> GroupNorm (-> ReLU); on the GPU in fp32 the norm (+ReLU) is the channels-last HIP GroupNorm
    (functions/fused.py), so the maps stay NHWC from the backbone to the encoder tokens with no layout copies."""
    if CONV_X3 and conv_x3.supported(x, conv):
        y = conv_x3.conv3x3(x, conv.weight, conv.bias)       # 3 x 3 FPN convolution: fp32-level results on the bf16 matrix cores
    elif CONV_X3 and conv_x3.conv1x1_supported(x, conv):
        y = conv_x3.conv1x1(x, conv.weight, conv.bias)       # library forward / input gradient, split-GEMM filter gradient
    else:
        y = F.conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation, conv.groups)
    if isinstance(gn, nn.GroupNorm) and group_norm_nhwc_supported(y, gn.num_groups):
        return group_norm_nhwc(y, gn.weight, gn.bias, gn.num_groups, gn.eps, relu)
    if gn is not None:
        y = gn(y)
    return F.relu(y) if relu else y


@SEM_SEG_HEADS_REGISTRY.register()
class MSDeformAttnPixelDecoder(nn.Module):
    @configurable
    def __init__(self, input_shape: Dict[str, ShapeSpec], *, transformer_dropout: float, transformer_nheads: int,
                 transformer_dim_feedforward: int, transformer_enc_layers: int, conv_dim: int, mask_dim: int,
                 norm: Optional[Union[str, Callable]] = None, transformer_in_features: List[str], common_stride: int):
        super().__init__()
        tf_shape = {k: v for k, v in input_shape.items() if k in transformer_in_features}
        by_stride = sorted(input_shape.items(), key=lambda kv: kv[1].stride)
        self.in_features = [k for k, _ in by_stride]
        self.feature_strides = [v.stride for _, v in by_stride]
        self.feature_channels = [v.channels for _, v in by_stride]
        tf_by_stride = sorted(tf_shape.items(), key=lambda kv: kv[1].stride)
        self.transformer_in_features = [k for k, _ in tf_by_stride]
        tf_channels = [v.channels for _, v in tf_by_stride]
        self.transformer_feature_strides = [v.stride for _, v in tf_by_stride]
        self.transformer_num_feature_levels = len(self.transformer_in_features)
        # low resolution first: input_proj[0] projects res5
        chans = tf_channels[::-1] if self.transformer_num_feature_levels > 1 else [tf_channels[-1]]
        self.input_proj = nn.ModuleList(
            [nn.Sequential(nn.Conv2d(c, conv_dim, kernel_size=1), nn.GroupNorm(32, conv_dim)) for c in chans])
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)
        self.transformer = MSDeformAttnTransformerEncoderOnly(
            d_model=conv_dim, dropout=transformer_dropout, nhead=transformer_nheads,
            dim_feedforward=transformer_dim_feedforward, num_encoder_layers=transformer_enc_layers,
            num_feature_levels=self.transformer_num_feature_levels)
        self.pe_layer = PositionEmbeddingSine(conv_dim // 2, normalize=True)
        self.mask_dim = mask_dim
        self.mask_features = Conv2d(conv_dim, mask_dim, kernel_size=1, stride=1, padding=0)
        c2_xavier_fill(self.mask_features)
        self.maskformer_num_feature_levels = 3
        self.common_stride = common_stride
        self.num_fpn_levels = int(np.log2(min(self.transformer_feature_strides)) - np.log2(self.common_stride))
        lateral_convs, output_convs = [], []
        use_bias = norm == ""
        for idx, in_channels in enumerate(self.feature_channels[: self.num_fpn_levels]):
            lateral = Conv2d(in_channels, conv_dim, kernel_size=1, bias=use_bias, norm=get_norm(norm, conv_dim))
            output = Conv2d(conv_dim, conv_dim, kernel_size=3, stride=1, padding=1, bias=use_bias,
                            norm=get_norm(norm, conv_dim), activation=F.relu)
            c2_xavier_fill(lateral)
            c2_xavier_fill(output)
            self.add_module(f"adapter_{idx + 1}", lateral)
            self.add_module(f"layer_{idx + 1}", output)
            lateral_convs.append(lateral)
            output_convs.append(output)
        self.lateral_convs = lateral_convs[::-1]          # top-down order
        self.output_convs = output_convs[::-1]

    @classmethod
    def from_config(cls, cfg, input_shape: Dict[str, ShapeSpec]):
        head = cfg.MODEL.SEM_SEG_HEAD
        return dict(
            input_shape={k: v for k, v in input_shape.items() if k in head.IN_FEATURES},
            conv_dim=head.CONVS_DIM, mask_dim=head.MASK_DIM, norm=head.NORM,
            transformer_dropout=cfg.MODEL.MASK_FORMER.DROPOUT, transformer_nheads=cfg.MODEL.MASK_FORMER.NHEADS,
            transformer_dim_feedforward=1024,             # hard-coded in the reference (:310)
            transformer_enc_layers=head.TRANSFORMER_ENC_LAYERS,
            transformer_in_features=head.DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES,
            common_stride=head.COMMON_STRIDE)

    def forward_features(self, features):
        """-> (mask_features, lowest-resolution encoder map, multi_scale_features[3]); fp32 like the
        reference (`@autocast(enabled=False)` + `.float()`, :318,324,348)."""
        with torch.autocast(device_type=next(iter(features.values())).device.type, enabled=False):
            srcs, pos = [], []
            for idx, f in enumerate(self.transformer_in_features[::-1]):
                x = _as_fp32_input(features[f], self.input_proj[idx][0])
                srcs.append(_conv_gn(self.input_proj[idx][0], self.input_proj[idx][1], x))
                pos.append(self.pe_layer(x))
            y, spatial_shapes, level_start_index, shapes_host = self.transformer(srcs, pos)
            bs = y.shape[0]
            sizes = [h * w for h, w in shapes_host]
            out = [z.transpose(1, 2).reshape(bs, -1, h, w) for z, (h, w) in zip(torch.split(y, sizes, dim=1), shapes_host)]
            for idx, f in enumerate(self.in_features[: self.num_fpn_levels][::-1]):
                x = _as_fp32_input(features[f], self.lateral_convs[idx])
                lat, outc = self.lateral_convs[idx], self.output_convs[idx]
                cur = _conv_gn(lat, lat.norm, x, relu=lat.activation is not None)
                if upsample_add_supported(out[-1], cur):
                    y = upsample_add(out[-1], cur)                # one pass; gather-form backward (functions/rowwise.py)
                else:
                    y = cur + F.interpolate(out[-1], size=cur.shape[-2:], mode="bilinear", align_corners=False)
                out.append(_conv_gn(outc, outc.norm, y, relu=outc.activation is not None))
            multi_scale = out[: self.maskformer_num_feature_levels]
            mf = self.mask_features
            if CONV_X3 and mf.norm is None and mf.activation is None and conv_x3.conv1x1_supported(out[-1], mf):
                return conv_x3.conv1x1(out[-1], mf.weight, mf.bias), out[0], multi_scale
            return mf(out[-1]), out[0], multi_scale
