"""Swin Transformer backbone (reference modeling/backbone/swin.py:24-774) with
the reference's module tree / state_dict keys and the
``D2SwinTransformer(cfg, input_shape)`` registry entry.

Written window-major: tokens are brought into (window, token, channel) order
once per block by a single gather index (cyclic shift + padding + partition
folded into one permutation) instead of pad -> roll -> view/permute/contiguous
copies, and sent back by the inverse scatter; the shifted-window mask and the
relative-position bias are added as one pre-combined additive term."""
import numpy as np
import torch
import torch.nn.functional as F
import torch.utils.checkpoint as checkpoint
from torch import nn

from ...functions import fp8
from ...functions import window_attention as wattn
from ...functions import swin_rows as _swin_rows
from ...functions.rowwise import add_layer_norm, supports_width

from ...compat import BACKBONE_REGISTRY, ShapeSpec


# BASELINE config 5 ("fp8 MFMA GEMMs"): D2SwinTransformer sets these from MODEL.SWIN.FP8_GEMM / FP8_MIN_K; the qkv / proj / MLP Linears
# of the stages with C >= FP8_MIN_K then run as MX-fp8 GEMMs on own kernels (include/pd_mx8.h) while autocast is on: inside the fused
# stage (swin_core.py), or one by one through functions/fp8.py on the module-by-module path
FP8 = {"enabled": False, "min_k": 384}
MERGE_FUSED = __import__("os").environ.get("PD_SWIN_MERGE_FUSED", "1") != "0"   # PatchMerging's gather + LayerNorm + bf16 cast as pd_swin_merge_ln_* (0: permuted copy, row LayerNorm, cast)
OWN_MERGE_NORM = __import__("os").environ.get("PD_SWIN_OWN_MERGE_NORM", "1") != "0"   # patch embedding / merging LayerNorms likewise (0: ATen)
OWN_OUT_NORM = __import__("os").environ.get("PD_SWIN_OWN_OUT_NORM", "1") != "0"   # the stages' output LayerNorms on pd_layernorm_rows_f32_* (0: ATen)
FUSED_STAGE = True          # modeling/backbone/swin_core.py where it applies (tests switch it off to compare the two paths)


def _linear(mod, x):
    if FP8["enabled"] and torch.is_autocast_enabled() and fp8.supported(x, mod.weight, FP8["min_k"]):
        return fp8.linear(x, mod.weight, mod.bias)
    return mod(x)


def _trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, std=std)


class DropPath(nn.Module):
    """per-sample stochastic depth (timm DropPath)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * mask / keep


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(_linear(self.fc2, self.drop(self.act(_linear(self.fc1, x)))))


def window_partition(x, window_size):
    B, H, W, C = x.shape
    x = x.view(B, H // window_size, window_size, W // window_size, window_size, C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, window_size, window_size, C)


def window_reverse(windows, window_size, H, W):
    B = int(windows.shape[0] / (H * W / window_size / window_size))
    x = windows.view(B, H // window_size, W // window_size, window_size, window_size, -1)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


class _TableLookup(torch.autograd.Function):
    """table[index] for the relative-position bias; backward = index_add_ (atomics) instead of torch's sort-based
    indexing backward, which costs ~80 us per block for a 529-row table"""

    @staticmethod
    def forward(ctx, table, index):
        ctx.save_for_backward(index)
        ctx.rows = table.shape[0]
        return table.index_select(0, index)

    @staticmethod
    def backward(ctx, g):
        (index,) = ctx.saved_tensors
        return torch.zeros((ctx.rows, g.shape[1]), dtype=g.dtype, device=g.device).index_add_(0, index, g), None


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        wh, ww = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * wh - 1) * (2 * ww - 1), num_heads))
        ch, cw = torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing="ij")
        coords = torch.stack([ch.reshape(-1), cw.reshape(-1)])                          # [2, N]
        rel = coords[:, :, None] - coords[:, None, :]                                   # [2, N, N]
        index = (rel[0] + wh - 1) * (2 * ww - 1) + (rel[1] + ww - 1)
        self.register_buffer("relative_position_index", index)
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        _trunc_normal_(self.relative_position_bias_table, std=0.02)

    def bias(self):
        n = self.window_size[0] * self.window_size[1]
        return _TableLookup.apply(self.relative_position_bias_table, self.relative_position_index.view(-1)).view(n, n, -1).permute(2, 0, 1)

    def forward(self, x, mask=None, regions=None, n_windows=1):
        """x [nW*B, N, C]; mask [nW, N, N] additive (0 / -100) or None; regions = the same mask as per-token region
        labels (functions/window_attention.shifted_window_regions) for the fused kernel."""
        B_, N, C = x.shape
        h = self.num_heads
        qkv = _linear(self.qkv, x)
        p = self.attn_drop.p if self.training else 0.0
        if wattn.supported(qkv, self.window_size, h, p) and (mask is None) == (regions is None):
            out = wattn.window_attention(qkv, self.relative_position_bias_table, regions, self.scale, n_windows)
            return self.proj_drop(_linear(self.proj, out))
        qkv = qkv.reshape(B_, N, 3, h, C // h).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        add = self.bias().unsqueeze(0)                                                  # [1,h,N,N]
        if mask is not None:
            nW = mask.shape[0]
            add = (add + mask.unsqueeze(1)).unsqueeze(0).expand(B_ // nW, -1, -1, -1, -1).reshape(B_, h, N, N)
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=add.to(q.dtype), dropout_p=p, scale=self.scale)
        return self.proj_drop(_linear(self.proj, out.transpose(1, 2).reshape(B_, N, C)))


def _layer_norm(norm, x):
    """nn.LayerNorm through the one-wavefront-per-row HIP kernels (functions/rowwise.py) where they apply (GPU, width a
    multiple of 256 up to 1024: Swin-B stages 2-4, Swin-L stage 3): one backward kernel instead of torch's three"""
    if (x.is_cuda and isinstance(norm, nn.LayerNorm) and norm.elementwise_affine and norm.weight.dtype == torch.float32
            and x.dtype in (torch.float32, torch.bfloat16) and supports_width(x.shape[-1])):
        return add_layer_norm(x, None, norm)[0]
    return norm(x)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size=7, shift_size=0, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop=0.0, attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim, self.num_heads, self.window_size, self.shift_size, self.mlp_ratio = \
            dim, num_heads, window_size, shift_size, mlp_ratio
        assert 0 <= self.shift_size < self.window_size, "shift_size must in 0-window_size"
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, (window_size, window_size), num_heads, qkv_bias, qk_scale, attn_drop, drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.H = self.W = None

    def forward(self, x, mask_matrix, gather=None):
        """x [B, H*W, C].  ``gather`` = window_gather_index(...) (index, inverse, #windows, has-padding flag)."""
        B, L, C = x.shape
        H, W = self.H, self.W
        assert L == H * W, "input feature has wrong size"
        ws = self.window_size
        shortcut = x
        x = _layer_norm(self.norm1, x)
        if gather is None:
            gather = window_gather_index(H, W, ws, self.shift_size, x.device)
        idx, inv, n_win, any_pad = gather
        xw = _WindowGather.apply(x, idx, inv, any_pad).reshape(B * n_win, ws * ws, C)   # padded slots read a zero row
        shifted = self.shift_size > 0
        regions = wattn.shifted_window_regions(H, W, self.shift_size, x.device) if shifted and ws == wattn.WINDOW else None
        aw = self.attn(xw, mask=mask_matrix if shifted else None, regions=regions, n_windows=n_win)
        x = _WindowScatter.apply(aw.reshape(B, n_win * ws * ws, C), idx, inv, any_pad)
        x = shortcut + self.drop_path(x)
        return x + self.drop_path(self.mlp(_layer_norm(self.norm2, x)))


class _WindowGather(torch.autograd.Function):
    """tokens [B, L, C] -> window-major slots [B, S, C] (S >= L; padded slots read zeros).  The slot map is a permutation
    of the tokens plus padding, so the backward is the INVERSE GATHER g[:, inv] - not the sort-based scatter-add torch
    runs for advanced indexing (indexing_backward_kernel: 20 % of a Swin-B training step)."""

    @staticmethod
    def forward(ctx, x, idx, inv, any_pad):
        ctx.save_for_backward(inv)
        xp = torch.cat([x, x.new_zeros(x.shape[0], 1, x.shape[2])], dim=1) if any_pad else x
        return xp.index_select(1, idx)

    @staticmethod
    def backward(ctx, g):
        (inv,) = ctx.saved_tensors
        return g.index_select(1, inv), None, None, None


class _WindowScatter(torch.autograd.Function):
    """window-major slots [B, S, C] -> tokens [B, L, C] (the inverse of _WindowGather); backward = gather with the slot map,
    zeros for the padded slots."""

    @staticmethod
    def forward(ctx, a, idx, inv, any_pad):
        ctx.save_for_backward(idx)
        ctx.any_pad = any_pad
        return a.index_select(1, inv)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        gp = torch.cat([g, g.new_zeros(g.shape[0], 1, g.shape[2])], dim=1) if ctx.any_pad else g
        return gp.index_select(1, idx), None, None, None


_GATHER_CACHE = {}


def window_gather_index(H, W, ws, shift, device):
    """window-major order of the (padded, cyclically shifted) token grid: replaces F.pad + torch.roll +
    window_partition (reference :254-270) and their inverses (:276-289) by one gather each way."""
    key = (H, W, ws, shift, str(device))
    if key not in _GATHER_CACHE:
        Hp, Wp = int(np.ceil(H / ws)) * ws, int(np.ceil(W / ws)) * ws
        grid = torch.full((Hp, Wp), H * W, dtype=torch.long)                             # H*W = the zero row
        grid[:H, :W] = torch.arange(H * W).view(H, W)
        if shift > 0:
            grid = torch.roll(grid, shifts=(-shift, -shift), dims=(0, 1))
        win = grid.view(Hp // ws, ws, Wp // ws, ws).permute(0, 2, 1, 3).reshape(-1)      # window-major
        any_pad = bool((win == H * W).any())
        inv = torch.empty(H * W, dtype=torch.long)
        pos = torch.arange(win.numel())
        keep = win < H * W
        inv[win[keep]] = pos[keep]
        _GATHER_CACHE[key] = (win.to(device), inv.to(device), (Hp // ws) * (Wp // ws), any_pad)
    return _GATHER_CACHE[key]


def _rows_norm(norm, x):
    """fp32 LayerNorm of token rows on pd_layernorm_rows_f32_* where it applies (GPU, fp32 rows, width <= 3072), like the stages' output norms
    (ATen: one forward and three backward kernels per norm, 0.6 ms per Swin-B step for the four norms of patch embedding / merging)"""
    if OWN_MERGE_NORM and isinstance(norm, nn.LayerNorm) and x.is_cuda:
        if x.dtype == torch.bfloat16 and torch.is_autocast_enabled():          # (autocast runs layer_norm in fp32: the same cast ATen would make)
            x = x.float()
        if _swin_rows.rows_layer_norm_supported(x, norm):
            return _swin_rows.rows_layer_norm(x, norm)
    return norm(x)


class PatchMerging(nn.Module):
    def __init__(self, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x, H, W):
        B, L, C = x.shape
        assert L == H * W, "input feature has wrong size"
        from ...functions import igemm
        from . import swin_core
        if (OWN_MERGE_NORM and MERGE_FUSED and isinstance(self.norm, nn.LayerNorm) and self.reduction.bias is None and swin_core.OWN_GEMM
                and igemm.own_linear_supported(x, self.reduction.weight) and _swin_rows.merge_layer_norm_supported(x, H, W, self.norm)):
            # (bf16 autocast with 16-bit weights, the training configuration:) gather + LayerNorm in one kernel, bf16 out — no permuted copy, no casts
            return igemm.OwnLinear.apply(_swin_rows.merge_layer_norm(x, H, W, self.norm), self.reduction.weight, None)
        x = x.view(B, H, W, C)
        if (H % 2 == 1) or (W % 2 == 1):
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        Hp, Wp = x.shape[1], x.shape[2]
        # the reference's cat([x[0::2, 0::2], x[1::2, 0::2], x[0::2, 1::2], x[1::2, 1::2]], -1) as ONE permuted copy: channel block
        # 2 * (column parity) + (row parity).  Same values; the backward is one permuted copy too instead of four zero-fills, four
        # strided copies and three gradient sums per merge (reference swin.py:325-337)
        x = x.view(B, Hp // 2, 2, Wp // 2, 2, C).permute(0, 1, 3, 4, 2, 5).reshape(B, -1, 4 * C)
        x = _rows_norm(self.norm, x)
        from ...functions import igemm
        from . import swin_core
        if swin_core.OWN_GEMM and self.reduction.bias is None and igemm.own_linear_supported(x, self.reduction.weight):
            return igemm.OwnLinear.apply(x, self.reduction.weight, None)           # pd_igemm_bf16 forward / input gradient, own weight gradient
        return self.reduction(x)


_MASK_CACHE = {}


def shifted_window_mask(H, W, ws, shift, device):
    """additive SW-MSA mask [nW, N, N] (0 / -100), reference :417-444."""
    key = (H, W, ws, shift, str(device))
    if key not in _MASK_CACHE:
        Hp, Wp = int(np.ceil(H / ws)) * ws, int(np.ceil(W / ws)) * ws
        img = torch.zeros((1, Hp, Wp, 1))
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                img[:, hs, wsl, :] = cnt
                cnt += 1
        mw = window_partition(img, ws).view(-1, ws * ws)
        am = mw.unsqueeze(1) - mw.unsqueeze(2)
        _MASK_CACHE[key] = am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0).to(device)
    return _MASK_CACHE[key]


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size=7, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop=0.0,
                 attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False):
        super().__init__()
        self.window_size, self.shift_size, self.depth, self.use_checkpoint = window_size, window_size // 2, depth, use_checkpoint
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, num_heads, window_size, 0 if (i % 2 == 0) else window_size // 2, mlp_ratio,
                                 qkv_bias, qk_scale, drop, attn_drop,
                                 drop_path[i] if isinstance(drop_path, list) else drop_path, norm_layer=norm_layer)
            for i in range(depth)])
        self.downsample = downsample(dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def forward(self, x, H, W, out_norm=None):
        """out_norm: the backbone's norm{i} of this stage's output, or None; the fused stage applies it inside its closing kernel and hands the result back
        as `self.normed` (None when it did not)"""
        from . import swin_core
        fused = FUSED_STAGE and swin_core.supported(self, x)
        if fused and FP8["enabled"] and swin_core.wants_mx8(x.shape[-1]) and not swin_core.mx8_ready(self):
            fused = False                                                 # fp8 asked for, fp32 module weights: the Linears go one by one (functions/fp8.py)
        self.normed = None
        if fused:
            x, self.normed = swin_core.run_stage(self, x, H, W, out_norm)  # the whole stage as one autograd node
        else:
            attn_mask = shifted_window_mask(H, W, self.window_size, self.shift_size, x.device)
            for blk in self.blocks:
                blk.H, blk.W = H, W
                g = window_gather_index(H, W, self.window_size, blk.shift_size, x.device)
                x = checkpoint.checkpoint(blk, x, attn_mask, g, use_reentrant=False) if self.use_checkpoint else blk(x, attn_mask, g)
        if self.downsample is not None:
            return x, H, W, self.downsample(x, H, W), (H + 1) // 2, (W + 1) // 2
        return x, H, W, x, H, W


class PatchEmbed(nn.Module):
    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        self.patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x):
        _, _, H, W = x.size()
        ph, pw = self.patch_size
        if W % pw != 0:
            x = F.pad(x, (0, pw - W % pw))
        if H % ph != 0:
            x = F.pad(x, (0, 0, 0, ph - H % ph))
        x = self.proj(x)
        if self.norm is not None:
            Wh, Ww = x.size(2), x.size(3)
            x = _rows_norm(self.norm, x.flatten(2).transpose(1, 2)).transpose(1, 2).view(-1, self.embed_dim, Wh, Ww)
        return x


class SwinTransformer(nn.Module):
    def __init__(self, pretrain_img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=[2, 2, 6, 2],
                 num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop_rate=0.0,
                 attn_drop_rate=0.0, drop_path_rate=0.2, norm_layer=nn.LayerNorm, ape=False, patch_norm=True,
                 out_indices=(0, 1, 2, 3), frozen_stages=-1, use_checkpoint=False):
        super().__init__()
        self.pretrain_img_size, self.num_layers, self.embed_dim = pretrain_img_size, len(depths), embed_dim
        self.ape, self.patch_norm, self.out_indices, self.frozen_stages = ape, patch_norm, out_indices, frozen_stages
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim, norm_layer if patch_norm else None)
        if self.ape:
            ps = (pretrain_img_size, pretrain_img_size) if isinstance(pretrain_img_size, int) else pretrain_img_size
            pp = (patch_size, patch_size) if isinstance(patch_size, int) else patch_size
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, embed_dim, ps[0] // pp[0], ps[1] // pp[1]))
            _trunc_normal_(self.absolute_pos_embed, std=0.02)
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(
                dim=int(embed_dim * 2 ** i), depth=depths[i], num_heads=num_heads[i], window_size=window_size,
                mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate,
                drop_path=dpr[sum(depths[:i]): sum(depths[: i + 1])], norm_layer=norm_layer,
                downsample=PatchMerging if (i < self.num_layers - 1) else None, use_checkpoint=use_checkpoint))
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        for i in out_indices:
            self.add_module(f"norm{i}", norm_layer(self.num_features[i]))
        self._freeze_stages()

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.patch_embed.eval()
            for p in self.patch_embed.parameters():
                p.requires_grad = False
        if self.frozen_stages >= 1 and self.ape:
            self.absolute_pos_embed.requires_grad = False
        if self.frozen_stages >= 2:
            self.pos_drop.eval()
            for i in range(0, self.frozen_stages - 1):
                self.layers[i].eval()
                for p in self.layers[i].parameters():
                    p.requires_grad = False

    def forward(self, x):
        x = self.patch_embed(x)
        Wh, Ww = x.size(2), x.size(3)
        if self.ape:
            x = x + F.interpolate(self.absolute_pos_embed, size=(Wh, Ww), mode="bicubic")
        x = self.pos_drop(x.flatten(2).transpose(1, 2))
        if FUSED_STAGE and x.is_cuda and self.training:
            from . import swin_core
            swin_core.draw_drop_path(self, x.shape[0], x.device)
        outs = {}
        for i, layer in enumerate(self.layers):
            ln = getattr(self, f"norm{i}") if i in self.out_indices else None
            x_out, H, W, x, Wh, Ww = layer(x, Wh, Ww, ln if OWN_OUT_NORM else None)
            if i in self.out_indices:
                normed, layer.normed = layer.normed, None
                if normed is not None:
                    x_out = normed                                           # the fused stage's closing kernel already normalised its output
                elif OWN_OUT_NORM and _swin_rows.rows_layer_norm_supported(x_out, ln):
                    x_out = _swin_rows.rows_layer_norm(x_out, ln)            # fp32 rows in, fp32 rows out: pd_layernorm_rows_f32_* (ATen: 1 - 2 ms per step)
                else:
                    x_out = ln(x_out)
                o = x_out.view(-1, H, W, self.num_features[i]).permute(0, 3, 1, 2)
                # on the GPU the tokens ARE the channels-last storage of the [B, C, H, W] map the pixel decoder wants: no NCHW copy
                # (reference :634 .contiguous(); the copy, the convolution's copy back and their two backward copies were 0.5 ms at config 3)
                outs[f"res{i + 2}"] = o if o.is_cuda else o.contiguous()
        return outs

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        return self                      # (the reference forgets the return, swin.py:684-687)


@BACKBONE_REGISTRY.register()
class D2SwinTransformer(SwinTransformer):
    def __init__(self, cfg, input_shape):
        s = cfg.MODEL.SWIN
        super().__init__(s.PRETRAIN_IMG_SIZE, s.PATCH_SIZE, 3, s.EMBED_DIM, list(s.DEPTHS), list(s.NUM_HEADS),
                         s.WINDOW_SIZE, s.MLP_RATIO, s.QKV_BIAS, s.QK_SCALE, s.DROP_RATE, s.ATTN_DROP_RATE,
                         s.DROP_PATH_RATE, nn.LayerNorm, s.APE, s.PATCH_NORM, use_checkpoint=s.USE_CHECKPOINT)
        FP8["enabled"], FP8["min_k"] = bool(s.get("FP8_GEMM", False)), int(s.get("FP8_MIN_K", 384))
        self._out_features = s.OUT_FEATURES
        self._out_feature_strides = {"res2": 4, "res3": 8, "res4": 16, "res5": 32}
        self._out_feature_channels = {f"res{i + 2}": self.num_features[i] for i in range(4)}

    def forward(self, x):
        assert x.dim() == 4, f"SwinTransformer takes an input of shape (N, C, H, W). Got {x.shape} instead!"
        y = super().forward(x)
        return {k: v for k, v in y.items() if k in self._out_features}

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n])
                for n in self._out_features}

    @property
    def size_divisibility(self):
        return 32
