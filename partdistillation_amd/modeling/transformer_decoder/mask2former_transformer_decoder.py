"""Masked-attention transformer decoder (reference
transformer_decoder/mask2former_transformer_decoder.py:21-472): Q learnable
queries, L x (masked cross-attention -> self-attention -> FFN), L+1 prediction
heads.  Same class names, ctor / from_config arguments, output dict and
state_dict keys.

Execution differs from the reference where the arithmetic allows it:
  * attention is written out (in_proj / out_proj weights of the same names) so
    the boolean mask stays [B,1,Q,HW] instead of being repeated over heads and
    the [B*h,Q,HW] probabilities are never head-averaged (the reference calls
    nn.MultiheadAttention with need_weights=True, :107-110);
  * the next layer's attention mask is ``mask_embed . resize(mask_features)``:
    bilinear resize is linear, so it commutes with the einsum (:449-456) and
    mask_features is resized three times per step instead of resizing ten
    [B,Q,H/4,W/4] prediction stacks; ``sigmoid(x) < 0.5`` is evaluated as x < 0;
  * positional tables are cached per shape.
"""
import logging
from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from ...compat import TRANSFORMER_DECODER_REGISTRY, configurable
from ...compat.layers import Conv2d, c2_xavier_fill
from ...functions import mlp_own
from ...functions.attention import masked_attention_d32
from ...functions.decoder_core import DecoderCore, DecoderSpec
from ...functions.rowwise import resize_bilinear_rows, resize_bilinear_rows_supported, supports_width
from .position_encoding import PositionEmbeddingSine


class _MHAParams(nn.Module):
    """Parameter container with nn.MultiheadAttention's names: in_proj_weight [3C,C], in_proj_bias, out_proj."""

    def __init__(self, d_model, nhead):
        super().__init__()
        self.embed_dim, self.num_heads, self.head_dim = d_model, nhead, d_model // nhead
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = nn.Linear(d_model, d_model)
        self.split_key_min_keys = 2048
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def forward(self, query, key, value, blocked: Optional[Tensor] = None):
        """seq-first [L,B,C]; ``blocked`` bool [B,Lq,Lk], True = may not attend."""
        Lq, B, C = query.shape
        Lk, h, d = key.shape[0], self.num_heads, self.head_dim
        w, b = self.in_proj_weight, self.in_proj_bias
        if key is query:                                       # q and k share the input: one GEMM
            qk = F.linear(query, w[: 2 * C], b[: 2 * C])
            q, k = qk[..., :C], qk[..., C:]
        else:
            q = F.linear(query, w[:C], b[:C])
            k = F.linear(key, w[C: 2 * C], b[C: 2 * C])
        v = F.linear(value, w[2 * C:], b[2 * C:])
        if d == 32 and q.is_cuda and Lk >= self.split_key_min_keys and q.dtype in (torch.bfloat16, torch.float32):
            # few queries x many keys: hand-written split-key attention (functions/attention.py); measured 6x (Lk=16384)
            # and 4x (Lk=4096) faster forward than the library flash kernel, which tiles over the 100 queries only
            o = masked_attention_d32(q, k, v, blocked, h, d ** -0.5)
            return self.out_proj(o)
        q = q.reshape(Lq, B, h, d).permute(1, 2, 0, 3)          # [B,h,Lq,d]
        k = k.reshape(Lk, B, h, d).permute(1, 2, 0, 3)
        v = v.reshape(Lk, B, h, d).permute(1, 2, 0, 3)
        mask = None
        if blocked is not None:
            mask = torch.zeros((B, 1, Lq, Lk), dtype=q.dtype, device=q.device).masked_fill_(blocked.unsqueeze(1), float("-inf"))
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
        o = o.permute(2, 0, 1, 3).reshape(Lq, B, C)
        return self.out_proj(o)


def _with_pos(t, pos):
    return t if pos is None else t + pos


class SelfAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        self.self_attn = _MHAParams(d_model, nhead)
        self.norm = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        self.normalize_before = normalize_before
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, tgt, tgt_mask=None, tgt_key_padding_mask=None, query_pos=None):
        x = self.norm(tgt) if self.normalize_before else tgt
        qk = _with_pos(x, query_pos)
        y = tgt + self.dropout(self.self_attn(qk, qk, x, tgt_mask))
        return y if self.normalize_before else self.norm(y)


class CrossAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        self.multihead_attn = _MHAParams(d_model, nhead)
        self.norm = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        self.normalize_before = normalize_before
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, tgt, memory, memory_mask=None, memory_key_padding_mask=None, pos=None, query_pos=None):
        x = self.norm(tgt) if self.normalize_before else tgt
        y = tgt + self.dropout(self.multihead_attn(_with_pos(x, query_pos), _with_pos(memory, pos), memory, memory_mask))
        return y if self.normalize_before else self.norm(y)


class FFNLayer(nn.Module):
    def __init__(self, d_model, dim_feedforward=2048, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm = nn.LayerNorm(d_model)
        self.normalize_before = normalize_before
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, tgt):
        x = self.norm(tgt) if self.normalize_before else tgt
        y = tgt + self.dropout(self.linear2(self.dropout(F.relu(self.linear1(x)))))
        return y if self.normalize_before else self.norm(y)


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        if mlp_own.supported(x, self.layers):                   # bf16 autocast on the GPU: one autograd node on own kernels
            return mlp_own.mlp(x, self.layers)
        for i, layer in enumerate(self.layers):
            x = layer(x)
            if i < self.num_layers - 1:
                x = F.relu(x)
        return x


def materialize_masks(out):
    """fills pred_masks / aux_outputs[*].pred_masks / all_masks of a decoder output dict from mask_embeds and
    mask_features: einsum("bqc,bchw->bqhw") of the reference (:449) for all L+1 heads as ONE batched GEMM
    [B, (L+1)*Q, C] x [B, C, HW] - ten [200 x 256 x 65536] products are too skinny for the matrix cores one at a time."""
    emb, mask_features = out["mask_embeds"], out["mask_features"]
    B_, Hh, Qn, Cc = emb.shape
    mf = mask_features.to(emb.dtype).flatten(2)                               # [B, C, HW]
    all_masks = torch.bmm(emb.reshape(B_, Hh * Qn, Cc), mf).view(B_, Hh, Qn, *mask_features.shape[-2:])
    masks = [all_masks[:, i] for i in range(Hh)]
    classes = list(out["all_logits"].unbind(0)) if out.get("all_logits") is not None else None
    out["pred_masks"], out["all_masks"] = masks[-1], all_masks                # all_masks: [B, heads (decoder order), Q, H, W]
    if classes is not None:
        out["aux_outputs"] = [{"pred_logits": a, "pred_masks": b} for a, b in zip(classes[:-1], masks[:-1])]
    else:
        out["aux_outputs"] = [{"pred_masks": b} for b in masks[:-1]]
    return out


@TRANSFORMER_DECODER_REGISTRY.register()
class MultiScaleMaskedTransformerDecoder(nn.Module):
    _version = 2

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        version = local_metadata.get("version", None)
        if version is None or version < 2:                      # legacy key: static_query -> query_feat
            renamed = False
            for k in list(state_dict.keys()):
                if "static_query" in k:
                    state_dict[k.replace("static_query", "query_feat")] = state_dict.pop(k)
                    renamed = True
            if renamed:
                logging.getLogger(__name__).warning(
                    f"Weight format of {self.__class__.__name__} have changed! Applying automatic conversion now ...")
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    @configurable
    def __init__(self, in_channels, mask_classification=True, *, num_classes: int, hidden_dim: int, num_queries: int,
                 nheads: int, dim_feedforward: int, dec_layers: int, pre_norm: bool, mask_dim: int,
                 enforce_input_project: bool, query_feature_normalize: bool):
        super().__init__()
        assert mask_classification, "Only support mask classification model"
        self.mask_classification = mask_classification
        self.pe_layer = PositionEmbeddingSine(hidden_dim // 2, normalize=True)
        self.num_heads, self.num_layers, self.hidden_dim = nheads, dec_layers, hidden_dim
        self.transformer_self_attention_layers = nn.ModuleList()
        self.transformer_cross_attention_layers = nn.ModuleList()
        self.transformer_ffn_layers = nn.ModuleList()
        for _ in range(dec_layers):
            self.transformer_self_attention_layers.append(SelfAttentionLayer(hidden_dim, nheads, 0.0, normalize_before=pre_norm))
            self.transformer_cross_attention_layers.append(CrossAttentionLayer(hidden_dim, nheads, 0.0, normalize_before=pre_norm))
            self.transformer_ffn_layers.append(FFNLayer(hidden_dim, dim_feedforward, 0.0, normalize_before=pre_norm))
        self.decoder_norm = nn.LayerNorm(hidden_dim)
        self.num_queries = num_queries
        self.query_feat = nn.Embedding(num_queries, hidden_dim)
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.num_feature_levels = 3
        self.level_embed = nn.Embedding(self.num_feature_levels, hidden_dim)
        self.input_proj = nn.ModuleList()
        for _ in range(self.num_feature_levels):
            if in_channels != hidden_dim or enforce_input_project:
                self.input_proj.append(Conv2d(in_channels, hidden_dim, kernel_size=1))
                c2_xavier_fill(self.input_proj[-1])
            else:
                self.input_proj.append(nn.Sequential())
        if self.mask_classification:
            self.class_embed = nn.Linear(hidden_dim, num_classes + 1)
        self.mask_embed = MLP(hidden_dim, hidden_dim, mask_dim, 3)
        self.query_feature_normalize = query_feature_normalize
        self.dense_masks = True            # False: training never materialises [B,Q,H,W] masks (sparse criterion)
        self.fused_core = True             # GPU: run the layer loop as one hand-written autograd node (functions/decoder_core.py)
        self.pre_norm = pre_norm
        self._pos_rows = {}

    @classmethod
    def from_config(cls, cfg, in_channels, mask_classification):
        mf = cfg.MODEL.MASK_FORMER
        assert mf.DEC_LAYERS >= 1
        return dict(in_channels=in_channels, mask_classification=mask_classification,
                    num_classes=cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES, hidden_dim=mf.HIDDEN_DIM,
                    num_queries=mf.NUM_OBJECT_QUERIES, nheads=mf.NHEADS, dim_feedforward=mf.DIM_FEEDFORWARD,
                    dec_layers=mf.DEC_LAYERS - 1,     # one head on the learnable queries + one per layer (:357-362)
                    pre_norm=mf.PRE_NORM, enforce_input_project=mf.ENFORCE_INPUT_PROJ,
                    query_feature_normalize=mf.QUERY_FEATURE_NORMALIZE, mask_dim=cfg.MODEL.SEM_SEG_HEAD.MASK_DIM)

    # ------------------------------------------------------------------ pieces
    def _memory(self, x):
        """per level: src [HW,B,C] (+level embedding), pos [HW,1,C], size."""
        src, pos, sizes = [], [], []
        for i in range(self.num_feature_levels):
            h, w = x[i].shape[-2:]
            sizes.append((h, w))
            pos.append(self.pe_layer.table(h, w, x[i].device).flatten(1).t().unsqueeze(1))
            s = self.input_proj[i](x[i]).flatten(2) + self.level_embed.weight[i][None, :, None]
            src.append(s.permute(2, 0, 1))
        return src, pos, sizes

    def _class_logits(self, decoder_output, extra):
        return self.class_embed(decoder_output)

    def forward_prediction_heads(self, output, mask_features, attn_mask_target_size, pooled=None, extra=None):
        decoder_output = self.decoder_norm(output).transpose(0, 1)              # [B,Q,C]
        outputs_class = self._class_logits(decoder_output, extra)
        mask_embed = self.mask_embed(decoder_output)
        if self.query_feature_normalize:
            mask_embed = F.normalize(mask_embed, p=2, dim=-1)
        outputs_mask = None            # the full-resolution masks of ALL heads are produced by one GEMM in forward()
        if pooled is None:
            pooled = F.interpolate(mask_features, size=attn_mask_target_size, mode="bilinear", align_corners=False)
        with torch.no_grad():
            logits = torch.einsum("bqc,bcn->bqn", mask_embed.detach().float(), pooled.detach().flatten(2).float())
            attn_mask = logits < 0                                              # sigmoid(x) < 0.5; [B,Q,HW]
        return outputs_class, outputs_mask, attn_mask, decoder_output, mask_embed

    # ------------------------------------------------------------------ fused layer loop (GPU)
    def _core_dtype(self, x):
        """GEMM dtype of the fused core, or None when the generic module path has to run."""
        w = self.transformer_ffn_layers[0].linear1.weight if self.num_layers else None
        if (not self.fused_core or w is None or not x[0].is_cuda or self.pre_norm or not supports_width(self.hidden_dim)
                or self.hidden_dim // self.num_heads != 32
                or any(not (isinstance(p, nn.Sequential) and len(p) == 0) for p in self.input_proj)
                or self.transformer_ffn_layers[0].linear1.out_features % 128):
            return None
        cdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else torch.float32
        if cdt not in (torch.float32, torch.bfloat16) or w.dtype != cdt:
            return None                    # e.g. autocast over fp32 (unshadowed) weights: the module path casts per use
        return cdt

    def _pos_table_rows(self, h, w, device):
        key = (h, w, str(device))
        if key not in self._pos_rows:
            self._pos_rows[key] = self.pe_layer.table(h, w, device).flatten(1).t().contiguous().float()      # [HW, C]
        return self._pos_rows[key]

    def _core_params(self):
        p = [self.query_feat.weight, self.query_embed.weight, self.level_embed.weight, self.decoder_norm.weight, self.decoder_norm.bias]
        for l in self.mask_embed.layers:
            p += [l.weight, l.bias]
        for i in range(self.num_layers):
            ca, sa, ff = self.transformer_cross_attention_layers[i], self.transformer_self_attention_layers[i], self.transformer_ffn_layers[i]
            p += [ca.multihead_attn.in_proj_weight, ca.multihead_attn.in_proj_bias, ca.multihead_attn.out_proj.weight,
                  ca.multihead_attn.out_proj.bias, ca.norm.weight, ca.norm.bias,
                  sa.self_attn.in_proj_weight, sa.self_attn.in_proj_bias, sa.self_attn.out_proj.weight, sa.self_attn.out_proj.bias,
                  sa.norm.weight, sa.norm.bias,
                  ff.linear1.weight, ff.linear1.bias, ff.linear2.weight, ff.linear2.bias, ff.norm.weight, ff.norm.bias]
        return p

    def _class_logits_stacked(self, d, extra):
        """d [L+1, B, Q, C] -> [L+1, B, Q, K+1]"""
        return self._class_logits(d, extra)

    def _forward_fused(self, x, mask_features, extra, cdt):
        bs, Q, C, L = x[0].shape[0], self.num_queries, self.hidden_dim, self.num_layers
        sizes = [tuple(t.shape[-2:]) for t in x]
        with torch.no_grad():                                                   # the three mask resolutions, once
            to_cdt = cdt == torch.bfloat16 and torch.is_autocast_enabled()
            if resize_bilinear_rows_supported(mask_features, sizes):
                # all levels by one launch, written as [B, HW, C] rows in the GEMM dtype (three ATen resizes + three casts before)
                rows = resize_bilinear_rows(mask_features.detach(), sizes, cdt if to_cdt else torch.float32)
                pooled = [r.transpose(1, 2) for r in rows]                      # [B, C, HW] views, as flatten(2) of the channels-last maps were
            else:
                pooled = [F.interpolate(mask_features, size=s, mode="bilinear", align_corners=False).flatten(2).float() for s in sizes]
                if to_cdt:
                    # the heads' mask logits are autocast bmm's: the B operand in the GEMM dtype once per level, not once per head
                    # (the same rounding, nine cast launches less)
                    pooled = [p.to(cdt) for p in pooled]
        spec = DecoderSpec(bs, Q, C, self.num_heads, L, sizes, [self._pos_table_rows(h, w, x[0].device) for h, w in sizes],
                           pooled, self.decoder_norm.eps, cdt, self.num_feature_levels)
        dec_outs, final_tgt = DecoderCore.apply(spec, *x, *self._core_params())          # [L+1, Q*B, C] fp32, [Q*B, C]
        d = dec_outs.view(L + 1, Q, bs, C).transpose(1, 2)                               # [L+1, B, Q, C]
        if (self.mask_classification and type(self)._class_logits is MultiScaleMaskedTransformerDecoder._class_logits
                and mlp_own.heads_supported(d, self.class_embed, self.mask_embed.layers)):
            all_logits, emb = mlp_own.heads(d, self.class_embed, self.mask_embed.layers)  # class head + MLP: one node, one bf16 copy of d
        else:
            all_logits = self._class_logits_stacked(d, extra) if self.mask_classification else None
            emb = self.mask_embed(d)                                                     # one MLP pass for the L+1 heads
        if self.query_feature_normalize:
            emb = F.normalize(emb, p=2, dim=-1)
        emb = emb.transpose(0, 1)                                                        # [B, L+1, Q, C]
        return self._assemble(all_logits, emb, d[-1], mask_features, final_tgt.view(Q, bs, C))

    def _assemble(self, all_logits, emb, dec_out, mask_features, output):
        classes = list(all_logits.unbind(0)) if self.mask_classification else None
        out = {"pred_logits": classes[-1] if classes is not None else None, "decoder_output": dec_out,
               "mask_embeds": emb, "mask_features": mask_features, "all_logits": all_logits}
        if self.dense_masks or not self.training:
            materialize_masks(out)
        else:
            # training with the batched criterion: no [B, heads*Q, H, W] tensor is formed.  The matcher samples the mask
            # FEATURES at its points (bilinear sampling is linear, so it commutes with the product) and only the matched
            # masks are evaluated densely (criterion_batched.py); materialize_masks(out) builds the rest on demand.
            out.update({"pred_masks": None, "all_masks": None,
                        "aux_outputs": [{"pred_logits": c, "pred_masks": None} for c in classes[:-1]] if classes is not None
                        else [{"pred_masks": None} for _ in range(emb.shape[1] - 1)]})
        self._finish(out, output)
        return out

    def forward(self, x, mask_features, mask=None):
        assert len(x) == self.num_feature_levels
        extra = self._prepare_extra(mask)
        cdt = self._core_dtype(x)
        if cdt is not None:
            return self._forward_fused(x, mask_features, extra, cdt)
        src, pos, sizes = self._memory(x)
        bs = src[0].shape[1]
        with torch.no_grad():                                                   # the three mask resolutions, once
            pooled = [F.interpolate(mask_features, size=s, mode="bilinear", align_corners=False) for s in sizes]
        query_embed = self.query_embed.weight.unsqueeze(1).expand(-1, bs, -1)
        output = self.query_feat.weight.unsqueeze(1).repeat(1, bs, 1)
        classes, masks, embeds = [], [], []
        cls, msk, attn_mask, dec_out, emb = self.forward_prediction_heads(output, mask_features, sizes[0], pooled[0], extra)
        classes.append(cls), masks.append(msk), embeds.append(emb)
        for i in range(self.num_layers):
            lvl = i % self.num_feature_levels
            attn_mask = attn_mask & ~attn_mask.all(-1, keepdim=True)            # fully-masked rows attend everywhere (:405)
            output = self.transformer_cross_attention_layers[i](output, src[lvl], memory_mask=attn_mask,
                                                                memory_key_padding_mask=None, pos=pos[lvl],
                                                                query_pos=query_embed)
            output = self.transformer_self_attention_layers[i](output, tgt_mask=None, tgt_key_padding_mask=None,
                                                               query_pos=query_embed)
            output = self.transformer_ffn_layers[i](output)
            nxt = (i + 1) % self.num_feature_levels
            cls, msk, attn_mask, dec_out, emb = self.forward_prediction_heads(output, mask_features, sizes[nxt], pooled[nxt], extra)
            classes.append(cls), masks.append(msk), embeds.append(emb)
        assert len(classes) == self.num_layers + 1
        return self._assemble(torch.stack(classes, 0), torch.stack(embeds, 1), dec_out, mask_features, output)

    def _prepare_extra(self, mask):
        return None                                                             # `mask` is ignored (:377-378)

    def _finish(self, out, output):
        pass

    @torch.jit.unused
    def _set_aux_loss(self, outputs_class, outputs_seg_masks):
        if self.mask_classification:
            return [{"pred_logits": a, "pred_masks": b} for a, b in zip(outputs_class[:-1], outputs_seg_masks[:-1])]
        return [{"pred_masks": b} for b in outputs_seg_masks[:-1]]
