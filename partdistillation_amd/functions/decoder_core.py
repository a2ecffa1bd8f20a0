"""The query side of the masked-attention decoder as ONE autograd node with a hand-written backward.

Reference: transformer_decoder/mask2former_transformer_decoder.py:380-447 — L x (masked cross-attention :102-114,
self-attention :44-54, FFN :167-171, all post-norm) with a prediction head (:449-459) in front of every layer.  Run
through eager autograd that loop is ~1 600 launches per training step on 200-row tensors (views, casts, weight slices
and their zero-filled gradients, three-kernel LayerNorm backwards, ...): the step is bound by issuing them, not by
executing them.  Here every sub-layer is a fixed sequence of library GEMMs (torch.addmm / mm on
the bf16 — or fp32 — operands) and hand-written HIP kernels:

  * pd_add_layernorm_{fwd,bwd}  residual + LayerNorm, which also emit the GEMM-dtype copies `y` / `y + query_pos` the
    next projections read and accumulate LayerNorm / bias / positional-table gradients;
  * pd_attn_{fwd,bwd}_d32       split-key masked attention (cross- and self-attention);
  * pd_relu_bwd_colsum / pd_colsum_acc, pd_mem_prep_{fwd,bwd}, pd_attn_mask_u8.

Tensors are seq-first like the reference: row = query * B + image.  The node returns the L+1 normalised decoder
outputs (decoder_norm applied); the class / mask-embedding heads that carry gradients are evaluated once on the stack
of all L+1 outputs by the caller, while the per-layer mask prediction that only feeds the next layer's attention
mask (no gradient: the reference detaches it, :457) runs inside the loop.
"""
from typing import List

import torch
from torch.autograd import Function

from .. import cmdbuf
from .. import lib as _lib
from . import rowwise as rw
from . import conv_bf16, igemm
from . import declayer as dl
from . import smallgemm as sg
from .attention import attn_bwd_raw, attn_fwd_raw


class DecoderSpec:
    """static description handed to DecoderCore (not a tensor argument)."""

    def __init__(self, B, Q, C, nheads, num_layers, sizes, pos_tables, pooled, eps, cdt, num_levels=3):
        self.B, self.Q, self.C, self.H, self.L = B, Q, C, nheads, num_layers
        self.sizes, self.pos, self.pooled, self.eps, self.cdt, self.num_levels = sizes, pos_tables, pooled, eps, cdt, num_levels


import os

# rows up to which the skinny-activation kernels (pd_sgemm_*) replace the GEMM library: forward / input gradient, and the
# weight gradient (whose contraction runs over the rows inside one workgroup, so it stays small)
SMALL_M = int(os.environ.get("PD_SGEMM_MAX_M", "1024"))
SMALL_M_WGRAD = 1024


def _small(x, cdt, limit=None):
    return cdt == torch.bfloat16 and x.shape[0] <= (SMALL_M if limit is None else limit)


KV_IGEMM_ROWS = int(__import__("os").environ.get("PD_KV_IGEMM_ROWS", "4096"))   # memory rows from which the decoder's key / value projections run on pd_igemm_bf16


def _lin(x, w, b, relu=False):
    """x W^T + b (ReLU)"""
    if _small(x, x.dtype) and w.shape[1] % 64 == 0:
        return sg.linear(x, w, b, relu)
    y = torch.addmm(b, x, w.t())
    return torch.relu_(y) if relu else y


def _lin_relu(x, w, b):
    return _lin(x, w, b, True)


def _dgrad(dy, w, relu_ref=None, out=None, accumulate=False):
    """dy W (+= into out) (masked by relu_ref > 0)"""
    if _small(dy, dy.dtype) and w.shape[0] % 64 == 0:
        return sg.dgrad(dy, w, relu_ref, out, accumulate)
    if accumulate:
        out.addmm_(dy, w)
        dx = out
    else:
        dx = torch.mm(dy, w) if out is None else torch.mm(dy, w, out=out)
    if relu_ref is not None:
        rw.relu_bwd_colsum(dx, relu_ref, None)
    return dx


def _wgrad(dy, x, out=None, bias_acc=None, queue=None, big=None):
    """dy^T x -> out (weight gradient);  bias_acc (fp32 accumulator slot, zero on entry) += dy.sum(0).
    queue (sg.WgradQueue): the small-row case is only QUEUED there — the backward pass runs all of them as one launch at its end."""
    if _small(dy, dy.dtype, SMALL_M_WGRAD):
        if queue is not None:
            return queue.add(dy, x, out, bias_acc)
        return sg.wgrad(dy, x, out, bias_acc)
    if big is not None and dy.dtype == torch.bfloat16 and dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0 and dy.is_contiguous() \
            and x.is_contiguous():
        # many rows (key / value projections of the memory tokens): the transpose-read filter-gradient kernel of the backbone
        # (1 x 1 "convolution" over the tokens), queued for the grouped launch — the library's kernel for [256, 32768] x [32768, 256]
        # takes 114 us
        dw = out if out is not None else torch.empty((dy.shape[1], x.shape[1]), dtype=torch.bfloat16, device=dy.device)
        big.append(conv_bf16.rows_entry(dy, x, dw, bias_acc))               # bias gradient: column sums of dy in the same launch
        return dw
    dw = torch.mm(dy.t(), x) if out is None else torch.mm(dy.t(), x, out=out)
    if bias_acc is not None:
        rw.colsum_acc(dy, bias_acc)
    return dw


GROUP_WGRADS = True   # False: one launch per weight gradient (tools/ comparisons)
MULTI_QKV = os.environ.get("PD_MULTI_QKV", "1") != "0"     # the q / k / v projections of an attention block as one launch (pd_sgemm_tn_multi_bf16)
FUSED_HEAD = os.environ.get("PD_FUSED_HEAD", "1") != "0"   # decoder_norm + mask-embedding MLP of a prediction head as one launch (pd_decoder_head_bf16)
# the row-local chains of a layer as four launches (pd_dec_fwd_a / _b, pd_dec_bwd_b / _a: csrc/declayer.hip) instead of ~25: 16 rows per
# workgroup, weights streamed once per workgroup from a packed copy
KV_BATCH = os.environ.get("PD_DEC_KVBATCH", "1") != "0"       # 0: a key and a value projection launch per layer (A/B)
FUSED_LAYER = os.environ.get("PD_DEC_FUSED", "1") != "0"


class _Acc:
    """fp32 accumulators (bias / LayerNorm gradients) carved out of one zero-filled buffer."""

    def __init__(self):
        self.n, self.bias_n = 0, 0

    def take(self, n):
        off = self.n
        self.n += n
        return (off, n)


N_GLOBAL = 11        # query_feat, query_embed, level_embed, dn_w, dn_b, mlp (w,b) x 3
N_LAYER = 18         # cross: in_w in_b out_w out_b n_w n_b | self: same | ffn: w1 b1 w2 b2 n_w n_b


def _drop_backward_of(key, rec, cache):
    for k in [k for k in cache if k[0] == "bwd" and k[1] == id(rec)]:      # a backward recording pins its forward recording's arenas
        cache.pop(k)


_RECS = cmdbuf.LRU(24, _drop_backward_of)      # recorded regions (cmdbuf.Recording), keyed by everything that decides their control flow


def _rec_get(key):
    return _RECS.get(key)


def _flat_params(o):
    if isinstance(o, torch.Tensor):
        return [o]
    if isinstance(o, (list, tuple)):
        return [t for x in o for t in _flat_params(x)]
    return []


def _rec_put(key, rec):
    _RECS.put(key, rec)                                    # beyond 24: the least recently used recordings + their arenas go


class DecoderCore(Function):
    @staticmethod
    def forward(ctx, spec: DecoderSpec, *t):
        ctx.set_materialize_grads(False)
        B, Q, C, H, L, cdt = spec.B, spec.Q, spec.C, spec.H, spec.L, spec.cdt
        nl = spec.num_levels
        xs = t[:nl]
        g = t[nl:nl + N_GLOBAL]
        query_feat, query_embed, level_embed, dn_w, dn_b = g[:5]
        mlp = g[5:11]
        layers = [t[nl + N_GLOBAL + i * N_LAYER: nl + N_GLOBAL + (i + 1) * N_LAYER] for i in range(L)]
        R = Q * B
        # ---- eager prologue (ATen): the learnable queries broadcast over the batch
        qpos = query_embed.contiguous()                                        # fp32 [Q, C]; row r uses qpos[r // B]
        tgt = query_feat.unsqueeze(1).expand(Q, B, C).reshape(R, C)             # fp32 copy
        tgtpos_c = (query_feat + query_embed).to(cdt).unsqueeze(1).expand(Q, B, C).reshape(R, C)
        fused_head = (FUSED_HEAD and cdt == torch.bfloat16 and C == 256 and all(tuple(mlp[j].shape) == (256, 256) for j in (0, 2, 4))
                      and all(mlp[j].is_contiguous() and mlp[j].dtype == torch.bfloat16 for j in range(6)))
        multi = MULTI_QKV and cdt == torch.bfloat16 and C <= 256 and C % 64 == 0
        # the layer loop as a RECORDED region (cmdbuf.py): needs every step inside it to be a pd_* call — the fused prediction head,
        # the multi-problem projections and bf16 pooled mask features (the mask logits then run on pd_sgemm_nn_bf16, not torch.bmm)
        pooled_ok = all(p_.dtype == torch.bfloat16 and p_.shape[2] % 8 == 0 and (p_.is_contiguous() or p_.transpose(1, 2).is_contiguous()) for p_ in spec.pooled)
        use_rec = (cmdbuf.usable() and xs[0].is_cuda and fused_head and multi and pooled_ok and any(ctx.needs_input_grad))
        # fused layer kernels: bf16, C = 256, feed-forward 2048, contiguous bf16 weights / biases, fp32 LayerNorm weights, bf16 pooled mask features
        fused_layer = (FUSED_LAYER and xs[0].is_cuda and fused_head and multi and pooled_ok and Q <= 1024 and igemm.supported(C, C) and
                       all(dl.supported(C, lay[12].shape[0], cdt) and tuple(lay[12].shape) == (dl.FF, C) and tuple(lay[14].shape) == (C, dl.FF) and
                           all(lay[j].dtype == torch.bfloat16 and lay[j].is_contiguous() for j in (0, 1, 2, 3, 6, 7, 8, 9, 12, 13, 14, 15)) and
                           all(lay[j].dtype == torch.float32 for j in (4, 5, 10, 11, 16, 17)) for lay in layers) and
                       dn_w.dtype == torch.float32 and query_embed.dtype == torch.float32)
        ctx.fused_layer = fused_layer
        flat_layers = [p_ for lay in layers for p_ in lay]
        if cmdbuf.DEBUG and not use_rec:
            import sys
            print("[cmdbuf] decoder forward runs eagerly:", dict(cuda=xs[0].is_cuda, fused_head=fused_head, multi=multi,
                  pooled=[(str(p_.dtype), p_.is_contiguous()) for p_ in spec.pooled], needs_grad=any(ctx.needs_input_grad)), file=sys.stderr, flush=True)
        if not use_rec:
            ctx.rec = None
            outs = DecoderCore._fwd_layers(spec, xs, qpos, tgt, tgtpos_c, level_embed, dn_w, dn_b, mlp, layers, fused_head, multi, fused_layer)
        else:
            consts = list(spec.pos[:nl]) + list(spec.pooled[:nl])
            head = list(xs) + [qpos, tgt, tgtpos_c] + consts
            params = [level_embed, dn_w, dn_b] + list(mlp) + flat_layers
            slots = head + params
            key = ("fwd", B, Q, C, H, L, nl, tuple(tuple(x.shape) for x in xs), params[0].data_ptr(), flat_layers[0].data_ptr(), str(xs[0].device),
                   _lib.current_stream(), fused_layer)
            rec = _rec_get(key)
            if rec is None or not rec.matches(slots):
                rec = cmdbuf.Recording(slots, "decoder forward", pinned=range(len(head), len(slots)))
                with rec:
                    outs = DecoderCore._fwd_layers(spec, xs, qpos, tgt, tgtpos_c, level_embed, dn_w, dn_b, mlp, layers, fused_head, multi, fused_layer)
                outs = rec.finish(outs)
                _rec_put(key, rec)
            else:
                outs = rec.replay(slots)
            ctx.rec, ctx.rec_gen = rec, rec.generation
        dec_outs, final_tgt, saved, head_stats, mem, mempos = outs
        ctx.spec, ctx.saved, ctx.head_stats = spec, saved, head_stats
        ctx.mem, ctx.mempos = mem, mempos
        ctx.params = (query_feat, query_embed, level_embed, dn_w, dn_b, mlp, layers)
        ctx.x_shapes = [x.shape for x in xs]
        return dec_outs, final_tgt.view(R, C)          # a view: the node must not own one of its own outputs (reference cycle)

    @staticmethod
    def _fwd_layers(spec, xs, qpos, tgt, tgtpos_c, level_embed, dn_w, dn_b, mlp, layers, fused_head, multi, fused_layer=False):
        """-> (dec_outs, final tgt, saved, head_stats, mem, mempos); inside a recording: pd_* launches and allocations only"""
        if fused_layer:
            return DecoderCore._fwd_layers_fused(spec, xs, qpos, tgt, tgtpos_c, level_embed, dn_w, dn_b, mlp, layers)
        B, Q, C, H, L, cdt = spec.B, spec.Q, spec.C, spec.H, spec.L, spec.cdt
        nl = spec.num_levels
        R, scale = Q * B, 32 ** -0.5
        dev = xs[0].device
        if cmdbuf.active() is not None:
            # these rows go into host-side problem tables (pd_sgemm_tn_multi_bf16) and are saved for the backward region: arena copies
            tgt, tgtpos_c = rw.copy_d2d(torch.empty_like(tgt), tgt), rw.copy_d2d(torch.empty_like(tgtpos_c), tgtpos_c)
        mem, mempos = [], []
        for l in range(nl):
            m, mp = rw.mem_prep_fwd(xs[l], level_embed[l], spec.pos[l], cdt)
            mem.append(m), mempos.append(mp)
        dec_outs = torch.empty((L + 1, R, C), dtype=torch.float32, device=dev)
        saved = []
        head_stats = []

        def head(i, tgt_f32, lvl):
            """decoder_norm -> dec_outs[i]; the (gradient-free) mask prediction for the next layer's attention"""
            if fused_head:                                         # LayerNorm + the 3-layer MLP + the batch-major fp32 copy: one launch
                stats = torch.empty((2, R), dtype=torch.float32, device=dev)
                # the mask embeddings in the dtype of the pooled mask features (bf16 under autocast): the product below then casts nothing
                ef_bf16 = lvl is not None and spec.pooled[lvl].dtype == torch.bfloat16
                ef = torch.empty((B, Q, C), dtype=torch.bfloat16 if ef_bf16 else torch.float32, device=dev) if lvl is not None else None
                _lib.check(_lib.load().pd_decoder_head_bf16(tgt_f32.data_ptr(), dn_w.data_ptr(), dn_b.data_ptr(), float(spec.eps), mlp[0].data_ptr(),
                                                            mlp[1].data_ptr(), mlp[2].data_ptr(), mlp[3].data_ptr(), mlp[4].data_ptr(), mlp[5].data_ptr(),
                                                            dec_outs[i].data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
                                                            ef.data_ptr() if ef is not None else None, int(ef_bf16), R, B, C, rw._stream()))
                head_stats.append((stats[0], stats[1]))
                if lvl is None:
                    return None
                pooled = spec.pooled[lvl]
                pooled_t = pooled.transpose(1, 2)                  # [B, HW, C]: contiguous when the mask features are channels-last
                if ef_bf16 and (pooled.is_contiguous() or pooled_t.is_contiguous()) and Q <= 1024 and C % 64 == 0 and pooled.shape[2] % 8 == 0:
                    # mask logits [Q, HW] = embeddings [Q, C] x pooled mask features [C, HW], per image, on the skinny-row GEMMs
                    # (reference :449: einsum("bqc,bchw->bqhw"); was torch.bmm -> hipBLASLt)
                    logits = torch.empty((B, Q, pooled.shape[2]), dtype=torch.bfloat16, device=dev)
                    for b in range(B):
                        if pooled_t.is_contiguous():
                            sg.linear(ef[b], pooled_t[b], None, False, out=logits[b])          # x [Q, C] @ w [HW, C]^T
                        else:
                            sg.dgrad(ef[b], pooled[b], out=logits[b])                          # dy [Q, C] @ w [C, HW]
                    return rw.attn_mask_u8(logits)
                return rw.attn_mask_u8(torch.bmm(ef, pooled))
            _, _, d_c, _, mean, rstd = _ln_into(tgt_f32, dn_w, dn_b, spec.eps, dec_outs[i], cdt)
            head_stats.append((mean, rstd))
            if lvl is None:
                return None
            e = _lin(_lin_relu(_lin_relu(d_c, mlp[0], mlp[1]), mlp[2], mlp[3]), mlp[4], mlp[5])        # [R, C]
            ef = torch.empty((B, Q, C), dtype=torch.float32, device=dev)
            ef.copy_(e.view(Q, B, C).transpose(0, 1))
            logits = torch.bmm(ef, spec.pooled[lvl])                                                     # [B, Q, HW] fp32
            return rw.attn_mask_u8(logits)

        mask = head(0, tgt, 0)
        for i in range(L):
            lvl = i % nl
            (ciw, cib, cow, cob, cnw, cnb, siw, sib, sow, sob, snw, snb, w1, b1, w2, b2, fnw, fnb) = layers[i]
            # ---- masked cross-attention
            if multi and mem[lvl].shape[0] >= KV_IGEMM_ROWS and igemm.supported(C, C):
                # key / value projections over >= 4 096 memory rows are real GEMMs (level 2: 32 768 x 256 -> 256 twice): on pd_igemm_bf16
                # 2 x 18 us where the skinny multi-problem launch took 89 (107 TFLOP/s); the 200-row query projection stays a skinny launch
                q = _lin(tgtpos_c, ciw[:C], cib[:C])
                k = igemm.linear(mempos[lvl], ciw[C:2 * C], cib[C:2 * C])
                v = igemm.linear(mem[lvl], ciw[2 * C:], cib[2 * C:])
            elif multi:                                            # q, k, v projections: one launch (two inputs, three weight slices)
                q, k, v = sg.linear_multi([(tgtpos_c, ciw[:C], cib[:C]), (mempos[lvl], ciw[C:2 * C], cib[C:2 * C]), (mem[lvl], ciw[2 * C:], cib[2 * C:])])
            else:
                q = _lin(tgtpos_c, ciw[:C], cib[:C])
                k = _lin(mempos[lvl], ciw[C:2 * C], cib[C:2 * C])
                v = _lin(mem[lvl], ciw[2 * C:], cib[2 * C:])
            o, lse = attn_fwd_raw(q, k, v, mask, B, H, scale)
            z, y, y_c, ypos_c, mean, rstd = rw.add_ln_fwd(_lin(o, cow, cob), tgt, cnw, cnb, spec.eps, c_dtype=cdt, want_yc=True,
                                                          pos=qpos, pos_div=B, want_ypos=True)
            cross = (tgtpos_c, q, k, v, mask, o, lse, z, mean, rstd)
            tgt, tgt_c, tgtpos_c = y, y_c, ypos_c
            # ---- self-attention
            if multi:
                q, k, v = sg.linear_multi([(tgtpos_c, siw[:C], sib[:C]), (tgtpos_c, siw[C:2 * C], sib[C:2 * C]), (tgt_c, siw[2 * C:], sib[2 * C:])])
            else:
                q = _lin(tgtpos_c, siw[:C], sib[:C])
                k = _lin(tgtpos_c, siw[C:2 * C], sib[C:2 * C])
                v = _lin(tgt_c, siw[2 * C:], sib[2 * C:])
            o, lse = attn_fwd_raw(q, k, v, None, B, H, scale)
            z, y, y_c, _, mean, rstd = rw.add_ln_fwd(_lin(o, sow, sob), tgt, snw, snb, spec.eps, c_dtype=cdt, want_yc=True)
            slf = (tgtpos_c, tgt_c, q, k, v, o, lse, z, mean, rstd)
            tgt, tgt_c = y, y_c
            # ---- FFN
            h = _lin_relu(tgt_c, w1, b1)
            z, y, _, ypos_c, mean, rstd = rw.add_ln_fwd(_lin(h, w2, b2), tgt, fnw, fnb, spec.eps, c_dtype=cdt, pos=qpos,
                                                        pos_div=B, want_ypos=True)
            ffn = (tgt_c, h, z, mean, rstd, y)
            tgt, tgtpos_c = y, ypos_c
            saved.append((cross, slf, ffn))
            mask = head(i + 1, tgt, (i + 1) % nl if i + 1 < L else None)
        return dec_outs, tgt, saved, head_stats, mem, mempos

    @staticmethod
    def _mask_from_ef(spec, ef, lvl):
        """attention mask of the next layer: mask logits [Q, HW] = embeddings [Q, C] x pooled mask features [C, HW] per image (reference
        :449 einsum + :452-456), thresholded to bytes"""
        B, Q = spec.B, spec.Q
        pooled = spec.pooled[lvl]
        pooled_t = pooled.transpose(1, 2)                          # [B, HW, C]: contiguous when the mask features are channels-last
        if pooled_t.is_contiguous() and sg.bmm_tn_supported(ef, pooled_t):
            return rw.attn_mask_u8(sg.bmm_tn(ef, pooled_t))            # all images in one launch
        logits = torch.empty((B, Q, pooled.shape[2]), dtype=torch.bfloat16, device=ef.device)
        for b in range(B):
            if pooled_t.is_contiguous():
                sg.linear(ef[b], pooled_t[b], None, False, out=logits[b])
            else:
                sg.dgrad(ef[b], pooled[b], out=logits[b])
        return rw.attn_mask_u8(logits)

    @staticmethod
    def _fwd_layers_fused(spec, xs, qpos, tgt, tgtpos_c, level_embed, dn_w, dn_b, mlp, layers):
        """the layer loop on the fused kernels of csrc/declayer.hip: per layer [key / value projections over the memory tokens, masked
        cross-attention] pd_dec_fwd_a [self-attention] pd_dec_fwd_b [mask logits of the next layer's attention mask]"""
        B, Q, C, H, L, cdt = spec.B, spec.Q, spec.C, spec.H, spec.L, spec.cdt
        nl = spec.num_levels
        R, scale = Q * B, 32 ** -0.5
        dev = xs[0].device
        if cmdbuf.active() is not None:
            tgt = rw.copy_d2d(torch.empty_like(tgt), tgt)
        mem, mempos = [], []
        for l in range(nl):
            m, mp = rw.mem_prep_fwd(xs[l], level_embed[l], spec.pos[l], cdt)
            mem.append(m), mempos.append(mp)
        dec_outs = torch.empty((L + 1, R, C), dtype=torch.float32, device=dev)
        saved, head_stats = [], []
        # every weight of the loop in the kernels' block order: one launch per step
        per = 6
        pk = dl.pack([w_ for lay in layers for w_ in (lay[0][:C], lay[2], lay[6], lay[8], lay[12], lay[14])] + [mlp[0], mlp[2], mlp[4]])
        mlp_p = [pk[per * L], mlp[1], pk[per * L + 1], mlp[3], pk[per * L + 2], mlp[5]]

        def qnext(i):
            return (pk[per * i], layers[i][1][:C])

        ws = dl.workspace(R, dev)                                  # the two-launch FFN's slabs (reused by every layer: the launches are ordered)
        # the key / value projections of the layers that share a memory level as ONE product per level and operand (2 x 3 launches instead
        # of 2 x 9; a level's tokens are read once), each layer's [rows, C] result a dense matrix of its own (PdIgemm.out_col_slab: with the
        # results as column slices of one [rows, layers x C] matrix the attention kernels read strided rows and ran 1-3 us slower each).
        # Measured: 12 launches and ~80 us of kernel time fewer under the profiler; the step itself does not move (19.05 vs 19.05 ms, 80-step
        # same-box pairs) although the decoder forward IS on the critical path (10 extra products there: + 0.38 ms) — kept for the launch count
        Kb = Vb = None
        if KV_BATCH and cdt == torch.bfloat16 and C % 128 == 0:
            grp = [[i for i in range(L) if i % nl == l] for l in range(nl)]
            wk = [torch.empty((len(g_) * C, C), dtype=cdt, device=dev) for g_ in grp]
            wv = [torch.empty((len(g_) * C, C), dtype=cdt, device=dev) for g_ in grp]
            bk = [torch.empty(len(g_) * C, dtype=cdt, device=dev) for g_ in grp]
            bv = [torch.empty(len(g_) * C, dtype=cdt, device=dev) for g_ in grp]
            pairs = []
            for l, g_ in enumerate(grp):
                for j, i in enumerate(g_):
                    ciw, cib = layers[i][0], layers[i][1]
                    pairs += [(wk[l][j * C:(j + 1) * C], ciw[C:2 * C]), (wv[l][j * C:(j + 1) * C], ciw[2 * C:]),
                              (bk[l][j * C:(j + 1) * C], cib[C:2 * C]), (bv[l][j * C:(j + 1) * C], cib[2 * C:])]
            for a0 in range(0, len(pairs), rw.MAX_COPY_SEGS):
                rw.copy_segments(pairs[a0:a0 + rw.MAX_COPY_SEGS])
            Kb = [igemm.linear(mempos[l], wk[l], bk[l], out_col_slab=C) if grp[l] else None for l in range(nl)]     # [layers of the level, rows, C]
            Vb = [igemm.linear(mem[l], wv[l], bv[l], out_col_slab=C) if grp[l] else None for l in range(nl)]
        r = dl.fwd_b(None, tgt, qpos, B, None, dn_w, dn_b, mlp_p, qnext(0), spec.eps, dec_outs[0])
        head_stats.append((r["hstats"][0], r["hstats"][1]))
        mask = DecoderCore._mask_from_ef(spec, r["ef"], 0)
        tgtpos_c, qc = r["ypos_c"], r["qc"]
        for i in range(L):
            lvl = i % nl
            (ciw, cib, cow, cob, cnw, cnb, siw, sib, sow, sob, snw, snb, w1, b1, w2, b2, fnw, fnb) = layers[i]
            _, p_co, p_si, p_so, p_w1, p_w2 = pk[per * i:per * i + per]
            # ---- masked cross-attention (the query projection ran in the previous head's launch)
            if Kb is not None:
                j = i // nl
                k, v = Kb[lvl][j], Vb[lvl][j]
            elif mem[lvl].shape[0] >= KV_IGEMM_ROWS:
                k = igemm.linear(mempos[lvl], ciw[C:2 * C], cib[C:2 * C])
                v = igemm.linear(mem[lvl], ciw[2 * C:], cib[2 * C:])
            else:
                k, v = sg.linear_multi([(mempos[lvl], ciw[C:2 * C], cib[C:2 * C]), (mem[lvl], ciw[2 * C:], cib[2 * C:])])
            o, lse = attn_fwd_raw(qc, k, v, mask, B, H, scale)
            z1, st1, y1, y1_c, y1pos_c, sq, sk, sv = dl.fwd_a(o, tgt, qpos, B, p_co, cob, cnw, cnb, spec.eps, p_si, sib)
            cross = (tgtpos_c, qc, k, v, mask, o, lse, z1, st1[0], st1[1])
            # ---- self-attention
            o_s, lse_s = attn_fwd_raw(sq, sk, sv, None, B, H, scale)
            slf = (y1pos_c, y1_c, sq, sk, sv, o_s, lse_s, None, None, None)
            # ---- output projection + LN, FFN + LN, head (+ the next layer's query projection)
            last = i + 1 == L
            r = dl.fwd_b(o_s, y1, qpos, B, (p_so, sob, snw, snb, p_w1, b1, p_w2, b2, fnw, fnb), dn_w, dn_b, None if last else mlp_p,
                         None if last else qnext(i + 1), spec.eps, dec_outs[i + 1], ws)
            slf = slf[:7] + (r["z2"], r["stats2"][0], r["stats2"][1])
            ffn = (r["y2_c"], r["h"], r["z3"], r["stats3"][0], r["stats3"][1], r["y3"])
            saved.append((cross, slf, ffn))
            head_stats.append((r["hstats"][0], r["hstats"][1]))
            tgt, tgtpos_c = r["y3"], r["ypos_c"]
            if not last:
                qc = r["qc"]
                mask = DecoderCore._mask_from_ef(spec, r["ef"], (i + 1) % nl)
        return dec_outs, tgt, saved, head_stats, mem, mempos

    @staticmethod
    def backward(ctx, d_out, d_final):
        spec = ctx.spec
        B, Q, C, H, L, cdt = spec.B, spec.Q, spec.C, spec.H, spec.L, spec.cdt
        nl = spec.num_levels
        R = Q * B
        query_feat, query_embed, level_embed, dn_w, dn_b, mlp, layers = ctx.params
        dev = query_feat.device
        d_out = d_out.contiguous() if d_out is not None else torch.zeros((L + 1, R, C), dtype=torch.float32, device=dev)
        d_fin = d_final.contiguous() if d_final is not None else None
        need_x = tuple(bool(ctx.needs_input_grad[1 + l]) for l in range(nl))
        tgt0 = query_feat.unsqueeze(1).expand(Q, B, C).reshape(R, C)
        rec_f = getattr(ctx, "rec", None)
        args = (spec, ctx.saved, ctx.head_stats, ctx.mem, ctx.mempos, ctx.x_shapes, d_out, d_fin, tgt0, level_embed, dn_w, layers, getattr(ctx, "fused_layer", False))
        if rec_f is None:
            outs = DecoderCore._bwd_layers(*args)
        else:
            if rec_f.generation != ctx.rec_gen:
                raise RuntimeError("the fused decoder ran another forward before this backward: the recorded region's activation arena was "
                                   "overwritten (set PD_CMDBUF=0 for graphs that keep several forward passes alive)")
            flat_layers = [p_ for lay in layers for p_ in lay]
            head = [d_out] + ([d_fin] if d_fin is not None else []) + [tgt0]
            params = [level_embed, dn_w] + flat_layers
            slots = head + params
            key = ("bwd", id(rec_f), d_fin is not None)
            rec = _rec_get(key)
            if rec is None or not rec.matches(slots):
                rec = cmdbuf.Recording(slots, "decoder backward", stable=[rec_f], pinned=range(len(head), len(slots)))
                with rec:
                    outs = DecoderCore._bwd_layers(*args)
                    outs = outs[:-1] + (cmdbuf.Fresh([g_ for lay in outs[-1] for g_ in lay]),)
                outs = rec.finish(outs)
                _rec_put(key, rec)
            else:
                cmdbuf.unalias_grads(_flat_params(ctx.params), rec.owns)
                outs = rec.replay(slots)
            fl = outs[-1]
            outs = outs[:-1] + ([tuple(fl[6 * i:6 * i + 6]) for i in range(L)],)
        buf, slots_, dz0, d_res, d_pos_c, dtoks, d_level, wgrads = outs
        lay, n_bias, s_dnw, s_dnb, s_pos = slots_

        def A(s):
            return buf[s[0]:s[0] + s[1]]
        # ---- eager epilogue (ATen): the learnable queries' gradients, the bias gradients in the parameters' dtype
        d_pos0 = d_pos_c.float().view(Q, B, C).sum(1)
        d_query_feat = (dz0 + d_res).view(Q, B, C).sum(1) + d_pos0
        d_query_embed = A(s_pos).view(Q, C) + d_pos0
        d_xs = [None] * nl
        for l in range(nl):
            if dtoks[l] is not None and need_x[l]:
                _, _, Hh, Ww = ctx.x_shapes[l]
                d_xs[l] = dtoks[l].view(B, Hh, Ww, C).permute(0, 3, 1, 2)
        bias_c = buf[:n_bias] if layers[0][1].dtype == torch.float32 else buf[:n_bias].to(layers[0][1].dtype)

        def Bc(s):
            return bias_c[s[0]:s[0] + s[1]]

        grads = [None]                                                        # spec
        grads += d_xs
        grads += [d_query_feat, d_query_embed, d_level if rec_f is None else d_level.detach(), A(s_dnw), A(s_dnb)] + [None] * 6
        for i in range(L):
            g_ciw, g_cow, g_siw, g_sow, g_w1, g_w2 = wgrads[i]
            s_ = lay[i]
            grads += [g_ciw, Bc(s_["cib"]), g_cow, Bc(s_["cob"]), A(s_["cnw"]), A(s_["cnb"]),
                      g_siw, Bc(s_["sib"]), g_sow, Bc(s_["sob"]), A(s_["snw"]), A(s_["snb"]),
                      g_w1, Bc(s_["b1"]), g_w2, Bc(s_["b2"]), A(s_["fnw"]), A(s_["fnb"])]
        return tuple(grads)

    @staticmethod
    def _bwd_layers(spec, saved, head_stats, mem_f, mempos_f, x_shapes, d_out, d_final, tgt0, level_embed, dn_w, layers, fused_layer=False):
        """-> (accumulator buffer, its slot table, dz of head 0, d(residual stream), d(tgt + query_pos), per-level token gradients, d(level
        embedding), per-layer weight gradients); inside a recording: pd_* launches and allocations only"""
        B, Q, C, H, L, cdt = spec.B, spec.Q, spec.C, spec.H, spec.L, spec.cdt
        nl = spec.num_levels
        R, scale = Q * B, 32 ** -0.5
        dev = d_out.device
        # fp32 accumulators: [linear biases ... | LayerNorm gammas / betas ... | decoder_norm | query_pos]
        acc = _Acc()
        lay = []
        for i in range(L):
            lay.append({"cib": acc.take(3 * C), "cob": acc.take(C), "sib": acc.take(3 * C), "sob": acc.take(C),
                        "b1": acc.take(layers[i][13].numel()), "b2": acc.take(C)})
        n_bias = acc.n
        for i in range(L):
            lay[i].update({"cnw": acc.take(C), "cnb": acc.take(C), "snw": acc.take(C), "snb": acc.take(C),
                           "fnw": acc.take(C), "fnb": acc.take(C)})
        s_dnw, s_dnb, s_pos = acc.take(C), acc.take(C), acc.take(Q * C)
        buf = torch.zeros(acc.n, dtype=torch.float32, device=dev)

        def A(s):
            return buf[s[0]:s[0] + s[1]]

        # input gradients over the memory tokens (dk W_k, dv W_v: 10^3..10^5 rows) on pd_igemm_bf16 with W^T as its weight operand:
        # all layers' transposes in one grouped launch
        big_rows = cdt == torch.bfloat16 and any(not _small(m, cdt) for m in mem_f) and C % 64 == 0
        if big_rows:
            wts = igemm.transposed([layers[i][0][j * C:(j + 1) * C] for i in range(L) for j in (1, 2)])

        def dgrad_mem(dy, i, j, acc_into):
            """dy [rows, C] @ W (the k / v slice of layer i's in_proj weight) (+ acc_into)"""
            w = layers[i][0][j * C:(j + 1) * C]
            if big_rows and not _small(dy, dy.dtype):
                return igemm.linear(dy, wts[2 * i + (j - 1)], res=acc_into)
            if acc_into is None:
                return _dgrad(dy, w)
            _dgrad(dy, w, out=acc_into, accumulate=True)
            return acc_into

        dmem: List = [None] * nl
        dmempos: List = [None] * nl
        wgrads = [None] * L
        wq = sg.WgradQueue() if GROUP_WGRADS else None                       # the ~8 weight gradients per layer: one launch at the end
        big = [] if GROUP_WGRADS else None                                   # ... and the two over the memory tokens: conv_bf16's group
        d_res = d_final                                                      # fp32 gradient w.r.t. the residual stream
        d_pos_c = None                                                       # GEMM-dtype gradient w.r.t. (tgt + query_pos)
        if fused_layer:                                                      # every transposed weight of the loop in block order: one launch
            pkT = dl.pack([w_ for lay_ in layers for w_ in (lay_[0][:C], lay_[2], lay_[6], lay_[8], lay_[12], lay_[14])], transpose=True)
            ws = dl.workspace(R, dev)
            dq_next = None                                                   # d(cross-attention queries) of layer i + 1

            def A2(s0):                                                      # [dgamma | dbeta]: adjacent accumulator slots
                return buf[s0[0]:s0[0] + 2 * s0[1]]
        for i in reversed(range(L)):
            lvl = i % nl
            (ciw, cib, cow, cob, cnw, cnb, siw, sib, sow, sob, snw, snb, w1, b1, w2, b2, fnw, fnb) = layers[i]
            cross, slf, ffn = saved[i]
            if fused_layer:
                cqT, coT, siT, soT, w1T, w2T = pkT[6 * i:6 * i + 6]
                hm, hr = head_stats[i + 1]
                x_c, h, z3, m3, r3, y3 = ffn
                tp_c, t_c, q, k, v, o, lse, z2, m2, r2 = slf
                # (mean, rstd) pairs are the two rows of one [2, R] tensor: the kernels take its base
                dz3_c, dh, dz2, dz2_c, d_os = dl.bwd_b(dq_next, pkT[6 * (i + 1)] if dq_next is not None else None, d_out[i + 1], d_res, y3, hm, dn_w,
                                                       A2(s_dnw), z3, m3, fnw, A2(lay[i]["fnw"]), A(lay[i]["b2"]),
                                                       A(s_pos) if dq_next is not None else None, B, w2T, h, w1T, z2, m2, snw, A2(lay[i]["snw"]),
                                                       A(lay[i]["sob"]), soT, ws)
                g_w2 = _wgrad(dz3_c, h, queue=wq)
                g_w1 = _wgrad(dh, x_c, bias_acc=A(lay[i]["b1"]), queue=wq)
                g_sow = _wgrad(dz2_c, o, queue=wq)
                dq, dk, dv = attn_bwd_raw(q, k, v, None, o, d_os, lse, B, H, scale)
                g_siw = torch.empty_like(siw)
                sb = A(lay[i]["sib"])
                _wgrad(dq, tp_c, g_siw[:C], sb[:C], queue=wq)
                _wgrad(dk, tp_c, g_siw[C:2 * C], sb[C:2 * C], queue=wq)
                _wgrad(dv, t_c, g_siw[2 * C:], sb[2 * C:], queue=wq)
                tp_c, q, k, v, mask, o, lse, z1, m1, r1 = cross
                dz1, dz1_c, d_oc = dl.bwd_a(dq, dk, dv, siT, dz2, z1, m1, cnw, A2(lay[i]["cnw"]), A(lay[i]["cob"]), A(s_pos), B, coT)
                g_cow = _wgrad(dz1_c, o, queue=wq)
                dq, dk, dv = attn_bwd_raw(q, k, v, mask, o, d_oc, lse, B, H, scale)
                g_ciw = torch.empty_like(ciw)
                cb = A(lay[i]["cib"])
                _wgrad(dq, tp_c, g_ciw[:C], cb[:C], queue=wq)
                _wgrad(dk, mempos_f[lvl], g_ciw[C:2 * C], cb[C:2 * C], queue=wq, big=big)
                _wgrad(dv, mem_f[lvl], g_ciw[2 * C:], cb[2 * C:], queue=wq, big=big)
                dmempos[lvl] = dgrad_mem(dk, i, 1, dmempos[lvl])
                dmem[lvl] = dgrad_mem(dv, i, 2, dmem[lvl])
                dq_next = dq
                if i == 0:
                    d_pos_c = _dgrad(dq, ciw[:C])                            # -> the learnable queries (eager epilogue)
                d_res = dz1
                wgrads[i] = (g_ciw, g_cow, g_siw, g_sow, g_w1, g_w2)
                continue
            # ---- head i+1 (decoder_norm of the FFN output)
            hm, hr = head_stats[i + 1]
            dzh, _ = rw.add_ln_bwd(ffn[5], hm, hr, dn_w, dy=d_out[i + 1], dgamma=A(s_dnw), dbeta=A(s_dnb))
            # ---- FFN
            x_c, h, z, mean, rstd, _ = ffn
            dz, dz_c = rw.add_ln_bwd(z, mean, rstd, fnw, dy=dzh, dy2=d_res, dypos_c=d_pos_c, dz_c_dtype=cdt,
                                     dgamma=A(lay[i]["fnw"]), dbeta=A(lay[i]["fnb"]), dbias=A(lay[i]["b2"]),
                                     dpos_acc=A(s_pos) if d_pos_c is not None else None, pos_div=B, out=dzh)
            g_w2 = _wgrad(dz_c, h, queue=wq)
            dh = _dgrad(dz_c, w2, relu_ref=h)                    # through the ReLU: dh *= (h > 0)
            g_w1 = _wgrad(dh, x_c, bias_acc=A(lay[i]["b1"]), queue=wq)
            dx_c = _dgrad(dh, w1)
            # ---- self-attention
            tp_c, t_c, q, k, v, o, lse, z, mean, rstd = slf
            dz, dz_c = rw.add_ln_bwd(z, mean, rstd, snw, dy=dz, dy_c=dx_c, dz_c_dtype=cdt, dgamma=A(lay[i]["snw"]),
                                     dbeta=A(lay[i]["snb"]), dbias=A(lay[i]["sob"]), out=dz)
            g_sow = _wgrad(dz_c, o, queue=wq)
            dq, dk, dv = attn_bwd_raw(q, k, v, None, o, _dgrad(dz_c, sow), lse, B, H, scale)
            g_siw = torch.empty_like(siw)
            sb = A(lay[i]["sib"])
            _wgrad(dq, tp_c, g_siw[:C], sb[:C], queue=wq)
            _wgrad(dk, tp_c, g_siw[C:2 * C], sb[C:2 * C], queue=wq)
            _wgrad(dv, t_c, g_siw[2 * C:], sb[2 * C:], queue=wq)
            d_tp = _dgrad(dq, siw[:C])
            _dgrad(dk, siw[C:2 * C], out=d_tp, accumulate=True)
            d_tc = _dgrad(dv, siw[2 * C:])
            # ---- cross-attention
            tp_c, q, k, v, mask, o, lse, z, mean, rstd = cross
            dz, dz_c = rw.add_ln_bwd(z, mean, rstd, cnw, dy=dz, dy_c=d_tc, dypos_c=d_tp, dz_c_dtype=cdt,
                                     dgamma=A(lay[i]["cnw"]), dbeta=A(lay[i]["cnb"]), dbias=A(lay[i]["cob"]),
                                     dpos_acc=A(s_pos), pos_div=B, out=dz)
            g_cow = _wgrad(dz_c, o, queue=wq)
            dq, dk, dv = attn_bwd_raw(q, k, v, mask, o, _dgrad(dz_c, cow), lse, B, H, scale)
            g_ciw = torch.empty_like(ciw)
            cb = A(lay[i]["cib"])
            _wgrad(dq, tp_c, g_ciw[:C], cb[:C], queue=wq)
            _wgrad(dk, mempos_f[lvl], g_ciw[C:2 * C], cb[C:2 * C], queue=wq, big=big)
            _wgrad(dv, mem_f[lvl], g_ciw[2 * C:], cb[2 * C:], queue=wq, big=big)
            d_pos_c = _dgrad(dq, ciw[:C])                                     # -> previous layer's FFN norm (or the queries)
            dmempos[lvl] = dgrad_mem(dk, i, 1, dmempos[lvl])
            dmem[lvl] = dgrad_mem(dv, i, 2, dmem[lvl])
            d_res = dz
            wgrads[i] = (g_ciw, g_cow, g_siw, g_sow, g_w1, g_w2)

        # ---- head 0
        hm, hr = head_stats[0]
        dz0, _ = rw.add_ln_bwd(tgt0, hm, hr, dn_w, dy=d_out[0], dgamma=A(s_dnw), dbeta=A(s_dnb))

        dtoks = [None] * nl
        d_level = torch.zeros_like(level_embed)
        use_colsum = level_embed.dtype == torch.float32 and C % 128 == 0
        for l in range(nl):
            _, _, Hh, Ww = x_shapes[l]
            if dmem[l] is None:                                              # level unused (fewer layers than levels)
                continue
            dtok = rw.mem_prep_bwd(dmem[l], dmempos[l], B, Hh, Ww, C)
            if use_colsum and dtok.is_contiguous():
                rw.colsum_acc(dtok.view(-1, C), d_level[l])                   # (ATen's column reduction takes 30 us for the 32 768 rows of level 0)
            else:
                torch.sum(dtok.view(-1, C), dim=0, out=d_level[l])
            dtoks[l] = dtok
        if wq is not None:
            wq.run()                                                         # before the bias accumulators are read
            conv_bf16.run_now(big)                                           # their bias sums are read (cast) by the caller
        return buf, (lay, n_bias, s_dnw, s_dnb, s_pos), dz0, d_res, d_pos_c, dtoks, d_level, wgrads


def _ln_into(x_f32, gamma, beta, eps, y_out, cdt):
    """LayerNorm of fp32 rows written straight into `y_out` (a slice of the stacked output) + its GEMM-dtype copy."""
    rows, C = x_f32.shape
    from .. import lib as _lib
    y_c = torch.empty((rows, C), dtype=cdt, device=x_f32.device)
    stats = torch.empty((2, rows), dtype=torch.float32, device=x_f32.device)
    _lib.check(_lib.load().pd_add_layernorm_fwd(None, 0, x_f32.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), None,
                                                y_out.data_ptr(), y_c.data_ptr(), None, 1, None, rw._DT[cdt], stats[0].data_ptr(),
                                                stats[1].data_ptr(), rows, C, rw._stream()))
    return None, y_out, y_c, None, stats[0], stats[1]
