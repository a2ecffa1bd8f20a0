"""The deformable-attention encoder of the pixel decoder as ONE autograd node with a hand-written backward (fp32).

Reference: pixel_decoder/msdeformattn.py MSDeformAttnTransformerEncoder(Layer) (:96-175) around
ops/modules/ms_deform_attn.py:86-131.  Per layer, on the [B*S, 256] token matrix (S = 21 504 at 1024x1024):

    value = value_proj(src);  offsets / logits = sampling_offsets / attention_weights(src + pos)
    loc, attn = pd_msda_prep_fwd(...)                     (softmax + reference point + offset / size, one pass)
    a = pd_msda_forward(value, loc, attn)                 (the HIP sampling core)
    src = LayerNorm(src + output_proj(a))                 (pd_add_layernorm_fwd)
    src = LayerNorm(src + linear2(relu(linear1(src))))    (pd_add_layernorm_fwd, which also emits src + pos for the next layer)

Forward / input-gradient GEMMs are library fp32 GEMMs (addmm / mm), weight gradients the split-K MFMA kernel
pd_gemm_wgrad_f32; everything between them is one hand-written kernel per step instead of the 3-kernel LayerNorm
backward, separate residual / positional adds, softmax / divide / add chains and gradient-accumulation adds that
eager autograd issues.  All arithmetic stays fp32 like the reference (`autocast(enabled=False)`, msdeformattn.py:318).
"""
import os

import torch
from torch.autograd import Function

from .. import MultiScaleDeformableAttention as MSDA
from .. import lib as _lib
from . import rowwise as rw
from .. import cmdbuf
from . import gemm as _gemm
from .gemm import (WgradQueue, gemm_tn_h2, gemm_tn_x3, gemm_tn_x3_pre, gemm_tn_x3_relu_bits, gemm_tn_x3_relumask, gemm_wgrad_acc,
                   h2_bits_supported, pre_supported, relu_bits_supported, row_amax, split3)

def _timed(kind, fn, *args):
    # bench.py's per-launch HIP-event hook lives next to the autograd face of the operator (imported lazily: that module
    # sits under modeling/, which imports this one)
    from ..modeling.pixel_decoder.ops.functions.ms_deform_attn_func import _timed as t
    return t(kind, fn, *args)


def _drop_backward_of(key, rec, cache):
    # a backward recording keeps its forward recording's arenas alive (stable=[rec_f]) and is keyed by id(rec_f): it goes with it
    for k in [k for k in cache if k[0] == "bwd" and k[1] == id(rec)]:
        cache.pop(k)


# recorded regions (cmdbuf.Recording) of the encoder, keyed by everything that decides their control flow.  Each forward / backward pair
# owns >= 2 x (64 + 8) MiB .. ~10 GB of arenas at 1024^2: bounded, least recently used first (variable input shapes)
_RECS = cmdbuf.LRU(int(__import__("os").environ.get("PD_ENCODER_REC_CAP", "6")), _drop_backward_of)


def _msda_timing():
    from ..modeling.pixel_decoder.ops.functions.ms_deform_attn_func import _TIMING as t
    return t


N_LAYER = 16   # so_w so_b aw_w aw_b vp_w vp_b op_w op_b n1_w n1_b l1_w l1_b l2_w l2_b n2_w n2_b


def msda_prep_fwd(offs, logits, ref, shapes, M, L, P):
    """offs [tokens, M*L*P*2], logits [tokens, M*L*P]: fp32, unit column stride, any row stride (column ranges of one
    projection output are fine) -> sampling locations, attention probabilities"""
    tokens = offs.shape[0]
    assert offs.stride(1) == 1 and logits.stride(1) == 1 and offs.dtype == torch.float32 and logits.dtype == torch.float32
    loc = torch.empty((tokens, M, L, P, 2), dtype=torch.float32, device=offs.device)
    attn = torch.empty((tokens, M, L, P), dtype=torch.float32, device=offs.device)
    _lib.check(_lib.load().pd_msda_prep_fwd(offs.data_ptr(), logits.data_ptr(), ref.data_ptr(), shapes.data_ptr(), loc.data_ptr(),
                                            attn.data_ptr(), tokens, M, L, P, offs.stride(0), logits.stride(0), rw._stream()))
    return loc, attn


def msda_prep_bwd(gloc, gattn, attn, shapes, tokens, M, L, P, out=None, amax=None):
    """-> (d_offs, d_logits); with `out` [tokens, 3*M*L*P] they are its column ranges [0, 2MLP) and [2MLP, 3MLP).
    amax [tokens] fp32: receives max |.| of every token's row of `out` (pd_msda_prep_bwd_amax; 8 heads, L P in {8, 12, 16})"""
    n = M * L * P
    if out is None:
        out = torch.empty((tokens, 3 * n), dtype=torch.float32, device=attn.device)
    d_offs, d_logits = out[:, :2 * n], out[:, 2 * n:]
    if amax is not None:
        _lib.check(_lib.load().pd_msda_prep_bwd_amax(gloc.data_ptr(), gattn.data_ptr(), attn.data_ptr(), shapes.data_ptr(), d_offs.data_ptr(),
                                                     d_logits.data_ptr(), amax.data_ptr(), tokens, M, L, P, out.stride(0), out.stride(0), rw._stream()))
        return d_offs, d_logits
    _lib.check(_lib.load().pd_msda_prep_bwd(gloc.data_ptr(), gattn.data_ptr(), attn.data_ptr(), shapes.data_ptr(), d_offs.data_ptr(),
                                            d_logits.data_ptr(), tokens, M, L, P, out.stride(0), out.stride(0), rw._stream()))
    return d_offs, d_logits


def prep_amax_supported(M, L, P):
    return M == 8 and P % 2 == 0 and L * P in (8, 12, 16)


def msda_forward_amax(value, shapes, lsi, loc, attn, im2col_step, row_amax):
    """MSDA.ms_deform_attn_forward that also leaves the absolute maxima of the output rows in `row_amax` [B * S] (zero-filled by the
    caller; pd_msda_forward_amax: fp32, 32 channels per head, 3 levels, 4 points)"""
    B, S, M, D = value.shape
    Lq, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    out = torch.empty((B, Lq, M * D), dtype=value.dtype, device=value.device)
    _lib.check(_lib.load().pd_msda_forward_amax(value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), loc.data_ptr(), attn.data_ptr(), out.data_ptr(),
                                                row_amax.data_ptr(), B, S, M, D, L, Lq, P, int(im2col_step), _lib.PD_F32, rw._stream()))
    return out


FUSED_MSDA = os.environ.get("PD_MSDA_FUSED", "1") != "0"   # softmax + sampling locations inside the MSDA kernels (pd_msda_fused_*): no prep launches


def msda_fused_supported(B, S, M, D, L, Lq, P):
    return FUSED_MSDA and bool(_lib.load().pd_msda_fused_supported(B, S, M, D, L, Lq, P))


def msda_fused_forward(value, shapes, lsi, oa, ref, row_amax):
    """value [B, S, M, 32]; oa [B * S, >= 3 M L P] fp32 = the raw sampling_offsets | attention_weights projection output; ref [B * S, L, 2]
    -> (output [B * S, M * 32], softmax statistics [B * S * M, 2]); row_amax [B * S] (zero-filled) receives the output rows' maxima"""
    B, S, M, D = value.shape
    L = int(shapes.shape[0])
    P = oa.shape[1] // (3 * M * L)
    assert oa.stride(1) == 1 and oa.dtype == torch.float32 and ref.is_contiguous() and ref.numel() == B * S * L * 2
    out = torch.empty((B * S, M * D), dtype=torch.float32, device=value.device)
    stats = torch.empty((B * S * M, 2), dtype=torch.float32, device=value.device)
    _lib.check(_lib.load().pd_msda_fused_forward(value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), oa.data_ptr(), oa.stride(0), ref.data_ptr(),
                                                 out.data_ptr(), stats.data_ptr(), row_amax.data_ptr() if row_amax is not None else None,
                                                 B, S, M, D, L, S, P, rw._stream()))
    return out, stats


def msda_fused_backward(value, shapes, lsi, oa, ref, stats, out, grad_out):
    """-> (grad_value [B, S, M, 32], d_oa [B * S, 3 M L P] = the gradient of the raw projection output, its rows' maxima [B * S])"""
    B, S, M, D = value.shape
    L = int(shapes.shape[0])
    n_oa = oa.shape[1]
    P = n_oa // (3 * M * L)
    # grad_value and the row maxima are both accumulated into (atomics): one allocation, so that the operator clears them with one memset
    both = torch.empty(value.numel() + B * S, dtype=torch.float32, device=value.device)
    gv, d_oa_am = both[:value.numel()].view(value.shape), both[value.numel():]
    d_oa = torch.empty((B * S, n_oa), dtype=torch.float32, device=value.device)
    scratch = torch.empty(B * S * M * L, dtype=torch.float32, device=value.device)
    grad_out = grad_out if grad_out.is_contiguous() else grad_out.contiguous()
    _lib.check(_lib.load().pd_msda_fused_backward(value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), oa.data_ptr(), oa.stride(0), ref.data_ptr(),
                                                  stats.data_ptr(), out.data_ptr(), grad_out.data_ptr(), gv.data_ptr(), d_oa.data_ptr(), d_oa.stride(0),
                                                  d_oa_am.data_ptr(), scratch.data_ptr(), B, S, M, D, L, S, P, rw._stream()))
    return gv, d_oa, d_oa_am


class EncoderSpec:
    def __init__(self, n_heads, n_levels, n_points, eps, im2col_step, reference_points, spatial_shapes, level_start_index):
        self.M, self.L, self.P, self.eps, self.im2col_step = n_heads, n_levels, n_points, eps, im2col_step
        self.ref, self.shapes, self.lsi = reference_points, spatial_shapes, level_start_index


USE_X3 = True        # the FFN GEMMs (1024-wide) through pd_gemm_tn_f32x3; tools / tests switch it off to compare
GROUP_WGRADS = os.environ.get("PD_GROUP_WGRADS", "1") != "0"  # the 5 weight gradients of every layer are queued during the backward pass and run as ONE grouped launch at its
                     # end (functions/gemm.WgradQueue -> pd_gemm_wgrad_f32x3_grouped) instead of 30 launches + 30 reduce launches
X3_PROJ = os.environ.get("PD_X3_PROJ", "1") != "0"       # the 256-wide projections (value / offsets+weights / output and their input gradients) on the same kernel: with the
                     # round-3 epilogue (stores no longer serialised on vmcnt(0)) 256 <- 256 at M = 43 008 runs 47.7 us against the
                     # library's ~57 (tools/probes/gemm_planes_probe.hip); False: torch.addmm / mm (Tensile fp32)
H2_WGRAD = os.environ.get("PD_H2_WGRAD", "1") != "0"   # ... and the weight gradients (pd_gemm_wgrad_f16x2_grouped: slab-wise scales)
H2 = os.environ.get("PD_H2", "1") != "0"   # forward / input-gradient GEMMs on the fp16 two-plane kernel (pd_gemm_tn_f16x2: 3 products per term instead of 6, operand rows
                     # scaled by powers of two from their absolute maxima, which the LayerNorm kernels and the GEMM epilogues emit as they write the
                     # rows): 1024 <- 256 at M = 43 008 96 us vs 132 (x3), 256 <- 1024 85 vs 132, 256 <- 256 29 vs 44 (tools/bench_gemm_h2.py)
PRESPLIT = False     # weight operand split into its bf16 planes once per use (pd_split3_bf16 + pd_gemm_tn_f32x3_pre).  Bit-identical results;
                     # measured 174.6 vs 178.6 us (1024 <- 256) and 157 vs 151 us (256 <- 1024) plus 8 us per split launch: no gain, so off


def _proj(x, w, b=None):
    """x [T, K] @ w [N, K]^T (+ b), fp32: a 256-wide projection of the encoder layer"""
    if X3_PROJ and USE_X3 and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.shape[1] % 4 == 0 and x.stride(1) == 1 \
            and x.stride(0) % 4 == 0:
        return gemm_tn_x3(x, w, b)
    return torch.addmm(b, x, w.t()) if b is not None else torch.mm(x, w.t())


def _ffn_gemm(x, w, b=None, relu=False):
    """x [T, K] @ w [N, K]^T (+ b) (ReLU), fp32.  The two FFN GEMMs of an encoder layer and their input gradients are the
    large ones of the layer (22.5 GFLOP each at config 2): they run on the bf16 matrix cores with every fp32 operand split
    exactly into three bf16 values (include/pd_gemm.h: pd_gemm_tn_f32x3, error vs fp64 at the library fp32 GEMM's level;
    measured 174 vs 211 us and 166 vs 183 us).  The 256-wide projections stay with the library (no gain there)."""
    if USE_X3 and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.shape[1] % 4 == 0:
        return gemm_tn_x3(x, w, b, relu)
    if relu:
        return torch._addmm_activation(b, x, w.t(), use_gelu=False)
    return torch.addmm(b, x, w.t()) if b is not None else torch.mm(x, w.t())


import ctypes as _ct


class _TrProblem(_ct.Structure):                                     # PdTransposeProblem, include/pd_rowwise.h
    _fields_ = [("src", _ct.c_void_p), ("dst", _ct.c_void_p), ("src_batch_stride", _ct.c_int64), ("src_row_stride", _ct.c_int64),
                ("batch", _ct.c_int32), ("rows", _ct.c_int32), ("cols", _ct.c_int32), ("reserved", _ct.c_int32)]


class _Stack:
    """[layers, ...] tensor indexed by layer, stored in either layer order (rev: the last layer first)"""

    def __init__(self, t, rev):
        self.t, self.rev = t, rev

    def __getitem__(self, i):
        return self.t[self.t.shape[0] - 1 - i if self.rev else i]


class EncoderCore(Function):
    @staticmethod
    def forward(ctx, spec: EncoderSpec, src, pos, *params):
        ctx.set_materialize_grads(False)
        B, S, C = src.shape
        M, L, P = spec.M, spec.L, spec.P
        nl = len(params) // N_LAYER
        T = B * S
        src2 = src.reshape(T, C)
        src2 = src2 if src2.is_contiguous() else src2.contiguous()
        pos2 = pos.reshape(T, C)
        pos2 = pos2 if pos2.is_contiguous() else pos2.contiguous()
        ref = spec.ref.reshape(T, L, 2)
        ref = ref if ref.is_contiguous() else ref.contiguous()
        saved = []
        x = src2
        h2 = H2 and USE_X3 and X3_PROJ and src2.is_cuda and C % 4 == 0 and src2.dtype == torch.float32 and pos2.dtype == torch.float32
        ctx.h2 = h2
        if h2:                                  # (query = src + pos is formed inside, with the operands' row maxima: pd_add_rows_amax_f32)
            return EncoderCore._forward_h2(ctx, spec, src2, pos2, None, ref, params, (B, S, C, nl))
        q = src2 + pos2
        for i in range(nl):
            (so_w, so_b, aw_w, aw_b, vp_w, vp_b, op_w, op_b, n1_w, n1_b, l1_w, l1_b, l2_w, l2_b, n2_w, n2_b) = params[i * N_LAYER:(i + 1) * N_LAYER]
            value = _proj(x, vp_w, vp_b)
            # sampling_offsets and attention_weights read the same input: one GEMM against their stacked weights
            w_oa, b_oa = torch.cat([so_w, aw_w]), torch.cat([so_b, aw_b])
            oa = _proj(q, w_oa, b_oa)                                                  # [T, 2MLP + MLP]
            n_off = so_w.shape[0]
            loc, attn = msda_prep_fwd(oa[:, :n_off], oa[:, n_off:], ref, spec.shapes, M, L, P)
            v4, loc6, attn5 = value.view(B, S, M, C // M), loc.view(B, S, M, L, P, 2), attn.view(B, S, M, L, P)
            a = _timed("fwd", MSDA.ms_deform_attn_forward, v4, spec.shapes, spec.lsi, loc6, attn5, spec.im2col_step).view(T, C)
            z1, y1, _, _, m1, r1 = rw.add_ln_fwd(_proj(a, op_w, op_b), x, n1_w, n1_b, spec.eps)
            hbits = None
            pre = USE_X3 and PRESPLIT and pre_supported(T, l1_w.shape[0], l1_w.shape[1]) and pre_supported(T, l2_w.shape[0], l2_w.shape[1])
            if pre:                       # weights split into their bf16 planes once here, not by every row tile of the GEMMs
                h, hbits = gemm_tn_x3_pre(y1, split3(l1_w), l1_b, mode=1, want_bits=True)
            elif USE_X3 and relu_bits_supported(T, l1_w.shape[0]):
                h, hbits = gemm_tn_x3_relu_bits(y1, l1_w, l1_b)                       # + the sign bits the backward's epilogue consumes
            else:
                h = _ffn_gemm(y1, l1_w, l1_b, relu=True)                             # bias + ReLU in the GEMM epilogue
            ffn2 = gemm_tn_x3_pre(h, split3(l2_w), l2_b) if pre else _ffn_gemm(h, l2_w, l2_b)
            last = i == nl - 1
            z2, y2, _, ypos, m2, r2 = rw.add_ln_fwd(ffn2, y1, n2_w, n2_b, spec.eps, c_dtype=torch.float32,
                                                    pos=pos2, pos_div=1, want_ypos=not last)
            saved.append((x, q, v4, loc6, attn5, a, z1, m1, r1, y1, h, z2, m2, r2, w_oa, hbits))
            x, q = y2, ypos
        ctx.spec, ctx.saved, ctx.params, ctx.dims = spec, saved, params, (B, S, C, nl)
        return x.view(B, S, C)

    @staticmethod
    def _forward_h2(ctx, spec, src2, pos2, q, ref, params, dims):
        """the same layer sequence with every GEMM on pd_gemm_tn_f16x2: each operand travels with the absolute maxima of its rows
        (LayerNorm outputs and the FFN's hidden activations get them from the kernel that writes them).
        The layer loop is a RECORDED region (cmdbuf.py): its ~60 launches are issued by one C call from the second step on."""
        B, S, C, nl = dims
        P_ = lambda i, j: params[i * N_LAYER + j]
        # weights with 256 input columns of all layers in ONE matrix (rows: so, aw, vp, op, l1 per layer) + one row-maxima launch;
        # the slices of that copy are the GEMM operands (so + aw adjacent = the stacked sampling_offsets / attention_weights matrix)
        n_off, n_aw = P_(0, 0).shape[0], P_(0, 2).shape[0]
        wk = torch.cat([P_(i, j) for i in range(nl) for j in (0, 2, 4, 6, 10)])
        b_oa_all = torch.cat([P_(i, j) for i in range(nl) for j in (1, 3)]).view(nl, n_off + n_aw)
        l2_all = torch.cat([P_(i, 12) for i in range(nl)])                                  # [nl * C, ffn]
        ctx.wk = wk
        use_rec = (cmdbuf.usable() and src2.is_cuda and any(ctx.needs_input_grad) and not _gemm._TIMING["on"]
                   and not _msda_timing()["on"])
        if not use_rec:
            ctx.rec = None
            x, saved, saved_am = EncoderCore._layers_h2(spec, src2, pos2, q, ref, wk, b_oa_all, l2_all, params, dims)
        else:
            slots = [src2, pos2, ref, wk, b_oa_all, l2_all, spec.shapes, spec.lsi] + list(params)
            key = ("fwd", B, S, C, nl, spec.M, spec.L, spec.P, str(src2.device), _lib.current_stream())
            rec = _RECS.get(key)
            if rec is None or not rec.matches(slots):
                rec = cmdbuf.Recording(slots, "encoder forward")
                with rec:
                    outs = EncoderCore._layers_h2(spec, src2, pos2, q, ref, wk, b_oa_all, l2_all, params, dims)
                x, saved, saved_am = rec.finish(outs)
                _RECS.put(key, rec)
            else:
                x, saved, saved_am = rec.replay(slots)
            ctx.rec, ctx.rec_gen = rec, rec.generation
        ctx.saved_am = saved_am
        ctx.spec, ctx.saved, ctx.params, ctx.dims = spec, saved, params, (B, S, C, nl)
        return x.view(B, S, C)

    @staticmethod
    def _layers_h2(spec, src2, pos2, q, ref, wk, b_oa_all, l2_all, params, dims):
        """-> (output tokens [T, C], per-layer saved tensors, per-layer saved row maxima); pd_* launches and allocations only"""
        B, S, C, nl = dims
        M, L, P = spec.M, spec.L, spec.P
        T = B * S
        P_ = lambda i, j: params[i * N_LAYER + j]
        n_off, n_aw, n_l1 = P_(0, 0).shape[0], P_(0, 2).shape[0], P_(0, 10).shape[0]
        per = n_off + n_aw + 2 * C + n_l1
        x_am = q_am = None
        if q is None and src2.dtype == torch.float32 and pos2.dtype == torch.float32 and C % 4 == 0:
            # q = src + pos, both operands' row maxima and — inside a recording — a copy of src at an arena address (the backward pass
            # writes the addresses of layer 0's operands into the host-side table of its grouped weight-gradient launch) in ONE launch
            q, src_c, x_am, q_am = rw.add_rows_amax(src2, pos2, copy_a=cmdbuf.active() is not None)
            src2 = src_c if src_c is not None else src2
        else:
            if q is None:
                assert cmdbuf.active() is None
                q = src2 + pos2
            if cmdbuf.active() is not None:
                src2, q = rw.copy_d2d(torch.empty_like(src2), src2), rw.copy_d2d(torch.empty_like(q), q)
        wk_am = row_amax(wk)
        l2_am = row_amax(l2_all)
        fwd_amax = C // M == 32 and L == 3 and P == 4 and src2.dtype == torch.float32       # the MSDA kernel that can emit its rows' maxima
        zam = torch.zeros((2 * nl if fwd_amax else nl, T), dtype=torch.float32, device=src2.device)   # atomic-max targets: FFN epilogues, MSDA outputs
        h_am_all, a_am_all = zam[:nl], (zam[nl:] if fwd_amax else None)
        x = src2
        if x_am is None:
            x_am, q_am = row_amax(src2), row_amax(q)
        saved, saved_am = [], []
        for i in range(nl):
            (so_w, so_b, aw_w, aw_b, vp_w, vp_b, op_w, op_b, n1_w, n1_b, l1_w, l1_b, l2_w, l2_b, n2_w, n2_b) = params[i * N_LAYER:(i + 1) * N_LAYER]
            o = i * per
            w_oa, oa_wam = wk[o:o + n_off + n_aw], wk_am[o:o + n_off + n_aw]
            o += n_off + n_aw
            vp_c, vp_wam = wk[o:o + C], wk_am[o:o + C]
            op_c, op_wam = wk[o + C:o + 2 * C], wk_am[o + C:o + 2 * C]
            l1_c, l1_wam = wk[o + 2 * C:o + 2 * C + n_l1], wk_am[o + 2 * C:o + 2 * C + n_l1]
            value = gemm_tn_h2(x, vp_c, vp_b, a_amax=x_am, b_amax=vp_wam)
            oa = gemm_tn_h2(q, w_oa, b_oa_all[i], a_amax=q_am, b_amax=oa_wam)           # [T, 2MLP + MLP]
            v4 = value.view(B, S, M, C // M)
            fused = fwd_amax and n_off == 2 * n_aw and msda_fused_supported(B, S, M, C // M, L, S, P)
            if fused:
                # softmax over the head's L P logits and reference point + offset / (W, H) inside the kernel: `oa` and the 8-byte softmax
                # statistics take the places of loc / attn among the saved tensors (the backward kernel re-forms both)
                a_am = a_am_all[i]
                a, stats = _timed("fwd", msda_fused_forward, v4, spec.shapes, spec.lsi, oa, ref, a_am)
                loc6, attn5 = oa, stats
            else:
                loc, attn = msda_prep_fwd(oa[:, :n_off], oa[:, n_off:], ref, spec.shapes, M, L, P)
                loc6, attn5 = loc.view(B, S, M, L, P, 2), attn.view(B, S, M, L, P)
            if fused:
                pass
            elif fwd_amax:
                a_am = a_am_all[i]
                a = _timed("fwd", msda_forward_amax, v4, spec.shapes, spec.lsi, loc6, attn5, spec.im2col_step, a_am).view(T, C)
            else:
                a = _timed("fwd", MSDA.ms_deform_attn_forward, v4, spec.shapes, spec.lsi, loc6, attn5, spec.im2col_step).view(T, C)
                a_am = row_amax(a)
            z1, y1, _, _, m1, r1, y1_am, _ = rw.add_ln_fwd(gemm_tn_h2(a, op_c, op_b, a_amax=a_am, b_amax=op_wam), x, n1_w, n1_b, spec.eps, amax=True)
            h_am = h_am_all[i]
            if h2_bits_supported(T, n_l1):
                h, hbits = gemm_tn_h2(y1, l1_c, l1_b, mode=1, want_bits=True, a_amax=y1_am, b_amax=l1_wam, c_amax=h_am)
            else:
                h, hbits = gemm_tn_h2(y1, l1_c, l1_b, mode=1, a_amax=y1_am, b_amax=l1_wam, c_amax=h_am), None
            ffn2 = gemm_tn_h2(h, l2_all[i * C:(i + 1) * C], l2_b, a_amax=h_am, b_amax=l2_am[i * C:(i + 1) * C])
            last = i == nl - 1
            z2, y2, _, ypos, m2, r2, y2_am, ypos_am = rw.add_ln_fwd(ffn2, y1, n2_w, n2_b, spec.eps, c_dtype=torch.float32,
                                                                     pos=pos2, pos_div=1, want_ypos=not last, amax=True)
            saved.append((x, q, v4, loc6, attn5, a, z1, m1, r1, y1, h, z2, m2, r2, ref if fused else None, hbits))
            saved_am.append((x_am, q_am, a_am, y1_am, h_am))
            x, q, x_am, q_am = y2, ypos, y2_am, ypos_am
        return x, saved, saved_am

    @staticmethod
    def _stack_t(params, nl, j):
        """[nl, K, N]: the transposes of weight j of all layers (the "B[N, K]" operands of the input-gradient GEMMs), one copy launch"""
        ws = [params[i * N_LAYER + j] for i in range(nl)]
        # the layers' parameters usually sit at one spacing in a flat buffer (engine/flat_params.py, last layer first): the stack is
        # then a strided VIEW and the transposed copy the only launch
        d = (ws[1].data_ptr() - ws[0].data_ptr()) if nl > 1 else 0
        es = ws[0].element_size()
        if nl > 1 and d != 0 and d % es == 0 and all(w.is_contiguous() and w.shape == ws[0].shape for w in ws) \
                and all(ws[i].data_ptr() - ws[0].data_ptr() == i * d for i in range(nl)) \
                and all(w.untyped_storage().data_ptr() == ws[0].untyped_storage().data_ptr() for w in ws):
            N_, K_ = ws[0].shape
            base = ws[0] if d > 0 else ws[-1]                 # lowest address first; _Stack maps the layer index back
            return _Stack(torch.as_strided(base, (nl, N_, K_), (abs(d) // es, K_, 1)).transpose(1, 2).contiguous(), d < 0)
        return _Stack(torch.stack(ws).transpose(1, 2).contiguous(), False)

    @staticmethod
    def _stacks_t_one_launch(params, nl, wk, n_oa, per, C):
        """the five transposed weight stacks of _backward_h2 (linear1, linear2, output_proj, value_proj of every layer from the flat
        parameter buffer, and the stacked sampling_offsets + attention_weights rows of `wk`) by ONE pd_transpose_batched_f32 launch into one
        buffer — or None when the layers' weights do not sit at one spacing (then the per-stack ATen copies run)"""
        import ctypes
        from .. import lib as _lib
        probs, views, total = [], [], 0
        for j in (10, 12, 6, 4):
            ws = [params[i * N_LAYER + j] for i in range(nl)]
            d = (ws[1].data_ptr() - ws[0].data_ptr()) if nl > 1 else 0
            es = ws[0].element_size()
            ok = (ws[0].is_cuda and ws[0].dtype == torch.float32 and all(w.is_contiguous() and w.shape == ws[0].shape for w in ws)
                  and (nl == 1 or (d != 0 and d % es == 0 and all(ws[i].data_ptr() - ws[0].data_ptr() == i * d for i in range(nl)))))
            if not ok:
                return None
            N_, K_ = ws[0].shape
            base = ws[0] if d >= 0 else ws[-1]
            probs.append((base.data_ptr(), abs(d) // es if nl > 1 else N_ * K_, K_, nl, N_, K_, d < 0))
        if not (wk.is_cuda and wk.dtype == torch.float32 and wk.is_contiguous()):
            return None
        probs.append((wk.data_ptr(), per * C, C, nl, n_oa, C, False))
        for p_ in probs:
            total += p_[3] * p_[4] * p_[5]
        buf = torch.empty(total, dtype=torch.float32, device=wk.device)
        descs = (_TrProblem * len(probs))()
        off = 0
        for d_, (src, bs, rs, batch, rows, cols, rev) in zip(descs, probs):
            d_.src, d_.dst, d_.src_batch_stride, d_.src_row_stride = src, buf.data_ptr() + 4 * off, bs, rs
            d_.batch, d_.rows, d_.cols, d_.reserved = batch, rows, cols, 0
            views.append(_Stack(buf[off:off + batch * rows * cols].view(batch, cols, rows), rev))
            off += batch * rows * cols
        _lib.check(_lib.load().pd_transpose_batched_f32(ctypes.byref(descs), len(probs), _lib.current_stream()))
        return views

    @staticmethod
    def _backward_h2(ctx, d_out):
        """backward of _forward_h2: an eager prologue (the transposed weight stacks: ATen copies), the layer loop as a recorded region
        (~110 launches incl. the grouped weight gradients: one C call from the second step on), three elementwise sums at the end"""
        spec, params = ctx.spec, ctx.params
        B, S, C, nl = ctx.dims
        T = B * S
        need_w = any(ctx.needs_input_grad[3:])
        n_off, n_aw, n_l1 = params[0].shape[0], params[2].shape[0], params[10].shape[0]
        per = n_off + n_aw + 2 * C + n_l1
        wk = ctx.wk
        stacks5 = EncoderCore._stacks_t_one_launch(params, nl, wk, n_off + n_aw, per, C)
        if stacks5 is not None:
            l1_t, l2_t, op_t, vp_t, oa_t = stacks5
        else:
            st = EncoderCore._stack_t
            l1_t, l2_t, op_t, vp_t = st(params, nl, 10), st(params, nl, 12), st(params, nl, 6), st(params, nl, 4)
            oa_t = _Stack(torch.as_strided(wk, (nl, n_off + n_aw, C), (per * C, C, 1)).transpose(1, 2).contiguous(), False)
        dy = d_out.reshape(T, C)
        dy = dy if dy.is_contiguous() else dy.contiguous()
        stacks = (l1_t, l2_t, op_t, vp_t, oa_t)
        ref = spec.ref.reshape(T, spec.L, 2)                       # (the fused MSDA backward re-forms the sampling locations from it)
        ref = ref if ref.is_contiguous() else ref.contiguous()
        rec_f = getattr(ctx, "rec", None)
        if rec_f is None:
            dy, dy2, dyq, d_pos, grads = EncoderCore._bwd_layers_h2(spec, params, ctx.dims, ctx.saved, ctx.saved_am, dy, stacks, need_w, ref)
        else:
            if rec_f.generation != ctx.rec_gen:
                raise RuntimeError("the fused encoder ran another forward before this backward: the recorded region's activation arena was "
                                   "overwritten (set PD_CMDBUF=0 for graphs that keep several forward passes alive)")
            slots = [dy] + [k.t for k in stacks] + [spec.shapes, spec.lsi, ref] + list(params)
            key = ("bwd", id(rec_f), need_w, tuple(k.rev for k in stacks))
            rec = _RECS.get(key)
            if rec is None or not rec.matches(slots):
                rec = cmdbuf.Recording(slots, "encoder backward", stable=[rec_f])
                with rec:
                    outs = EncoderCore._bwd_layers_h2(spec, params, ctx.dims, ctx.saved, ctx.saved_am, dy, stacks, need_w, ref)
                    outs = outs[:4] + (cmdbuf.Fresh(outs[4]),)
                dy, dy2, dyq, d_pos, grads = rec.finish(outs)
                _RECS.put(key, rec)
            else:
                cmdbuf.unalias_grads(params, rec.owns)
                dy, dy2, dyq, d_pos, grads = rec.replay(slots)
        if all(t_.is_cuda and t_.dtype == torch.float32 and t_.is_contiguous() and t_.numel() == dy.numel() for t_ in (dy, dy2, dyq, d_pos)) \
                and dy.numel() % 4 == 0:
            # d(src) = (dy + dy2) + dyq and d(pos) = d_pos + dyq by one launch (the same sums, in the same order)
            d_src, d_pos_out = torch.empty_like(dy), torch.empty_like(d_pos)
            _lib.check(_lib.load().pd_sum3_sum2_f32(dy.data_ptr(), dy2.data_ptr(), dyq.data_ptr(), d_pos.data_ptr(), d_src.data_ptr(),
                                                    d_pos_out.data_ptr(), dy.numel(), _lib.current_stream()))
            d_pos = d_pos_out
        else:
            d_pos = d_pos + dyq if rec_f is not None else d_pos.add_(dyq)      # (never in place on an arena tensor another replay re-reads)
            d_src = dy + dy2
            d_src += dyq
        if not need_w:
            grads = [None] * len(grads)
        return (None, d_src.view(B, S, C), d_pos.view(B, S, C), *grads)

    @staticmethod
    def _bwd_layers_h2(spec, params, dims, saved, saved_am, dy, stacks, need_w, ref):
        """-> (dy, dy2, dyq: the three fp32 terms of d(src), d_pos without its last term, parameter gradients); pd_* launches and
        allocations only (recordable)"""
        B, S, C, nl = dims
        M, L, P = spec.M, spec.L, spec.P
        T = B * S
        dev = dy.device
        l1_t, l2_t, op_t, vp_t, oa_t = stacks
        wh2 = H2_WGRAD
        queue = WgradQueue(h2=wh2) if (need_w and GROUP_WGRADS and _gemm.WGRAD_X3) else None
        if not need_w:
            wgrad = lambda *a, **k: None
        elif queue is not None:
            wgrad = (lambda dy_, x_, dw_, db_=None, ya=None, xa=None: queue.add(dy_, x_, dw_, db_, ya if wh2 else None, xa if wh2 else None))
        else:
            wgrad = (lambda dy_, x_, dw_, db_=None, ya=None, xa=None: gemm_wgrad_acc(dy_, x_, dw_, db_, h2=wh2, y_amax=ya if wh2 else None,
                                                                                     x_amax=xa if wh2 else None))
        # every parameter gradient of the encoder lives in ONE zero-filled fp32 buffer (a single memset): the LayerNorm /
        # ReLU kernels and the split-K weight-gradient GEMMs all accumulate (+=) into their slices
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        n_off = params[0].shape[0]
        n_oa = n_off + params[2].shape[0]                               # stacked sampling_offsets + attention_weights rows
        oa_base, oa_per = total, n_oa * C + (n_oa + 3) // 4 * 4
        total += nl * oa_per
        buf = torch.zeros(total, dtype=torch.float32, device=dev)

        def OA(i):
            o = oa_base + i * oa_per
            return buf[o:o + n_oa * C].view(n_oa, C), buf[o + n_oa * C:o + n_oa * C + n_oa]

        def G(i, j):
            p = params[i * N_LAYER + j]
            o = offs[i * N_LAYER + j]
            return buf[o:o + p.numel()].view(p.shape)
        d_pos = torch.zeros((T, C), dtype=torch.float32, device=dev)
        grads = [None] * (nl * N_LAYER)
        t_am = lambda w: _Stack(row_amax(w.t.view(-1, w.t.shape[2])).view(nl, w.t.shape[1]), w.rev)
        l1_tam, l2_tam, op_tam, vp_tam, oa_tam = t_am(l1_t), t_am(l2_t), t_am(op_t), t_am(vp_t), t_am(oa_t)
        dh_am_all = torch.zeros((nl, T), dtype=torch.float32, device=dev)
        dy2 = dyq = None                                       # further fp32 terms of d(src_l): via value_proj, via (src + pos)
        for i in reversed(range(nl)):
            n1_w, l1_w, n2_w = params[i * N_LAYER + 8], params[i * N_LAYER + 10], params[i * N_LAYER + 14]
            x, q, v4, loc6, attn5, a, z1, m1, r1, y1, h, z2, m2, r2, fused_ref, hbits = saved[i]
            (g_sow, g_sob, g_aww, g_awb, g_vpw, g_vpb, g_opw, g_opb, g_n1w, g_n1b, g_l1w, g_l1b, g_l2w, g_l2b, g_n2w,
             g_n2b) = [G(i, j) for j in range(N_LAYER)]
            # ---- FFN + norm2
            dz2, _, dz2_am = rw.add_ln_bwd(z2, m2, r2, n2_w, dy=dy, dy2=dy2, dypos_c=dyq, dgamma=g_n2w, dbeta=g_n2b, dbias=g_l2b,
                                           dpos_acc=d_pos if dyq is not None else None, pos_div=1, amax=True)
            x_am, q_am, a_am, y1_am, h_am = saved_am[i]
            wgrad(dz2, h, g_l2w, None, dz2_am, h_am)
            if hbits is not None:
                dh = gemm_tn_h2(dz2, l2_t[i], None, mode=2, bits=hbits, colsum=g_l1b, a_amax=dz2_am, b_amax=l2_tam[i], c_amax=dh_am_all[i])
                dh_am = dh_am_all[i]
            else:
                dh = rw.relu_bwd_colsum(gemm_tn_h2(dz2, l2_t[i], a_amax=dz2_am, b_amax=l2_tam[i]), h, g_l1b)
                dh_am = row_amax(dh)
            wgrad(dh, y1, g_l1w, None, dh_am, y1_am)
            dy1 = gemm_tn_h2(dh, l1_t[i], a_amax=dh_am, b_amax=l1_tam[i])
            # ---- deformable attention + norm1 (with queued weight gradients dz2 stays alive until the grouped launch: dz1 gets its own buffer)
            dz1, _, dz1_am = rw.add_ln_bwd(z1, m1, r1, n1_w, dy=dz2, dy2=dy1, dgamma=g_n1w, dbeta=g_n1b, dbias=g_opb,
                                           out=None if queue is not None else dz2, amax=True)
            wgrad(dz1, a, g_opw, None, dz1_am, a_am)
            da = gemm_tn_h2(dz1, op_t[i], a_amax=dz1_am, b_amax=op_tam[i]).view(B, S, C)
            if fused_ref is not None:
                # (loc6, attn5) hold the raw projection output and the softmax statistics: the kernel writes d(projection output) itself
                gv, d_oa, d_oa_am = _timed("bwd", msda_fused_backward, v4, spec.shapes, spec.lsi, loc6, ref, attn5, a, da)
            else:
                gv, gloc, gattn = _timed("bwd", MSDA.ms_deform_attn_backward, v4, spec.shapes, spec.lsi, loc6, attn5, da, spec.im2col_step)
                d_oa = torch.empty((T, n_oa), dtype=torch.float32, device=dev)
            if fused_ref is not None:
                pass
            elif prep_amax_supported(M, L, P):
                d_oa_am = torch.empty(T, dtype=torch.float32, device=dev)
                msda_prep_bwd(gloc, gattn, attn5, spec.shapes, T, M, L, P, out=d_oa, amax=d_oa_am)
            else:
                msda_prep_bwd(gloc, gattn, attn5, spec.shapes, T, M, L, P, out=d_oa)
                d_oa_am = row_amax(d_oa)
            g_oaw, g_oab = OA(i)
            wgrad(d_oa, q, g_oaw, g_oab, d_oa_am, q_am)                   # both weight gradients in one split-K GEMM
            g_sow, g_aww, g_sob, g_awb = g_oaw[:n_off], g_oaw[n_off:], g_oab[:n_off], g_oab[n_off:]
            dq = gemm_tn_h2(d_oa, oa_t[i], a_amax=d_oa_am, b_amax=oa_tam[i])
            gv2 = gv.view(T, C)
            gv_am = row_amax(gv2)
            wgrad(gv2, x, g_vpw, g_vpb, gv_am, x_am)
            dxv = gemm_tn_h2(gv2, vp_t[i], a_amax=gv_am, b_amax=vp_tam[i])
            grads[i * N_LAYER:(i + 1) * N_LAYER] = [g_sow, g_sob, g_aww, g_awb, g_vpw, g_vpb, g_opw, g_opb, g_n1w, g_n1b,
                                                     g_l1w, g_l1b, g_l2w, g_l2b, g_n2w, g_n2b]
            dy, dy2, dyq = dz1, dxv, dq
        if queue is not None:
            queue.flush()
        return dy, dy2, dyq, d_pos, grads

    @staticmethod
    def backward(ctx, d_out):
        if getattr(ctx, "h2", False) and X3_PROJ and USE_X3:
            return EncoderCore._backward_h2(ctx, d_out)
        spec, params = ctx.spec, ctx.params
        B, S, C, nl = ctx.dims
        M, L, P = spec.M, spec.L, spec.P
        T = B * S
        dev = d_out.device
        # frozen encoder (FREEZE_KEYS contains "encoder", the shipped scripts' setting: base_trainer.py:97-100): none of its
        # parameters requires a gradient, so the five weight-gradient GEMMs per layer (a third of the encoder's backward
        # FLOPs) are skipped; the input gradient still flows (input_proj / level_embed in front of it stay trainable)
        need_w = any(ctx.needs_input_grad[3:])
        from . import gemm as _gemm
        h2 = getattr(ctx, "h2", False)
        wh2 = h2 and H2_WGRAD
        queue = WgradQueue(h2=wh2) if (need_w and GROUP_WGRADS and _gemm.WGRAD_X3 and d_out.is_cuda) else None
        if not need_w:
            wgrad = lambda *a, **k: None
        elif queue is not None:
            wgrad = (lambda dy_, x_, dw_, db_=None, ya=None, xa=None: queue.add(dy_, x_, dw_, db_, ya if wh2 else None, xa if wh2 else None))
        else:
            wgrad = (lambda dy_, x_, dw_, db_=None, ya=None, xa=None: gemm_wgrad_acc(dy_, x_, dw_, db_, h2=wh2, y_amax=ya if wh2 else None,
                                                                                     x_amax=xa if wh2 else None))
        # every parameter gradient of the encoder lives in ONE zero-filled fp32 buffer (a single memset): the LayerNorm /
        # ReLU kernels and the split-K weight-gradient GEMMs all accumulate (+=) into their slices
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        n_oa = params[0].shape[0] + params[2].shape[0]                  # stacked sampling_offsets + attention_weights rows
        oa_base, oa_per = total, n_oa * C + (n_oa + 3) // 4 * 4
        total += nl * oa_per
        buf = torch.zeros(total, dtype=torch.float32, device=dev)

        def OA(i):
            o = oa_base + i * oa_per
            return buf[o:o + n_oa * C].view(n_oa, C), buf[o + n_oa * C:o + n_oa * C + n_oa]

        def G(i, j):
            p = params[i * N_LAYER + j]
            o = offs[i * N_LAYER + j]
            return buf[o:o + p.numel()].view(p.shape)
        d_pos = torch.zeros((T, C), dtype=torch.float32, device=dev)
        grads = [None] * (nl * N_LAYER)
        # the input-gradient GEMMs want W^T row-major ([in, out] -> the "B[N, K]" operand with N = in): all layers' transposes in one
        # stack + one copy per weight shape instead of one small launch per layer and weight
        def T_all(j):
            ws = [params[i * N_LAYER + j] for i in range(nl)]
            # the layers' parameters usually sit at one spacing in a flat buffer (engine/flat_params.py, last layer first): the stack is
            # then a strided VIEW and the transposed copy the only launch (torch.stack was a second one per weight kind, 42 us per step)
            d = (ws[1].data_ptr() - ws[0].data_ptr()) if nl > 1 else 0
            es = ws[0].element_size()
            if nl > 1 and d != 0 and d % es == 0 and all(w.is_contiguous() and w.shape == ws[0].shape for w in ws) \
                    and all(ws[i].data_ptr() - ws[0].data_ptr() == i * d for i in range(nl)) \
                    and all(w.untyped_storage().data_ptr() == ws[0].untyped_storage().data_ptr() for w in ws):
                N_, K_ = ws[0].shape
                base = ws[0] if d > 0 else ws[-1]                 # lowest address first; _Stack maps the layer index back
                return _Stack(torch.as_strided(base, (nl, N_, K_), (abs(d) // es, K_, 1)).transpose(1, 2).contiguous(), d < 0)
            return _Stack(torch.stack(ws).transpose(1, 2).contiguous(), False)
        l1_t, l2_t = T_all(10), T_all(12)
        op_t, vp_t = (T_all(6), T_all(4)) if X3_PROJ and USE_X3 else (None, None)
        oa_t = _Stack(torch.stack([sv[14] for sv in ctx.saved]).transpose(1, 2).contiguous(), False) if op_t is not None else None
        if h2:
            # row maxima of the transposed weights (the B operands of the input-gradient GEMMs), one launch per stack
            t_am = lambda w: _Stack(row_amax(w.t.view(-1, w.t.shape[2])).view(nl, w.t.shape[1]), w.rev)
            l1_tam, l2_tam, op_tam, vp_tam, oa_tam = t_am(l1_t), t_am(l2_t), t_am(op_t), t_am(vp_t), t_am(oa_t)
            dh_am_all = torch.zeros((nl, T), dtype=torch.float32, device=dev)
        dy = d_out.reshape(T, C)
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dy2 = dyq = None                                       # further fp32 terms of d(src_l): via value_proj, via (src + pos)
        for i in reversed(range(nl)):
            (so_w, so_b, aw_w, aw_b, vp_w, vp_b, op_w, op_b, n1_w, n1_b, l1_w, l1_b, l2_w, l2_b, n2_w, n2_b) = params[i * N_LAYER:(i + 1) * N_LAYER]
            x, q, v4, loc6, attn5, a, z1, m1, r1, y1, h, z2, m2, r2, w_oa, hbits = ctx.saved[i]
            (g_sow, g_sob, g_aww, g_awb, g_vpw, g_vpb, g_opw, g_opb, g_n1w, g_n1b, g_l1w, g_l1b, g_l2w, g_l2b, g_n2w,
             g_n2b) = [G(i, j) for j in range(N_LAYER)]
            # ---- FFN + norm2
            dz2, _, dz2_am = rw.add_ln_bwd(z2, m2, r2, n2_w, dy=dy, dy2=dy2, dypos_c=dyq, dgamma=g_n2w, dbeta=g_n2b, dbias=g_l2b,
                                           dpos_acc=d_pos if dyq is not None else None, pos_div=1, amax=True)
            if h2:
                x_am, q_am, a_am, y1_am, h_am = ctx.saved_am[i]
                wgrad(dz2, h, g_l2w, None, dz2_am, h_am)
                if hbits is not None:
                    dh = gemm_tn_h2(dz2, l2_t[i], None, mode=2, bits=hbits, colsum=g_l1b, a_amax=dz2_am, b_amax=l2_tam[i], c_amax=dh_am_all[i])
                    dh_am = dh_am_all[i]
                else:
                    dh = rw.relu_bwd_colsum(gemm_tn_h2(dz2, l2_t[i], a_amax=dz2_am, b_amax=l2_tam[i]), h, g_l1b)
                    dh_am = row_amax(dh)
                wgrad(dh, y1, g_l1w, None, dh_am, y1_am)
                dy1 = gemm_tn_h2(dh, l1_t[i], a_amax=dh_am, b_amax=l1_tam[i])
                del dh
                dz1, _, dz1_am = rw.add_ln_bwd(z1, m1, r1, n1_w, dy=dz2, dy2=dy1, dgamma=g_n1w, dbeta=g_n1b, dbias=g_opb,
                                               out=None if queue is not None else dz2, amax=True)
                wgrad(dz1, a, g_opw, None, dz1_am, a_am)
                da = gemm_tn_h2(dz1, op_t[i], a_amax=dz1_am, b_amax=op_tam[i]).view(B, S, C)
                gv, gloc, gattn = _timed("bwd", MSDA.ms_deform_attn_backward, v4, spec.shapes, spec.lsi, loc6, attn5, da, spec.im2col_step)
                d_oa = torch.empty((T, w_oa.shape[0]), dtype=torch.float32, device=dev)
                if prep_amax_supported(M, L, P):
                    d_oa_am = torch.empty(T, dtype=torch.float32, device=dev)
                    msda_prep_bwd(gloc, gattn, attn5, spec.shapes, T, M, L, P, out=d_oa, amax=d_oa_am)
                else:
                    msda_prep_bwd(gloc, gattn, attn5, spec.shapes, T, M, L, P, out=d_oa)
                    d_oa_am = row_amax(d_oa)
                g_oaw, g_oab = OA(i)
                wgrad(d_oa, q, g_oaw, g_oab, d_oa_am, q_am)                   # both weight gradients in one split-K GEMM
                n_off = so_w.shape[0]
                g_sow, g_aww, g_sob, g_awb = g_oaw[:n_off], g_oaw[n_off:], g_oab[:n_off], g_oab[n_off:]
                dq = gemm_tn_h2(d_oa, oa_t[i], a_amax=d_oa_am, b_amax=oa_tam[i])
                gv2 = gv.view(T, C)
                gv_am = row_amax(gv2)
                wgrad(gv2, x, g_vpw, g_vpb, gv_am, x_am)
                dxv = gemm_tn_h2(gv2, vp_t[i], a_amax=gv_am, b_amax=vp_tam[i])
                grads[i * N_LAYER:(i + 1) * N_LAYER] = [g_sow, g_sob, g_aww, g_awb, g_vpw, g_vpb, g_opw, g_opb, g_n1w, g_n1b,
                                                         g_l1w, g_l1b, g_l2w, g_l2b, g_n2w, g_n2b]
                dy, dy2, dyq = dz1, dxv, dq
                continue
            wgrad(dz2, h, g_l2w)
            pre = hbits is not None and USE_X3 and PRESPLIT and pre_supported(T, l2_w.shape[1], l2_w.shape[0]) and pre_supported(T, l1_w.shape[1], l1_w.shape[0])
            if pre:
                dh = gemm_tn_x3_pre(dz2, split3(l2_w, transpose=True), mode=2, bits=hbits, colsum=g_l1b)
            elif hbits is not None:
                dh = gemm_tn_x3_relumask(dz2, l2_t[i], hbits, g_l1b)                # ReLU backward + bias gradient in the epilogue
            else:
                dh = rw.relu_bwd_colsum(_ffn_gemm(dz2, l2_t[i]), h, g_l1b)
            wgrad(dh, y1, g_l1w)
            dy1 = gemm_tn_x3_pre(dh, split3(l1_w, transpose=True)) if pre else _ffn_gemm(dh, l1_t[i])
            del dh
            # ---- deformable attention + norm1
            # (with queued weight gradients dz2 stays alive until the grouped launch: dz1 gets its own buffer)
            dz1, _ = rw.add_ln_bwd(z1, m1, r1, n1_w, dy=dz2, dy2=dy1, dgamma=g_n1w, dbeta=g_n1b, dbias=g_opb, out=None if queue is not None else dz2)
            wgrad(dz1, a, g_opw)
            da = (_proj(dz1, op_t[i]) if op_t is not None else torch.mm(dz1, op_w)).view(B, S, C)
            gv, gloc, gattn = _timed("bwd", MSDA.ms_deform_attn_backward, v4, spec.shapes, spec.lsi, loc6, attn5, da, spec.im2col_step)
            d_oa = torch.empty((T, w_oa.shape[0]), dtype=torch.float32, device=dev)
            msda_prep_bwd(gloc, gattn, attn5, spec.shapes, T, M, L, P, out=d_oa)
            g_oaw, g_oab = OA(i)
            wgrad(d_oa, q, g_oaw, g_oab)                                  # both weight gradients in one split-K GEMM
            n_off = so_w.shape[0]
            g_sow, g_aww, g_sob, g_awb = g_oaw[:n_off], g_oaw[n_off:], g_oab[:n_off], g_oab[n_off:]
            dq = _proj(d_oa, oa_t[i]) if op_t is not None else torch.mm(d_oa, w_oa)
            gv2 = gv.view(T, C)
            wgrad(gv2, x, g_vpw, g_vpb)
            dxv = _proj(gv2, vp_t[i]) if vp_t is not None else torch.mm(gv2, vp_w)
            grads[i * N_LAYER:(i + 1) * N_LAYER] = [g_sow, g_sob, g_aww, g_awb, g_vpw, g_vpb, g_opw, g_opb, g_n1w, g_n1b,
                                                     g_l1w, g_l1b, g_l2w, g_l2b, g_n2w, g_n2b]
            dy, dy2, dyq = dz1, dxv, dq
        if queue is not None:
            queue.flush()
        d_pos += dyq
        d_src = dy + dy2
        d_src += dyq
        if not need_w:
            grads = [None] * len(grads)
        return (None, d_src.view(B, S, C), d_pos.view(B, S, C), *grads)
