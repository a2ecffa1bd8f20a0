"""One Swin stage (all SwinTransformerBlocks of a BasicLayer, reference modeling/backbone/swin.py:239-299, 446-452) as ONE
autograd node with a hand-written backward, in the style of functions/decoder_core.py / encoder_core.py.

Per block the forward is 8 launches and the backward 17: the glue between the GEMMs — residual adds, DropPath scales,
both LayerNorms, F.pad + torch.roll + window_partition and their inverses, the bf16 casts — is the row kernel of
include/pd_swin.h (once before each attention, once before each MLP), the attention is include/pd_window_attention.h,
the Linears are pd_igemm_bf16 (include/pd_igemm.h) — or, with MODEL.SWIN.FP8_GEMM on a stage whose width allows it, pd_mx8_gemm
(include/pd_mx8.h: MX-fp8 operands, BASELINE config 5).  The MLP output of block k is never added on its own: it rides into block
k+1's first row kernel as the pending residual.  Same arithmetic as the module-by-module path under bf16 autocast (fp32
residual stream, fp32 LayerNorm statistics, bf16 GEMM operands, exact-erf GELU)."""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from ... import cmdbuf
from ... import lib as _lib
from ...functions import conv_bf16, igemm, mx8, smallgemm
from ...functions import rowwise as _rw
from ...functions import swin_rows as rows
from ...functions import window_attention as wattn

TAIL_FUSED = __import__("os").environ.get("PD_SWIN_TAIL_FUSED", "1") != "0"    # stage end + output norm as pd_swin_tail_ln_* (0: ATen chain + the row LayerNorm)
NREP = 8                 # copies of a stage's LayerNorm column-sum accumulators (see _bwd_blocks)
N_BLOCK = 13            # norm1.w, norm1.b, qkv.w, qkv.b, table, proj.w, proj.b, norm2.w, norm2.b, fc1.w, fc1.b, fc2.w, fc2.b
_MAPS = {}


def window_maps(H, W, shift, device):
    """int32 row maps of the padded, cyclically shifted 12 x 12 window partition: token -> window-major slot, the padded
    slots (zero rows), slots per image, windows per image"""
    from .swin import window_gather_index
    key = (H, W, shift, str(device))
    if key not in _MAPS:
        win, inv, n_win, _ = window_gather_index(H, W, wattn.WINDOW, shift, device)
        zero = (win == H * W).nonzero().flatten().to(torch.int32)
        _MAPS[key] = (inv.to(torch.int32).contiguous(), zero.contiguous() if zero.numel() else None, int(win.numel()), n_win)
    return _MAPS[key]


def _bf(t):
    return t if t.dtype == torch.bfloat16 else t.to(torch.bfloat16)


OWN_GEMM = __import__("os").environ.get("PD_SWIN_OWN_GEMM", "1") != "0"   # qkv / proj / fc1 (+ GELU) / fc2 and their input gradients (+ GELU') on
                     # pd_igemm_bf16 (include/pd_igemm.h) instead of the library's addmm / mm + separate GELU kernels


def _own(x, *ws):
    return (OWN_GEMM and x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous()
            and all(w.dtype == torch.bfloat16 and w.is_contiguous() and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0 for w in ws))


def _own_w(*ws):
    return OWN_GEMM and all(w.is_cuda and w.dtype == torch.bfloat16 and w.is_contiguous() and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0 for w in ws)


def _lin(x, w, b):
    """x [M, K] bf16 @ w [N, K]^T + b"""
    if _own(x, w):
        return igemm.linear(x, w, b)
    return torch.addmm(_bf(b), x, _bf(w).t())


TR_WGRAD = __import__("os").environ.get("PD_SWIN_TR_WGRAD", "0") != "0"     # True: weight gradients through the transpose-read kernel of csrc/conv_bf16.hip (a Linear over the stage's tokens = a
                     # 1 x 1 convolution over pixels), queued for the step's grouped launch.  Measured on config 3 (Swin-B, 2 x 1024^2):
                     # 51.8 ms per step against 49.8 with the split-rows / library kernels below — that kernel is built for the
                     # HBM-bound filter gradients of R50 (N K / (N + K) <= 64 flop per byte); 18 of Swin-B's 24 blocks have
                     # 1536 x 512 .. 2048 x 512 outputs over 8 192 tokens, which are compute-bound.


def _wgrad(dy, x, w, b, big=None, pend=None):
    """weight + bias gradient of a Linear over all tokens of the stage, in the parameters' dtypes.  bf16 parameters (the training
    configuration): the weight gradient is queued in `big` for conv_bf16's grouped transpose-read launch, the bias gradient is a
    column sum.  Otherwise pd_wgrad_bf16 (include/pd_igemm.h) for bf16 operands — 36-46 us at the 10 368-token stage of Swin-B against
    72-79 for the library and 85-115 for the split-rows kernel, profiles/r04_swin_wgrad_sweep.txt — except few-row / large-result
    shapes (igemm.wgrad_prefers_library).  Without it: small [N, K] outputs (few tiles, every one walking 10^4..10^5 rows) go to the split-rows matrix-core
    kernel of include/pd_smallgemm.h — measured 2.5-3x the library at K <= 256, break-even at N*K ~ 1 M
    (tools/bench_wgrad_split.py); larger outputs have enough tiles for the library GEMM."""
    if (big is not None and w.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dy.is_contiguous()
            and x.is_contiguous() and dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0):
        dw = torch.empty((dy.shape[1], x.shape[1]), dtype=torch.bfloat16, device=dy.device)
        big.append(conv_bf16.rows_entry(dy, x, dw))
        db = torch.zeros(dy.shape[1], dtype=torch.float32, device=dy.device)
        _rw.colsum_acc(dy, db)
        return dw, db                                        # fp32: cast with the stage's other bias gradients (_cast_bias_grads)
    if OWN_GEMM and igemm.wgrad_supported(dy, x) and (cmdbuf.active() is not None or not igemm.wgrad_prefers_library(dy.shape[0], dy.shape[1], x.shape[1])):
        db = torch.zeros(dy.shape[1], dtype=torch.float32, device=dy.device)
        if pend is not None:                                     # queued: the block's four weight gradients leave by ONE call (pd_wgrad_bf16_seq)
            dw = torch.empty((dy.shape[1], x.shape[1]), dtype=w.dtype if w.dtype == torch.float32 else torch.bfloat16, device=dy.device)
            pend.append((dy, x, dw, db))
            return dw, db
        dw = igemm.wgrad(dy, x, None, db, out_dtype=w.dtype if w.dtype == torch.float32 else torch.bfloat16)   # dW and the column sums of dY in one pass over dY
    elif w.shape[0] * w.shape[1] <= 1_100_000:
        dw, db = smallgemm.wgrad_split(dy, x, True)
    else:
        dw = torch.mm(dy.t(), x)
        if dy.is_cuda and dy.is_contiguous() and dy.shape[1] % 128 == 0 and dy.dtype in (torch.bfloat16, torch.float32):
            db = torch.zeros(dy.shape[1], dtype=torch.float32, device=dy.device)
            _rw.colsum_acc(dy, db)                              # (ATen's column reduction takes 2-3 x as long at [2 048, 4 096])
        else:
            db = dy.sum(0, dtype=torch.float32)
    return (dw if dw.dtype == w.dtype else dw.to(w.dtype)), (db if db.dtype == b.dtype or db.dtype == torch.float32 else db.to(b.dtype))


def _cast_bias_grads(grads, params, slots):
    """the fp32 bias gradients of a stage (4 per block) in their parameters' dtype: ONE concatenation + ONE cast whose pieces are handed
    out as views, instead of a cast launch per bias (96 per step at Swin-B: 0.4 ms of 5 us kernels and their host time)"""
    todo = [i for i in slots if grads[i] is not None and grads[i].dtype != params[i].dtype]
    if not todo:
        return
    by_dtype = {}
    for i in todo:
        by_dtype.setdefault(params[i].dtype, []).append(i)
    for dt, idx in by_dtype.items():
        if len(idx) == 1:
            grads[idx[0]] = grads[idx[0]].to(dt)
            continue
        flat = torch.cat([grads[i].reshape(-1) for i in idx]).to(dt)
        o = 0
        for i in idx:
            n = grads[i].numel()
            grads[i] = flat[o:o + n].view(grads[i].shape)
            o += n


def _drop_backward_of(key, rec, cache):
    for k in [k for k in cache if k[0] == "bwd" and k[1] == id(rec)]:      # a backward recording pins its forward recording's arenas
        cache.pop(k)


_RECS = cmdbuf.LRU(32, _drop_backward_of)             # recorded regions (cmdbuf.Recording) per stage geometry, least recently used out first


def _rec_put(key, rec):
    _RECS.put(key, rec)


def _stage_consts(spec, device):
    """the cached index tensors the stage's kernels read (row maps, padded rows, shift regions): built (ATen) before a recorded region,
    declared to it as address-stable slots"""
    out = []
    for shift in sorted(set(spec["shifts"])):
        ymap, zero, _, _ = window_maps(spec["H"], spec["W"], shift, device)
        out += [ymap] + ([zero] if zero is not None else [])
        if shift > 0:
            out += list(wattn.shifted_window_regions(spec["H"], spec["W"], shift, device))
    return out


class SwinStage(Function):
    """x fp32 [B, L, C] -> fp32 [B, L, C].  spec = dict(H, W, heads, shifts [depth], scale, eps, dp = None | fp32
    [depth, 2, B] DropPath scales (keep mask / keep_prob)).

    With every Linear on pd_igemm_bf16 / pd_wgrad_bf16 the block loops are pd_* launches and allocations only, and run as RECORDED regions
    (cmdbuf.py): 8 x depth launches forward, 21 x depth backward replayed by one C call each instead of ~30 Python calls per block —
    the host issued a Swin-B step in 35.7 ms against 35 ms of GPU work before (tools/bench_config3.py)."""

    @staticmethod
    def forward(ctx, x, spec, nw, nb, *params):
        """nw / nb: weight and bias of the stage's OUTPUT norm (reference swin.py:675-680) or None; with them the node returns (stage output, its norm) and
        the joining of the last block's MLP output with the stream runs inside the norm's kernel (pd_swin_tail_ln_*), otherwise (stage output, None)"""
        if not x.is_cuda:
            raise RuntimeError("the fused Swin stage runs on the GPU only (no CPU fallback in partdistillation_amd)")
        ctx.set_materialize_grads(False)
        B, L, C = x.shape
        dp = spec["dp"]
        depth = len(params) // N_BLOCK
        x2 = x.contiguous().view(B * L, C)
        weights = [params[k * N_BLOCK + j] for k in range(depth) for j in (2, 5, 9, 11)]
        use_rec = (cmdbuf.usable() and OWN_GEMM and x2.dtype == torch.float32 and any(ctx.needs_input_grad)
                   and all(w.dtype == torch.bfloat16 and w.is_contiguous() and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0 for w in weights)
                   and all(p.is_contiguous() for p in params))
        if not use_rec:
            ctx.rec = None
            r, cur, saved = SwinStage._fwd_blocks(x2, spec, params, dp)
        else:
            consts = _stage_consts(spec, x.device)
            head = [x2] + ([dp] if dp is not None else [])
            slots = head + list(params) + consts
            key = ("fwd", B, L, C, spec["H"], spec["W"], spec["heads"], tuple(spec["shifts"]), dp is not None, params[0].data_ptr(), str(x.device),
                   _lib.current_stream(), bool(spec.get("mx8")))
            rec = _RECS.get(key)
            if rec is None or not rec.matches(slots):
                rec = cmdbuf.Recording(slots, "swin stage forward", pinned=range(len(head), len(slots)))
                with rec:
                    outs = SwinStage._fwd_blocks(x2, spec, params, dp)
                outs = rec.finish(outs)
                _rec_put(key, rec)
            else:
                outs = rec.replay(slots)
            r, cur, saved = outs
            ctx.rec, ctx.rec_gen = rec, rec.generation
        # ---- eager epilogue: the last block's MLP output joins the stream — inside the output norm's kernel when the stage has one
        rscale = dp[depth - 1, 1] if dp is not None else None
        ctx.spec, ctx.depth, ctx.shape, ctx.saved = spec, depth, (B, L, C), saved
        ctx.tail = False
        if nw is not None and TAIL_FUSED and rows.tail_ln_supported(C, nw, nb) and r.dtype == torch.bfloat16 and cur.dtype == torch.float32:
            rs = rscale.contiguous() if rscale is not None else None
            out, y, st = rows.tail_ln_fwd(cur.view(B * L, C), r.view(B * L, C), rs, L, nw, nb, spec["out_eps"])
            out = out.view(B, L, C)
            ctx.tail = True
            ctx.save_for_backward(nw, out, st, rs, *params)               # (`out` is the node's own output: saved the way autograd wants outputs saved)
            return out, y.view(B, L, C)
        out = r.view(B, L, C).float()
        if rscale is not None:
            out = out * rscale.view(B, 1, 1)
        out = out.add_(cur.view(B, L, C))
        ctx.save_for_backward(nw if nw is not None else params[0], *params)
        return out, None

    @staticmethod
    def _fwd_blocks(x2, spec, params, dp):
        """-> (MLP output of the last block bf16, residual stream before it fp32, saved activations); inside a recording: pd_* launches
        and allocations only"""
        H, W = spec["H"], spec["W"]
        C = x2.shape[1]
        depth = len(params) // N_BLOCK
        B = x2.shape[0] // (H * W)
        L = H * W
        cur = x2
        if cmdbuf.active() is not None:                        # saved for the backward region (block 0's stream IS the input): an arena copy
            cur = _rw.copy_d2d(torch.empty_like(x2), x2)
        r, rscale = None, None
        saved = []
        mx = bool(spec.get("mx8")) and _own_w(*[params[k * N_BLOCK + j] for k in range(depth) for j in (2, 5, 9, 11)])
        if mx:                                                  # the stage's 4 x depth weights as MX e4m3 along their input axis: one launch
            wq = mx8.quantize_grouped([params[k * N_BLOCK + j] for k in range(depth) for j in (2, 5, 9, 11)], mx8.E4M3)
        for k in range(depth):
            n1w, n1b, qw, qb, table, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b = params[k * N_BLOCK:(k + 1) * N_BLOCK]
            shift = spec["shifts"][k]
            ymap, zero, S, nW = window_maps(H, W, shift, x2.device)
            regions = wattn.shifted_window_regions(H, W, shift, x2.device) if shift > 0 else None
            table = table if table.is_contiguous() else table.contiguous()
            if mx:                                              # the LayerNorm rows leave as MX e4m3 operands next to their bf16 copy
                s1, y1, st1, y1q = rows.ln_fwd(cur, r, None, L, rscale, n1w, n1b, spec["eps"], ymap, S, zero, B, L, mx=mx8.E4M3)
                qkv = mx8.linear(y1q, wq[4 * k], qb)
            else:
                s1, y1, st1 = rows.ln_fwd(cur, r, None, L, rscale, n1w, n1b, spec["eps"], ymap, S, zero, B, L)
                qkv = _lin(y1, qw, qb)
            if mx:                                              # a (token, head) piece of the attention output is one MX block: it leaves quantised
                ao, lse, aoq = wattn.fwd_raw(qkv.view(B * nW, wattn.TOKENS, 3 * C), table, regions, spec["scale"], nW, mx=mx8.E4M3)
                po = mx8.linear(aoq, wq[4 * k + 1], pb)
            else:
                ao, lse = wattn.fwd_raw(qkv.view(B * nW, wattn.TOKENS, 3 * C), table, regions, spec["scale"], nW)
                po = _lin(ao.view(-1, C), pw, pb)
            sc1 = dp[k, 0] if dp is not None else None
            if mx:                                              # fc1's epilogue hands fc2 its operand: GELU(h) again as MX e4m3 along the 4 C axis
                s2, y2, st2, y2q = rows.ln_fwd(s1, po, ymap, S, sc1, n2w, n2b, spec["eps"], None, L, None, B, L, mx=mx8.E4M3)
                a, h, aq = mx8.linear(y2q, wq[4 * k + 2], f1b, act=mx8.ACT_GELU, want_pre=True, out_mx=mx8.E4M3)
                f = mx8.linear(aq, wq[4 * k + 3], f2b)
            else:
                s2, y2, st2 = rows.ln_fwd(s1, po, ymap, S, sc1, n2w, n2b, spec["eps"], None, L, None, B, L)
                if _own(y2, f1w):
                    a, h = igemm.linear(y2, f1w, f1b, act=igemm.ACT_GELU, want_pre=True)     # bias + exact-erf GELU in the GEMM epilogue; h kept for GELU'
                else:
                    h = torch.addmm(_bf(f1b), y2, _bf(f1w).t())
                    a = F.gelu(h)
                f = _lin(a, f2w, f2b)
            saved += [s1, y1, st1, qkv, ao, lse, s2, y2, st2, h, a]
            cur, r, rscale = s2, f, (dp[k, 1] if dp is not None else None)
        return r, cur, saved

    @staticmethod
    def backward(ctx, dout, dy=None):
        spec, depth, (B, L, C) = ctx.spec, ctx.depth, ctx.shape
        dp = spec["dp"]
        sv, saved = ctx.saved_tensors, ctx.saved
        nw, params = sv[0], (sv[4:] if ctx.tail else sv[1:])
        if dout is None and (dy is None or not ctx.tail):
            return (None,) * (4 + len(params))
        g_nw = g_nb = None
        if ctx.tail and dy is not None:
            # ---- eager prologue, one kernel: stream gradient = (gradient of the stage output from the next stage) + LayerNorm'(gradient of its norm), and its
            # DropPath-scaled 16-bit copy (the gradient of the last block's MLP output)
            s_out, st, rs = sv[1], sv[2], sv[3]
            dyc = dy.contiguous().view(B * L, C)
            dyc = dyc if dyc.dtype == torch.float32 else dyc.float()
            dsum = None
            if dout is not None:
                dsum = dout.contiguous().view(B * L, C)
                dsum = dsum if dsum.dtype == torch.float32 else dsum.float()
            dsup, df, g_nw, g_nb = rows.tail_ln_bwd(dyc, dsum, s_out.view(B * L, C), st, nw, rs, L)
        else:
            # ---- eager prologue (ATen): the gradient of the last block's MLP output in 16 bits
            dsup = dout.contiguous().view(B * L, C)
            dsup = dsup if dsup.dtype == torch.float32 else dsup.float()
            last = dp[depth - 1, 1] if dp is not None else None
            df = (dsup.view(B, L, C) * last.view(B, 1, 1) if last is not None else dsup).to(torch.bfloat16).view(B * L, C)
        rec_f = getattr(ctx, "rec", None)
        if rec_f is None:
            dx, grads = SwinStage._bwd_blocks(spec, params, saved, dsup, df, dp, (B, L, C))
        else:
            if rec_f.generation != ctx.rec_gen:
                raise RuntimeError("the fused Swin stage ran another forward before this backward: the recorded region's activation arena was "
                                   "overwritten (set PD_CMDBUF=0 for graphs that keep several forward passes of one stage alive)")
            consts = _stage_consts(spec, dsup.device)
            head = [dsup, df] + ([dp] if dp is not None else [])
            slots = head + list(params) + consts
            key = ("bwd", id(rec_f))
            rec = _RECS.get(key)
            if rec is None or not rec.matches(slots):
                rec = cmdbuf.Recording(slots, "swin stage backward", stable=[rec_f], pinned=range(len(head), len(slots)))
                with rec:
                    dx, grads = SwinStage._bwd_blocks(spec, params, saved, dsup, df, dp, (B, L, C))
                    outs = (dx, cmdbuf.Fresh(grads))
                outs = rec.finish(outs)
                _rec_put(key, rec)
            else:
                cmdbuf.unalias_grads(params, rec.owns)
                outs = rec.replay(slots)
            dx, grads = outs
        grads = list(grads)
        _cast_bias_grads(grads, params, [k * N_BLOCK + j for k in range(depth) for j in (3, 6, 10, 12)])
        return (dx.view(B, L, C), None, g_nw, g_nb, *grads)

    @staticmethod
    def _bwd_blocks(spec, params, saved, dsup, df, dp, shape):
        """-> (gradient of the stage input fp32 [B * L, C], the 13 x depth parameter gradients); inside a recording: pd_* launches and
        allocations only"""
        B, L, C = shape
        H, W = spec["H"], spec["W"]
        depth = len(params) // N_BLOCK
        dev = dsup.device
        if cmdbuf.active() is not None:                        # df goes into problem structs (host memory the replay re-reads): an arena copy
            df = _rw.copy_d2d(torch.empty_like(df), df)
        # dgamma1, dbeta1, dgamma2, dbeta2 of every block, in NREP copies the LayerNorm' workgroups spread their column-sum atomics over
        # (pd_swin.h: ~500 workgroups adding into the same 2 C addresses serialise: 19 of 47 us at [14 112, 768]); summed once below
        norm_r = torch.zeros((NREP, depth, 4, C), dtype=torch.float32, device=dev)
        norm_g, rep = norm_r[0], dict(n_rep=NREP, rep_stride=depth * 4 * C)
        grads = [None] * (depth * N_BLOCK)
        big = [] if TR_WGRAD and cmdbuf.active() is None else None
        tab0 = params[4]
        tables_g = torch.zeros((depth,) + tuple(tab0.shape), dtype=tab0.dtype, device=dev) \
            if all(params[k * N_BLOCK + 4].shape == tab0.shape and params[k * N_BLOCK + 4].dtype == tab0.dtype for k in range(depth)) else None

        own = _own(df, *[params[k * N_BLOCK + j] for k in range(depth) for j in (2, 5, 9, 11)])
        if own:                                                  # W^T of the stage's 4 x depth Linears: one grouped launch
            wts = igemm.transposed([params[k * N_BLOCK + j] for k in range(depth) for j in (2, 5, 9, 11)])
        mx = bool(spec.get("mx8")) and own
        if mx:                                                   # ... and those again as MX e4m3 along THEIR contraction axis (the output features)
            wtq = mx8.quantize_grouped(wts, mx8.E4M3)
            gf, dfq = mx8.GRAD_FORMAT, None
        for k in reversed(range(depth)):
            n1w, n1b, qw, qb, table, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b = params[k * N_BLOCK:(k + 1) * N_BLOCK]
            s1, y1, st1, qkv, ao, lse, s2, y2, st2, h, a = saved[k * 11:(k + 1) * 11]
            if own:
                qw_t, pw_t, f1w_t, f2w_t = wts[4 * k:4 * k + 4]
            shift = spec["shifts"][k]
            ymap, zero, S, nW = window_maps(H, W, shift, dev)
            regions = wattn.shifted_window_regions(H, W, shift, dev) if shift > 0 else None
            g = grads[k * N_BLOCK:(k + 1) * N_BLOCK]
            table = table if table.is_contiguous() else table.contiguous()
            # MLP
            pend = [] if own else None
            g[11], g[12] = _wgrad(df, a, f2w, f2b, big, pend)
            if mx:                                               # gradients travel as MX e5m2; the weight gradients keep reading the bf16 copies
                if dfq is None:                                  # the last block's df comes from the eager prologue: a pass of its own
                    dfq = mx8.quantize(df, gf)
                dh, dhq = mx8.linear(dfq, wtq[4 * k + 3], gate=h, gate_mode=mx8.GATE_GELU, a_fmt=gf, out_mx=gf)
                dy2 = mx8.linear(dhq, wtq[4 * k + 2], a_fmt=gf)
            elif own:
                dh = igemm.linear(df, f2w_t, gate=h, gate_mode=igemm.GATE_GELU)        # (df W2) * GELU'(h) in the epilogue
                dy2 = igemm.linear(dh, f1w_t)
            else:
                da = torch.mm(df, _bf(f2w))
                dh = torch.ops.aten.gelu_backward(da, h)
                dy2 = torch.mm(dh, _bf(f1w))
            g[9], g[10] = _wgrad(dh, y2, f1w, f1b, big, pend)
            # LayerNorm 2 + the residual it sits on; gradient of the (window-major) proj output rides out as `dr`
            sc1 = dp[k, 0] if dp is not None else None
            if mx:                                               # the rows LayerNorm' writes leave as MX operands of the next input-gradient GEMM
                ds2, dpo, dpoq = rows.ln_bwd(dy2, None, L, dsup, s2, st2, n2w, True, ymap, S, sc1, zero, norm_g[k, 2], norm_g[k, 3], B, L, mx=gf, **rep)
                dao = mx8.linear(dpoq, wtq[4 * k + 1], a_fmt=gf)
            else:
                ds2, dpo = rows.ln_bwd(dy2, None, L, dsup, s2, st2, n2w, True, ymap, S, sc1, zero, norm_g[k, 2], norm_g[k, 3], B, L, **rep)
                dao = igemm.linear(dpo, pw_t) if own else torch.mm(dpo, _bf(pw))
            g[5], g[6] = _wgrad(dpo, ao.view(-1, C), pw, pb, big, pend)
            if mx:
                dqkv, dtable, dqkvq = wattn.bwd_raw(qkv.view(B * nW, wattn.TOKENS, 3 * C), table, regions, ao,
                                                    dao.view(B * nW, wattn.TOKENS, C), lse, spec["scale"], nW,
                                                    dtable=tables_g[k] if tables_g is not None else None, mx=gf)
                dqkv = dqkv.view(-1, 3 * C)
                dy1 = mx8.linear(dqkvq, wtq[4 * k], a_fmt=gf)
            else:
                dqkv, dtable = wattn.bwd_raw(qkv.view(B * nW, wattn.TOKENS, 3 * C), table, regions, ao,
                                             dao.view(B * nW, wattn.TOKENS, C), lse, spec["scale"], nW,
                                             dtable=tables_g[k] if tables_g is not None else None)
                dqkv = dqkv.view(-1, 3 * C)
                dy1 = igemm.linear(dqkv, qw_t) if own else torch.mm(dqkv, _bf(qw))
            g[4] = dtable
            g[2], g[3] = _wgrad(dqkv, y1, qw, qb, big, pend)
            if pend:
                igemm.wgrad_seq(pend)
            # LayerNorm 1; for k > 0 its input was (block k-1 stream + DropPath * MLP output): `dr` = that MLP's output gradient
            prev = (dp[k - 1, 1] if dp is not None else None) if k > 0 else None
            if mx and k > 0:
                ds1, df, dfq = rows.ln_bwd(dy1, ymap, S, ds2, s1, st1, n1w, True, None, L, prev, None, norm_g[k, 0], norm_g[k, 1], B, L, mx=gf, **rep)
            else:
                ds1, df = rows.ln_bwd(dy1, ymap, S, ds2, s1, st1, n1w, k > 0, None, L, prev, None, norm_g[k, 0], norm_g[k, 1], B, L, **rep)
            dsup = ds1
            grads[k * N_BLOCK:(k + 1) * N_BLOCK] = g
        norm_t = torch.zeros((depth, 4, C), dtype=torch.float32, device=dev)
        _rw.colsum_acc(norm_r.view(NREP, depth * 4 * C), norm_t.view(-1))          # the copies' sum
        for k in range(depth):
            grads[k * N_BLOCK + 0], grads[k * N_BLOCK + 1], grads[k * N_BLOCK + 7], grads[k * N_BLOCK + 8] = norm_t[k, 0], norm_t[k, 1], norm_t[k, 2], norm_t[k, 3]
        if big:
            conv_bf16.submit(big)                                # joins the step's deferred group when engine/trainer.py opened one
        return dsup, grads


def block_params(blk):
    return (blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.qkv.bias, blk.attn.relative_position_bias_table,
            blk.attn.proj.weight, blk.attn.proj.bias, blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias,
            blk.mlp.fc2.weight, blk.mlp.fc2.bias)


def supported(layer, x):
    """bf16-autocast GPU run of a standard window-12, head_dim-32 stage (Swin-B / Swin-L as shipped) without attention /
    projection / MLP dropout and without activation checkpointing; anything else takes the module-by-module path."""
    import torch.nn as nn
    b0 = layer.blocks[0]
    C = x.shape[-1]
    ok = (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and torch.is_autocast_enabled()
          and torch.get_autocast_dtype("cuda") == torch.bfloat16 and not layer.use_checkpoint
          and layer.window_size == wattn.WINDOW and C == b0.num_heads * wattn.HEAD_DIM and C in rows.WIDTHS)
    for blk in layer.blocks:
        ok = ok and isinstance(blk.norm1, nn.LayerNorm) and isinstance(blk.mlp.act, nn.GELU) and blk.attn.qkv.bias is not None
        ok = ok and blk.attn.attn_drop.p == 0.0 and blk.attn.proj_drop.p == 0.0 and blk.mlp.drop.p == 0.0
        ok = ok and getattr(blk.mlp.act, "approximate", "none") == "none" and blk.norm1.eps == b0.norm1.eps == blk.norm2.eps
    return ok


def wants_mx8(C):
    """MODEL.SWIN.FP8_GEMM (BASELINE config 5): a stage runs its Linears on pd_mx8_gemm when its width reaches FP8_MIN_K and every
    contraction of the block (C, 3 C, 4 C: forward and input gradient) is whole 128-byte K-steps"""
    from .swin import FP8
    return bool(FP8["enabled"] and C >= FP8["min_k"] and C % 128 == 0)


def mx8_ready(layer):
    """the stage's weights are what the MX path of the fused stage takes (bf16 copies of the master weights, as engine/flat_params.py keeps them)"""
    return _own_w(*[w for blk in layer.blocks for w in (blk.attn.qkv.weight, blk.attn.proj.weight, blk.mlp.fc1.weight, blk.mlp.fc2.weight)])


def draw_drop_path(net, B, device):
    """the DropPath scales (keep mask / keep_prob, reference :35-51) of EVERY stage of the backbone in four launches — random numbers, + keep_prob,
    floor, / keep_prob — instead of four per stage; run_stage() picks its [depth, 2, B] slice up"""
    rates = [float(getattr(blk.drop_path, "drop_prob", 0.0) or 0.0) for layer in net.layers for blk in layer.blocks]
    if not net.training or not any(r > 0 for r in rates):
        return
    key = (tuple(rates), str(device))
    if getattr(net, "_keep_key", None) != key:                            # built once: a host list -> device copy synchronises
        net._keep = torch.tensor([1.0 - r for r in rates], dtype=torch.float32, device=device).view(len(rates), 1, 1)
        net._keep_key = key
    dp = (torch.rand((len(rates), 2, B), device=device) + net._keep).floor_().div_(net._keep)
    o = 0
    for layer in net.layers:
        d = len(layer.blocks)
        if layer.training and any(r > 0 for r in rates[o:o + d]):
            layer._dp_drawn = dp[o:o + d]
        o += d


def run_stage(layer, x, H, W, out_norm=None):
    """-> (stage output fp32 [B, L, C], LayerNorm of it by `out_norm` (an nn.LayerNorm: the backbone's norm{i}) or None when out_norm is None / not fusable)"""
    B = x.shape[0]
    depth = len(layer.blocks)
    rates = [float(getattr(blk.drop_path, "drop_prob", 0.0) or 0.0) for blk in layer.blocks]
    dp = layer.__dict__.pop("_dp_drawn", None)                            # drawn for all stages at once by draw_drop_path()
    if dp is not None and (tuple(dp.shape) != (depth, 2, B) or dp.device != x.device or not layer.training):
        dp = None
    if dp is None and layer.training and any(r > 0 for r in rates):
        key = (tuple(rates), str(x.device))
        if getattr(layer, "_keep_key", None) != key:                      # built once: a host list -> device copy synchronises
            layer._keep = torch.tensor([1.0 - r for r in rates], dtype=torch.float32, device=x.device).view(depth, 1, 1)
            layer._keep_key = key
        dp = (torch.rand((depth, 2, B), device=x.device) + layer._keep).floor_().div_(layer._keep)   # DropPath (:35-51)
    nw = nb = None
    if (out_norm is not None and isinstance(out_norm, torch.nn.LayerNorm) and out_norm.elementwise_affine and out_norm.bias is not None
            and tuple(out_norm.normalized_shape) == (x.shape[-1],)):
        nw, nb = out_norm.weight, out_norm.bias
    spec = dict(H=H, W=W, heads=layer.blocks[0].num_heads, shifts=[blk.shift_size for blk in layer.blocks],
                scale=layer.blocks[0].attn.scale, eps=layer.blocks[0].norm1.eps, dp=dp, mx8=wants_mx8(x.shape[-1]) and mx8_ready(layer),
                out_eps=out_norm.eps if nw is not None else 0.0)
    params = [p for blk in layer.blocks for p in block_params(blk)]
    # stages 2-4 receive the bf16 output of PatchMerging's Linear; the module-by-module path (like the reference under AMP)
    # then keeps a 16-bit residual stream, this one keeps it in fp32 throughout
    return SwinStage.apply(x.float(), spec, nw, nb, *params)
