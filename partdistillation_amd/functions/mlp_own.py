"""The decoder's mask-embedding MLP (Linear-ReLU-Linear-ReLU-Linear over the [heads, B, Q, C] decoder outputs; reference
mask2former_transformer_decoder.py:198-204, used at :447) as ONE autograd node on pd_igemm_bf16 / pd_wgrad_bf16 under bf16 autocast:
bias + ReLU in the forward epilogues, the ReLU masks in the input-gradient epilogues, the three weight gradients and their bias sums by
one pd_wgrad_bf16_seq call.  The module path (nn.Linear under autocast) casts per layer and runs ~25 library / ATen launches for the
same 2 000 rows; this is 12.  GPU + bf16 autocast only — MLP.forward keeps the module path otherwise."""
import torch
from torch.autograd import Function

from . import igemm


class MlpOwn(Function):
    @staticmethod
    def forward(ctx, x, *params):
        """x [..., K] (fp32 or bf16), params = (w0, b0, w1, b1, ...) bf16 weights [N_i, K_i], biases bf16 / fp32"""
        n = len(params) // 2
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        x2 = x2 if x2.dtype == torch.bfloat16 else x2.to(torch.bfloat16)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        acts = [x2]
        for i in range(n):
            acts.append(igemm.linear(acts[-1], params[2 * i], params[2 * i + 1], act=igemm.ACT_RELU if i < n - 1 else igemm.ACT_NONE))
        ctx.save_for_backward(*acts[:-1], *params[0::2])
        ctx.n, ctx.shp, ctx.xdt, ctx.bdt = n, shp, x.dtype, [params[2 * i + 1].dtype for i in range(n)]
        return acts[-1].view(*shp[:-1], params[2 * (n - 1)].shape[0])

    @staticmethod
    def backward(ctx, dy):
        n = ctx.n
        acts, ws = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        g = dy.reshape(-1, ws[-1].shape[0])
        g = g if g.dtype == torch.bfloat16 else g.to(torch.bfloat16)
        g = g if g.is_contiguous() else g.contiguous()
        need_x = ctx.needs_input_grad[0]
        wts = igemm.transposed(list(ws[(0 if need_x else 1):])) if (need_x or n > 1) else []
        wts = ([None] if not need_x else []) + list(wts)
        gs = [None] * n
        gs[n - 1] = g
        for i in range(n - 1, 0, -1):                                   # through the ReLU of layer i - 1: * (its output > 0)
            gs[i - 1] = igemm.linear(gs[i], wts[i], gate=acts[i], gate_mode=igemm.GATE_RELU)
        dx = None
        if need_x:
            dx = igemm.linear(gs[0], wts[0]).view(ctx.shp)
            dx = dx if dx.dtype == ctx.xdt else dx.to(ctx.xdt)
        offs, tot = [], 0
        for w in ws:
            offs.append(tot)
            tot += w.shape[0]
        db_all = torch.zeros(tot, dtype=torch.float32, device=g.device)
        dws = [torch.empty_like(w) for w in ws]
        igemm.wgrad_seq([(gs[i], acts[i], dws[i], db_all[offs[i]:offs[i] + ws[i].shape[0]]) for i in range(n)])
        db16 = db_all.to(torch.bfloat16) if any(dt == torch.bfloat16 for dt in ctx.bdt) else None
        out = [dx]
        for i in range(n):
            src = db16 if ctx.bdt[i] == torch.bfloat16 else db_all
            out += [dws[i], src[offs[i]:offs[i] + ws[i].shape[0]]]
        return tuple(out)


class HeadsOwn(Function):
    """Both prediction heads of the decoder over the same [heads, B, Q, C] outputs as one node: the class head (K + 1 <= 8 columns:
    pd_skinny_linear_*, fp32 logits) and the mask-embedding MLP above.  One bf16 copy of the input serves both; the backward adds the
    class head's input gradient to the MLP's while converting it to the input's dtype (reference mask2former_transformer_decoder.py:446-447)."""

    @staticmethod
    def forward(ctx, x, cw, cb, *params):
        from .. import lib as _lib
        n = len(params) // 2
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        x2 = x2 if x2.dtype == torch.bfloat16 else x2.to(torch.bfloat16)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        R, C, K = x2.shape[0], x2.shape[1], cw.shape[0]
        logits = torch.empty((R, K), dtype=torch.float32, device=x.device)
        cb_dtype = cb.dtype
        cw, cb = cw.contiguous(), cb.to(cw.dtype).contiguous()
        _lib.check(_lib.load().pd_skinny_linear_fwd(x2.data_ptr(), cw.data_ptr(), cb.data_ptr(), _DT[cw.dtype], logits.data_ptr(), R, C, K,
                                                    _lib.current_stream()))
        acts = [x2]
        for i in range(n):
            acts.append(igemm.linear(acts[-1], params[2 * i], params[2 * i + 1], act=igemm.ACT_RELU if i < n - 1 else igemm.ACT_NONE))
        ctx.save_for_backward(cw, *acts[:-1], *params[0::2])
        ctx.n, ctx.shp, ctx.xdt, ctx.bdt, ctx.cbdt, ctx.cb_param_dtype = n, shp, x.dtype, [params[2 * i + 1].dtype for i in range(n)], cb.dtype, cb_dtype
        return logits.view(*shp[:-1], K), acts[-1].view(*shp[:-1], params[2 * (n - 1)].shape[0])

    @staticmethod
    def backward(ctx, dlog, dy):
        from .. import lib as _lib
        n = ctx.n
        cw, acts, ws = ctx.saved_tensors[0], ctx.saved_tensors[1:n + 1], ctx.saved_tensors[n + 1:]
        x2 = acts[0]
        R, C, K = x2.shape[0], x2.shape[1], cw.shape[0]
        need_x = ctx.needs_input_grad[0]
        out_mlp, dx_mlp = [None] * (2 * n), None
        if dy is not None:
            g = dy.reshape(-1, ws[-1].shape[0])
            g = g if g.dtype == torch.bfloat16 else g.to(torch.bfloat16)
            g = g if g.is_contiguous() else g.contiguous()
            wts = igemm.transposed(list(ws[(0 if need_x else 1):])) if (need_x or n > 1) else []
            wts = ([None] if not need_x else []) + list(wts)
            gs = [None] * n
            gs[n - 1] = g
            for i in range(n - 1, 0, -1):
                gs[i - 1] = igemm.linear(gs[i], wts[i], gate=acts[i], gate_mode=igemm.GATE_RELU)
            if need_x:
                dx_mlp = igemm.linear(gs[0], wts[0])                          # bf16 [R, C]
            offs, tot = [], 0
            for w in ws:
                offs.append(tot)
                tot += w.shape[0]
            db_all = torch.zeros(tot, dtype=torch.float32, device=g.device)
            dws = [torch.empty_like(w) for w in ws]
            igemm.wgrad_seq([(gs[i], acts[i], dws[i], db_all[offs[i]:offs[i] + ws[i].shape[0]]) for i in range(n)])
            db16 = db_all.to(torch.bfloat16) if any(dt == torch.bfloat16 for dt in ctx.bdt) else None
            for i in range(n):
                src = db16 if ctx.bdt[i] == torch.bfloat16 else db_all
                out_mlp[2 * i], out_mlp[2 * i + 1] = dws[i], src[offs[i]:offs[i] + ws[i].shape[0]]
        dx = dcw = dcb = None
        if dlog is not None:
            lib = _lib.load()
            dl = dlog.reshape(R, K)
            dl = dl if dl.dtype == torch.float32 else dl.float()
            dl = dl if dl.is_contiguous() else dl.contiguous()
            dcw, dcb = torch.empty_like(cw), torch.empty((K,), dtype=ctx.cbdt, device=cw.device)
            part = torch.empty((lib.pd_skinny_linear_partial_floats(R, C, K),), dtype=torch.float32, device=cw.device)
            if need_x:
                dx = torch.empty((R, C), dtype=ctx.xdt, device=cw.device)
            _lib.check(lib.pd_skinny_linear_bwd(x2.data_ptr(), cw.data_ptr(), _DT[cw.dtype], dl.data_ptr(),
                                                dx_mlp.data_ptr() if dx_mlp is not None else None, dx.data_ptr() if dx is not None else None,
                                                _DT[ctx.xdt], part.data_ptr(), dcw.data_ptr(), dcb.data_ptr(), _DT[ctx.cbdt], R, C, K,
                                                _lib.current_stream()))
            dx = dx.view(ctx.shp) if dx is not None else None
            dcb = dcb if dcb.dtype == ctx.cb_param_dtype else dcb.to(ctx.cb_param_dtype)      # (a bias kept in another dtype than the weight)
        elif dx_mlp is not None:
            dx = dx_mlp.view(ctx.shp)
            dx = dx if dx.dtype == ctx.xdt else dx.to(ctx.xdt)
        return (dx, dcw, dcb, *out_mlp)


_DT = {torch.float32: 0, torch.bfloat16: 2}
HEADS = __import__("os").environ.get("PD_HEADS_OWN", "1") != "0"                # 0: class head as nn.Linear beside the MLP node (tools/ A/B runs)


def heads_supported(x, class_layer, layers):
    return (HEADS and supported(x, layers) and isinstance(class_layer, torch.nn.Linear) and class_layer.bias is not None
            and class_layer.weight.dtype in _DT and class_layer.bias.dtype in _DT and class_layer.out_features <= 8
            and class_layer.in_features == x.shape[-1] and x.shape[-1] % 4 == 0 and x.dtype in _DT)


def heads(x, class_layer, layers):
    """-> (class logits [..., K + 1] fp32, mask embeddings [..., mask_dim] bf16)"""
    return HeadsOwn.apply(x, class_layer.weight, class_layer.bias, *[p for l in layers for p in (l.weight, l.bias)])


ENABLED = __import__("os").environ.get("PD_MLP_OWN", "1") != "0"                # 0: the module path (tools/ A/B runs)


def supported(x, layers):
    if not (ENABLED and x.is_cuda and torch.is_grad_enabled() and len(layers) <= igemm.WGRAD_SEQ_MAX):
        return False
    for l in layers:
        if l.bias is None or not igemm.own_linear_supported(x, l.weight) or l.bias.dtype not in (torch.bfloat16, torch.float32):
            return False
    return x.numel() // x.shape[-1] >= 64 and x.dtype in (torch.float32, torch.bfloat16)


def mlp(x, layers):
    return MlpOwn.apply(x, *[p for l in layers for p in (l.weight, l.bias)])
