"""Token-row kernels of the fused Swin block (pd_swin_ln_fwd / pd_swin_ln_bwd, include/pd_swin.h): residual add with a
DropPath scale + LayerNorm, reading / writing through row maps (window partition of the padded, shifted grid)."""
import torch

from .. import lib as _lib

WIDTHS = tuple(64 * e for e in (1, 2, 3, 4, 6, 8, 12, 16, 24))


def _p(t):
    return t.data_ptr() if t is not None else None


def ln_fwd(x, r, rmap, r_rows, rscale, gamma, beta, eps, ymap, y_rows, zero_rows, images, L, mx=None):
    """x fp32 [images*L, C]; r bf16 rows (or None) -> (s fp32 [images*L, C] (x itself when r is None), y bf16
    [images*y_rows, C], [mean, rstd]) and, with mx = an fp8 format of functions/mx8.py, y again as an MX operand (q uint8 [rows, C],
    scales uint8 [rows, C / 32])"""
    if not x.is_cuda:
        raise RuntimeError("pd_swin_ln_fwd runs on the GPU only (no CPU fallback in partdistillation_amd)")
    C = x.shape[-1]
    assert x.dtype == torch.float32 and x.is_contiguous() and gamma.dtype == torch.float32 and (r is None or (r.dtype == torch.bfloat16 and r.is_contiguous()))
    s = torch.empty_like(x) if r is not None else x
    y = torch.empty((images * y_rows, C), dtype=torch.bfloat16, device=x.device)
    stats = torch.empty((2, images * L), dtype=torch.float32, device=x.device)
    nz = 0 if zero_rows is None else zero_rows.numel()
    yq = torch.empty((images * y_rows, C), dtype=torch.uint8, device=x.device) if mx is not None else None
    ys = torch.empty((images * y_rows, C // 32), dtype=torch.uint8, device=x.device) if mx is not None else None
    _lib.check(_lib.load().pd_swin_ln_fwd(x.data_ptr(), _p(r), _p(rmap), r_rows, _p(rscale), gamma.data_ptr(), beta.data_ptr(),
                                          float(eps), s.data_ptr(), y.data_ptr(), _p(ymap), y_rows, _p(zero_rows), nz,
                                          stats[0].data_ptr(), stats[1].data_ptr(), images, L, C, _p(yq), _p(ys), mx if mx is not None else 0,
                                          _lib.current_stream()))
    return (s, y, stats) if mx is None else (s, y, stats, (yq, ys))


def ln_bwd(dy, ymap, y_rows, dsup, s, stats, gamma, want_dr, rmap, r_rows, rscale, zero_rows, dgamma, dbeta, images, L, mx=None, n_rep=1, rep_stride=0):
    """-> (ds fp32 [images*L, C], dr bf16 [images*r_rows, C] or None) and, with mx = an fp8 format and want_dr, dr again as an MX operand;
    dgamma / dbeta (fp32 [C]) are accumulated into — copy (workgroup % n_rep) of them, rep_stride floats apart, when n_rep > 1 (the caller sums)"""
    C = s.shape[-1]
    assert dy.dtype == torch.bfloat16 and dy.is_contiguous() and (dsup is None or (dsup.dtype == torch.float32 and dsup.is_contiguous()))
    ds = torch.empty_like(s)
    dr = torch.empty((images * r_rows, C), dtype=torch.bfloat16, device=s.device) if want_dr else None
    nz = 0 if (zero_rows is None or not want_dr) else zero_rows.numel()
    mx = mx if want_dr else None
    dq = torch.empty((images * r_rows, C), dtype=torch.uint8, device=s.device) if mx is not None else None
    dsc = torch.empty((images * r_rows, C // 32), dtype=torch.uint8, device=s.device) if mx is not None else None
    _lib.check(_lib.load().pd_swin_ln_bwd(dy.data_ptr(), _p(ymap), y_rows, _p(dsup), s.data_ptr(), stats[0].data_ptr(),
                                          stats[1].data_ptr(), gamma.data_ptr(), ds.data_ptr(), _p(dr), _p(rmap), r_rows,
                                          _p(rscale), _p(zero_rows) if nz else None, nz, dgamma.data_ptr(), dbeta.data_ptr(),
                                          images, L, C, _p(dq), _p(dsc), mx if mx is not None else 0, int(n_rep), int(rep_stride), _lib.current_stream()))
    return (ds, dr) if mx is None else (ds, dr, (dq, dsc))


class _RowsLayerNorm(torch.autograd.Function):
    """fp32 LayerNorm over the last dimension on pd_layernorm_rows_f32_{fwd,bwd} (the Swin stages' output norms, reference swin.py:675-680; the patch-merging
    norms over 4 C channels :339 and the patch embedding's :565)"""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        stats = torch.empty((2, rows), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().pd_layernorm_rows_f32_fwd(x2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), y.data_ptr(), stats[0].data_ptr(),
                                                         stats[1].data_ptr(), rows, C, _lib.current_stream()))
        ctx.save_for_backward(x2, stats, gamma)
        ctx.shape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, stats, gamma = ctx.saved_tensors
        C = x2.shape[1]
        g = dy.reshape(-1, C)
        g = g if (g.is_contiguous() and g.dtype == torch.float32) else g.float().contiguous()
        dx = torch.empty_like(x2)
        dgb = torch.zeros((2, C), dtype=torch.float32, device=x2.device)
        _lib.check(_lib.load().pd_layernorm_rows_f32_bwd(g.data_ptr(), x2.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), gamma.data_ptr(), dx.data_ptr(),
                                                         dgb[0].data_ptr(), dgb[1].data_ptr(), x2.shape[0], C, _lib.current_stream()))
        return dx.view(ctx.shape), dgb[0], dgb[1], None


def rows_layer_norm_supported(x, ln):
    return (x.is_cuda and x.dtype == torch.float32 and x.shape[-1] % 4 == 0 and x.shape[-1] <= 3072 and ln.weight is not None and ln.bias is not None
            and ln.weight.dtype == torch.float32 and ln.bias.dtype == torch.float32 and tuple(ln.normalized_shape) == (x.shape[-1],))


def rows_layer_norm(x, ln):
    return _RowsLayerNorm.apply(x, ln.weight, ln.bias, ln.eps)


class _MergeLayerNorm(torch.autograd.Function):
    """PatchMerging's 2 x 2 gather + LayerNorm over 4 C channels on pd_swin_merge_ln_{fwd,bwd} (reference swin.py:325-339): x fp32 [B, H W, C] -> bf16
    [B, (H/2)(W/2), 4 C], the operand of the reduction Linear; the gradient arrives in 16 bits from that Linear and leaves as fp32 dx"""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, H, W):
        B, L, C = x.shape
        x = x if x.is_contiguous() else x.contiguous()
        rows = B * (H // 2) * (W // 2)
        y = torch.empty((B, rows // B, 4 * C), dtype=torch.bfloat16, device=x.device)
        stats = torch.empty((2, rows), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().pd_swin_merge_ln_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
                                                    B, H, W, C, _lib.current_stream()))
        ctx.save_for_backward(x, stats, gamma)
        ctx.hw = (H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats, gamma = ctx.saved_tensors
        B, L, C = x.shape
        H, W = ctx.hw
        g = dy if dy.dtype == torch.bfloat16 else dy.to(torch.bfloat16)
        g = g if g.is_contiguous() else g.contiguous()
        dx = torch.empty_like(x)
        dgb = torch.zeros((2, 4 * C), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().pd_swin_merge_ln_bwd(g.data_ptr(), x.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), gamma.data_ptr(), dx.data_ptr(),
                                                    dgb[0].data_ptr(), dgb[1].data_ptr(), B, H, W, C, _lib.current_stream()))
        return dx, dgb[0], dgb[1], None, None, None


def merge_layer_norm_supported(x, H, W, ln):
    C = x.shape[-1]
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[1] == H * W and H % 2 == 0 and W % 2 == 0 and C % 4 == 0 and 4 * C <= 3072
            and ln.weight is not None and ln.bias is not None and ln.weight.dtype == torch.float32 and ln.bias.dtype == torch.float32
            and tuple(ln.normalized_shape) == (4 * C,))


def merge_layer_norm(x, H, W, ln):
    return _MergeLayerNorm.apply(x, ln.weight, ln.bias, ln.eps, H, W)


def tail_ln_supported(C, nw, nb):
    return (C % 4 == 0 and C <= 3072 and nw is not None and nb is not None and nw.dtype == torch.float32 and nb.dtype == torch.float32
            and nw.numel() == C and nb.numel() == C and nw.is_contiguous() and nb.is_contiguous())


def tail_ln_fwd(cur, r, rscale, L, gamma, beta, eps):
    """the end of a fused Swin stage + its output norm (pd_swin_tail_ln_fwd): cur fp32 [rows, C], r bf16 [rows, C], rscale fp32 [images] or None
    -> (s = cur + rscale r fp32, y = LN(s) fp32, [mean, rstd])"""
    rows, C = cur.shape
    assert cur.dtype == torch.float32 and cur.is_contiguous() and r.dtype == torch.bfloat16 and r.is_contiguous() and r.shape == cur.shape
    s, y = torch.empty_like(cur), torch.empty_like(cur)
    stats = torch.empty((2, rows), dtype=torch.float32, device=cur.device)
    _lib.check(_lib.load().pd_swin_tail_ln_fwd(cur.data_ptr(), r.data_ptr(), _p(rscale), int(L), gamma.data_ptr(), beta.data_ptr(), float(eps), s.data_ptr(),
                                               y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), rows, C, _lib.current_stream()))
    return s, y, stats


def tail_ln_bwd(dy, dsum, s, stats, gamma, rscale, L):
    """-> (dsup fp32 = dsum + LayerNorm'(dy), df bf16 = rscale dsup, dgamma, dbeta)"""
    rows, C = s.shape
    assert dy.dtype == torch.float32 and dy.is_contiguous() and (dsum is None or (dsum.dtype == torch.float32 and dsum.is_contiguous()))
    dsup = torch.empty_like(s)
    df = torch.empty((rows, C), dtype=torch.bfloat16, device=s.device)
    dgb = torch.zeros((2, C), dtype=torch.float32, device=s.device)
    _lib.check(_lib.load().pd_swin_tail_ln_bwd(dy.data_ptr(), _p(dsum), s.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), gamma.data_ptr(), _p(rscale), int(L),
                                               dsup.data_ptr(), df.data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr(), rows, C, _lib.current_stream()))
    return dsup, df, dgb[0], dgb[1]
