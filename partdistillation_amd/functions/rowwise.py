"""ctypes faces of the token-wise kernels (include/pd_rowwise.h).  Plain functions on raw tensors — no autograd: they
are the building blocks of the hand-written forward/backward passes (functions/decoder_core.py, AddLayerNorm below).
GPU only; there is no fallback."""
import ctypes

import torch
from torch.autograd import Function

from .. import lib as _lib

_DT = {torch.float32: _lib.PD_F32, torch.bfloat16: _lib.PD_BF16}


def _stream():
    return _lib.current_stream()


def _p(t):
    return None if t is None else t.data_ptr()


def _need_cuda(t, who):
    if not t.is_cuda:
        raise RuntimeError(f"{who} runs on the GPU only (no CPU fallback in partdistillation_amd)")


def add_ln_fwd(x, res, gamma, beta, eps, *, want_z=True, want_y=True, c_dtype=None, want_yc=False, pos=None, pos_div=1,
               want_ypos=False, amax=False):
    """-> (z, y, y_c, ypos_c, mean, rstd); x [rows,C] (fp32 | bf16) and / or res fp32 [rows,C].
    amax: also -> (..., y_amax, ypos_amax | None), the absolute row maxima of y and y + pos (pd_add_layernorm_fwd_amax)."""
    ref = x if x is not None else res
    _need_cuda(ref, "pd_add_layernorm_fwd")
    rows, C = ref.shape
    dev = ref.device
    assert (x is None or x.is_contiguous()) and (res is None or (res.is_contiguous() and res.dtype == torch.float32))
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.numel() == C and beta.numel() == C
    z = torch.empty((rows, C), dtype=torch.float32, device=dev) if want_z else None
    y = torch.empty((rows, C), dtype=torch.float32, device=dev) if want_y else None
    y_c = torch.empty((rows, C), dtype=c_dtype, device=dev) if want_yc else None
    ypos_c = torch.empty((rows, C), dtype=c_dtype, device=dev) if want_ypos else None
    stats = torch.empty((2, rows), dtype=torch.float32, device=dev)
    if want_ypos:
        assert pos is not None and pos.dtype == torch.float32 and pos.is_contiguous() and pos.shape[-1] == C
    if amax:
        am = torch.empty((2, rows), dtype=torch.float32, device=dev)
        _lib.check(_lib.load().pd_add_layernorm_fwd_amax(
            _p(x), _DT[x.dtype] if x is not None else 0, _p(res), gamma.data_ptr(), beta.data_ptr(), float(eps), _p(z), _p(y), _p(y_c),
            _p(pos) if want_ypos else None, int(pos_div), _p(ypos_c), _DT[c_dtype] if c_dtype is not None else 0,
            stats[0].data_ptr(), stats[1].data_ptr(), am[0].data_ptr(), am[1].data_ptr() if want_ypos else None, rows, C, _stream()))
        return z, y, y_c, ypos_c, stats[0], stats[1], am[0], (am[1] if want_ypos else None)
    _lib.check(_lib.load().pd_add_layernorm_fwd(
        _p(x), _DT[x.dtype] if x is not None else 0, _p(res), gamma.data_ptr(), beta.data_ptr(), float(eps), _p(z), _p(y), _p(y_c),
        _p(pos) if want_ypos else None, int(pos_div), _p(ypos_c), _DT[c_dtype] if c_dtype is not None else 0,
        stats[0].data_ptr(), stats[1].data_ptr(), rows, C, _stream()))
    return z, y, y_c, ypos_c, stats[0], stats[1]


def add_ln_bwd(z, mean, rstd, gamma, *, dy=None, dy2=None, dy_c=None, dypos_c=None, dz_c_dtype=None, dgamma=None, dbeta=None,
               dbias=None, dpos_acc=None, pos_div=1, out=None, amax=False):
    """-> (dz fp32, dz_c | None).  dgamma / dbeta / dbias / dpos_acc are fp32 accumulators (+=).
    amax: -> (dz, dz_c, dz_amax), dz_amax [rows] = the absolute row maxima of dz (pd_add_layernorm_bwd_amax)."""
    rows, C = z.shape
    assert z.dtype == torch.float32 and gamma.dtype == torch.float32 and mean.dtype == torch.float32
    cd = None
    for t in (dy_c, dypos_c):
        if t is not None:
            assert t.is_contiguous() and (cd is None or cd == t.dtype)
            cd = t.dtype
    for t in (dy, dy2):
        assert t is None or (t.is_contiguous() and t.dtype == torch.float32)
    dz = out if out is not None else torch.empty((rows, C), dtype=torch.float32, device=z.device)
    dz_c = torch.empty((rows, C), dtype=dz_c_dtype, device=z.device) if dz_c_dtype is not None else None
    if amax:
        am = torch.empty(rows, dtype=torch.float32, device=z.device)
        _lib.check(_lib.load().pd_add_layernorm_bwd_amax(
            _p(dy), _p(dy2), _p(dy_c), _p(dypos_c), _DT[cd] if cd is not None else 0, z.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
            gamma.data_ptr(), dz.data_ptr(), _p(dz_c), _DT[dz_c_dtype] if dz_c_dtype is not None else 0, _p(dgamma), _p(dbeta),
            _p(dbias), _p(dpos_acc), int(pos_div), am.data_ptr(), rows, C, _stream()))
        return dz, dz_c, am
    _lib.check(_lib.load().pd_add_layernorm_bwd(
        _p(dy), _p(dy2), _p(dy_c), _p(dypos_c), _DT[cd] if cd is not None else 0, z.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
        gamma.data_ptr(), dz.data_ptr(), _p(dz_c), _DT[dz_c_dtype] if dz_c_dtype is not None else 0, _p(dgamma), _p(dbeta),
        _p(dbias), _p(dpos_acc), int(pos_div), rows, C, _stream()))
    return dz, dz_c


def copy_d2d(dst, src):
    """dst <- src (same byte size, both dense): hipMemcpyAsync on the launch stream as a C-ABI call (recordable: cmdbuf.py)"""
    n = src.numel() * src.element_size()
    assert dst.numel() * dst.element_size() == n and dst.is_contiguous() and src.is_contiguous()
    _lib.check(_lib.load().pd_memcpy_d2d_async(dst.data_ptr(), src.data_ptr(), n, _stream()))
    return dst


class _CopySeg(ctypes.Structure):                                   # PdCopySeg (include/pd_rowwise.h)
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("bytes", ctypes.c_int64)]


MAX_COPY_SEGS = 48


def copy_segments(pairs):
    """[(dst, src), ...] dense tensors of equal byte size: all copies in ONE launch (pd_copy_segments) — concatenations of slices of
    several tensors without a torch.cat per destination.  Inside a recorded region the sources must be stable (parameters)."""
    from .. import cmdbuf
    assert 0 < len(pairs) <= MAX_COPY_SEGS
    segs = (_CopySeg * len(pairs))()
    for sgm, (dst, src) in zip(segs, pairs):
        n = src.numel() * src.element_size()
        assert dst.numel() * dst.element_size() == n and dst.is_contiguous() and src.is_contiguous()
        if cmdbuf.active() is not None:
            cmdbuf.require_stable(src.data_ptr(), "source of a grouped copy")
        sgm.src, sgm.dst, sgm.bytes = src.data_ptr(), dst.data_ptr(), n
    _lib.check(_lib.load().pd_copy_segments(segs, len(pairs), _stream()))


def add_rows_amax(a, b, copy_a=False):
    """-> (q = a + b, a copy of a or None, row maxima of a, row maxima of q): fp32 [rows, cols] contiguous, one launch (pd_add_rows_amax_f32)"""
    assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 2
    q = torch.empty_like(a)
    ac = torch.empty_like(a) if copy_a else None
    am = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
    qm = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
    _lib.check(_lib.load().pd_add_rows_amax_f32(a.data_ptr(), b.data_ptr(), q.data_ptr(), ac.data_ptr() if ac is not None else None, am.data_ptr(),
                                                qm.data_ptr(), a.shape[0], a.shape[1], _stream()))
    return q, ac, am, qm


def colsum_acc(x, acc):
    """acc[N] (fp32) += x.sum(0); x [rows, N] contiguous."""
    assert x.is_contiguous() and acc.dtype == torch.float32 and acc.numel() == x.shape[1]
    _lib.check(_lib.load().pd_colsum_acc(x.data_ptr(), _DT[x.dtype], x.shape[0], x.shape[1], acc.data_ptr(), _stream()))


def relu_bwd_colsum(dh, h, acc=None):
    """dh *= (h > 0) in place; acc[N] += dh.sum(0)."""
    assert dh.is_contiguous() and h.is_contiguous() and dh.dtype == h.dtype and dh.shape == h.shape
    _lib.check(_lib.load().pd_relu_bwd_colsum(dh.data_ptr(), h.data_ptr(), _DT[dh.dtype], dh.shape[0], dh.shape[1], _p(acc), _stream()))
    return dh


def _token_view(x):
    """[B,C,H,W] feature map -> (fp32 tensor holding it, batch stride) with element (b, p, c) at b*stride + p*C + c."""
    B, C, H, W = x.shape
    t = x.permute(0, 2, 3, 1)
    if t.dtype != torch.float32 or t.stride(3) != 1 or t.stride(2) != C or t.stride(1) != W * C:
        t = t.float().contiguous()
    return t, t.stride(0)


def mem_prep_fwd(x, level_embed, pos, c_dtype):
    """x [B,C,H,W] (any strides; channels-last views are read in place) -> mem_c, mempos_c  [HW*B, C] seq-first."""
    _need_cuda(x, "pd_mem_prep_fwd")
    B, C, H, W = x.shape
    t, bstride = _token_view(x)
    assert pos.dtype == torch.float32 and pos.is_contiguous() and (level_embed is None or level_embed.dtype == torch.float32)
    mem = torch.empty((H * W * B, C), dtype=c_dtype, device=x.device)
    mempos = torch.empty((H * W * B, C), dtype=c_dtype, device=x.device)
    _lib.check(_lib.load().pd_mem_prep_fwd(t.data_ptr(), bstride, _p(level_embed), pos.data_ptr(), mem.data_ptr(), mempos.data_ptr(),
                                           _DT[c_dtype], B, H * W, C, _stream()))
    return mem, mempos


def mem_prep_bwd(dmem, dmempos, B, H, W, C):
    """-> gradient of the [B,C,H,W] map, as a channels-last view of a fresh [B,HW,C] fp32 buffer."""
    ref = dmem if dmem is not None else dmempos
    dtok = torch.empty((B, H * W, C), dtype=torch.float32, device=ref.device)
    _lib.check(_lib.load().pd_mem_prep_bwd(_p(dmem), _p(dmempos), _DT[ref.dtype], dtok.data_ptr(), H * W * C, B, H * W, C, _stream()))
    return dtok


def attn_mask_u8(logits):
    """logits [..., n] -> uint8 mask of the same shape: 1 = blocked (logit < 0); rows blocked everywhere are released."""
    _need_cuda(logits, "pd_attn_mask_u8")
    assert logits.is_contiguous()
    n = logits.shape[-1]
    mask = torch.empty(logits.shape, dtype=torch.uint8, device=logits.device)
    _lib.check(_lib.load().pd_attn_mask_u8(logits.data_ptr(), _DT[logits.dtype], logits.numel() // max(n, 1), n, mask.data_ptr(), _stream()))
    return mask


def matcher_point_terms(x, want_f32=True):
    """x [..., n] fp32 / bf16 point logits -> (x as fp32 | None, sigmoid(x), sum softplus(x) over n, sum sigmoid(x) over n): the
    Hungarian matcher's per-point terms in one pass (pd_matcher_point_terms)"""
    _need_cuda(x, "pd_matcher_point_terms")
    assert x.is_contiguous() and x.dtype in _DT
    n = x.shape[-1]
    rows = x.numel() // max(n, 1)
    xf = torch.empty(x.shape, dtype=torch.float32, device=x.device) if want_f32 else None
    sg = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    if x.numel() == 0:                                          # no points (or no rows): empty maps, zero sums
        z = torch.zeros(x.shape[:-1], dtype=torch.float32, device=x.device)
        return xf, sg, z, z.clone()
    sums = torch.empty((2,) + tuple(x.shape[:-1]), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().pd_matcher_point_terms(x.data_ptr(), _DT[x.dtype], rows, n, xf.data_ptr() if xf is not None else None, sg.data_ptr(),
                                                  sums[0].data_ptr(), sums[1].data_ptr(), _stream()))
    return xf, sg, sums[0], sums[1]


def point_sample_nhwc(x, coords, out_dtype=torch.float32):
    """x [B,C,H,W] fp32 (channels-last memory is read in place), coords [B,P,2] (x, y) in [0,1] -> [B,P,C]:
    F.grid_sample(x, 2*coords-1, bilinear, zeros, align_corners=False) for points shared by all channels.
    out_dtype bfloat16: the fp32 samples are rounded on the way out (no separate cast pass)."""
    _need_cuda(x, "pd_point_sample_nhwc_f32")
    B, C, H, W = x.shape
    t = x.permute(0, 2, 3, 1)
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.float().contiguous()
    coords = coords.float().contiguous()
    P = coords.shape[1]
    out = torch.empty((B, P, C), dtype=out_dtype, device=x.device)
    fn = _lib.load().pd_point_sample_nhwc_f32_bf16 if out_dtype == torch.bfloat16 else _lib.load().pd_point_sample_nhwc_f32
    assert out_dtype in (torch.float32, torch.bfloat16)
    _lib.check(fn(t.data_ptr(), coords.data_ptr(), out.data_ptr(), B, H, W, C, P, _stream()))
    return out


class PointSamplePlanar(Function):
    """x [N,C,H,W] fp32 contiguous, coords [N,P,2] (x, y) in [0,1] -> [N,C,P]: F.grid_sample(x, 2*coords-1, bilinear, zeros,
    align_corners=False) at per-map points (pd_point_sample_planar_f32); gradient with respect to x only."""

    @staticmethod
    def forward(ctx, x, coords):
        N, C, H, W = x.shape
        P = coords.shape[1]
        out = torch.empty((N, C, P), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().pd_point_sample_planar_f32(x.data_ptr(), coords.data_ptr(), out.data_ptr(), N, C, H, W, P, _stream()))
        ctx.save_for_backward(coords)
        ctx.shape = (N, C, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        (coords,) = ctx.saved_tensors
        N, C, H, W = ctx.shape
        L = _lib.load()
        alloc = torch.zeros if L.pd_point_sample_planar_bwd_needs_zero_n(N, C, H, W) else torch.empty
        gx = alloc((N, C, H, W), dtype=torch.float32, device=g.device)
        g = g.contiguous()
        _lib.check(_lib.load().pd_point_sample_planar_bwd_f32(g.data_ptr(), coords.data_ptr(), gx.data_ptr(), N, C, H, W, coords.shape[1], _stream()))
        return gx, None


def point_sample_planar_supported(x, coords):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and coords.dtype == torch.float32
            and coords.dim() == 3 and coords.shape[0] == x.shape[0] and coords.shape[2] == 2 and not coords.requires_grad)


def point_sample_planar(x, coords):
    return PointSamplePlanar.apply(x, coords.contiguous())


def resize_bilinear_rows(x, sizes, out_dtype=torch.float32):
    """x fp32 [B, C, H, W] stored channels-last -> [F.interpolate(x, size=s, mode="bilinear", align_corners=False) as rows [B, h w, C] for s in
    sizes] in out_dtype, ONE launch (pd_resize_bilinear_nhwc_f32); no gradient"""
    import ctypes
    _need_cuda(x, "pd_resize_bilinear_nhwc_f32")
    B, C, H, W = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last) and C % 4 == 0 and 0 < len(sizes) <= 4
    outs = [torch.empty((B, h * w, C), dtype=out_dtype, device=x.device) for h, w in sizes]
    hs = (ctypes.c_int * len(sizes))(*[int(h) for h, _ in sizes])
    ws = (ctypes.c_int * len(sizes))(*[int(w) for _, w in sizes])
    ptrs = (ctypes.c_void_p * len(sizes))(*[o.data_ptr() for o in outs])
    _lib.check(_lib.load().pd_resize_bilinear_nhwc_f32(x.data_ptr(), B, H, W, C, hs, ws, ptrs, len(sizes), _DT[out_dtype], _stream()))
    return outs


def resize_bilinear_rows_supported(x, sizes):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] % 4 == 0 and 0 < len(sizes) <= 4
            and x.is_contiguous(memory_format=torch.channels_last))


def supports_width(C):
    return C % 256 == 0 and C <= 1024


class AddLayerNorm(Function):
    """y = LayerNorm(x + res); optionally also returns y + pos (fp32).  x fp32|bf16 [..., C], res fp32 or None."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, pos):
        ctx.set_materialize_grads(False)
        shape = x.shape
        C = shape[-1]
        x2 = x.reshape(-1, C)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        r2 = None
        if res is not None:
            r2 = res.reshape(-1, C)
            r2 = r2 if r2.is_contiguous() else r2.contiguous()
        p2 = None
        if pos is not None:
            p2 = pos.reshape(-1, C)
            p2 = p2 if p2.is_contiguous() else p2.contiguous()
            assert p2.shape[0] == x2.shape[0]
        keep_z = r2 is not None or x2.dtype != torch.float32
        z, y, _, ypos, mean, rstd = add_ln_fwd(x2, r2, gamma, beta, eps, want_z=keep_z, want_y=True, c_dtype=torch.float32,
                                               pos=p2, pos_div=1, want_ypos=p2 is not None)
        ctx.save_for_backward(z if keep_z else x2, mean, rstd, gamma)
        ctx.has_res, ctx.has_pos, ctx.x_dtype, ctx.shape = res is not None, pos is not None, x.dtype, shape
        if pos is not None:
            return y.view(shape), ypos.view(shape)
        return y.view(shape), None

    @staticmethod
    def backward(ctx, dy, dypos):
        z, mean, rstd, gamma = ctx.saved_tensors
        C = z.shape[1]
        dy = None if dy is None else dy.reshape(-1, C).contiguous()
        dypos = None if dypos is None else dypos.reshape(-1, C).contiguous()
        acc = torch.zeros((2, C), dtype=torch.float32, device=z.device)
        dz, dz_c = add_ln_bwd(z, mean, rstd, gamma, dy=dy, dy2=dypos, dgamma=acc[0], dbeta=acc[1],
                              dz_c_dtype=ctx.x_dtype if ctx.x_dtype != torch.float32 else None)
        dx = (dz_c if dz_c is not None else dz).view(ctx.shape)
        return dx, (dz.view(ctx.shape) if ctx.has_res else None), acc[0], acc[1], None, (dypos.view(ctx.shape) if ctx.has_pos else None)


def add_layer_norm(x, res, norm: torch.nn.LayerNorm, pos=None):
    """LayerNorm(x + res) with `norm`'s parameters through the fused kernel; -> (y, y + pos | None)."""
    return AddLayerNorm.apply(x, res, norm.weight, norm.bias, norm.eps, pos)


class UpsampleAdd(Function):
    """cur + F.interpolate(lo, size=cur.shape[-2:], mode="bilinear", align_corners=False) on fp32 channels-last maps
    (exact 2x only: the backward is the gather-form kernel)."""

    @staticmethod
    def forward(ctx, lo, cur):
        B, C, h, w = lo.shape
        H, W = cur.shape[-2:]
        # lo as one level of a [B, tokens, C] tensor (the encoder's output): dense NHWC images a batch stride apart — read in place
        if lo.stride()[1:] == (1, w * C, C) and lo.stride(0) >= h * w * C and lo.stride(0) % 4 == 0:
            lo_c, lo_bs = lo, lo.stride(0)
        else:
            lo_c, lo_bs = lo.contiguous(memory_format=torch.channels_last), 0
        cur_c = cur.contiguous(memory_format=torch.channels_last)
        y = torch.empty_like(cur_c, memory_format=torch.channels_last)
        if C == 256:            # + the pixel maxima for the fp16 two-plane 3 x 3 convolution that reads y (functions/amax_cache.py)
            am = torch.empty(B * H * W, dtype=torch.float32, device=y.device)
            _lib.check(_lib.load().pd_upsample_add_amax_nhwc_f32(lo_c.data_ptr(), lo_bs, cur_c.data_ptr(), y.data_ptr(), am.data_ptr(), B, h, w, H, W, C,
                                                                 _stream()))
            ctx.y_am = am
        else:
            _lib.check(_lib.load().pd_upsample_add_nhwc_f32(lo_c.data_ptr(), lo_bs, cur_c.data_ptr(), y.data_ptr(), B, h, w, H, W, C, _stream()))
            ctx.y_am = None
        ctx.dims = (B, C, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, h, w = ctx.dims
        dy = dy.contiguous(memory_format=torch.channels_last)
        dlo = torch.empty((B, C, h, w), dtype=torch.float32, device=dy.device, memory_format=torch.channels_last)
        _lib.check(_lib.load().pd_upsample2x_bwd_nhwc_f32(dy.data_ptr(), dlo.data_ptr(), B, h, w, C, _stream()))
        return dlo, dy


def upsample_add_supported(lo, cur):
    return (lo.is_cuda and lo.dtype == torch.float32 and cur.dtype == torch.float32 and lo.shape[1] % 4 == 0
            and cur.shape[-2] == 2 * lo.shape[-2] and cur.shape[-1] == 2 * lo.shape[-1] and not torch.is_autocast_enabled("cuda"))


def upsample_add(lo, cur):
    y = UpsampleAdd.apply(lo, cur)
    am = getattr(y.grad_fn, "y_am", None) if y.grad_fn is not None else None
    if am is not None:
        from . import amax_cache
        amax_cache.put(y, am)
    return y
