"""ctypes faces of the fused query-side decoder layer kernels (include/pd_declayer.h, csrc/declayer.hip): plain functions on raw
tensors, no autograd — the building blocks of functions/decoder_core.py's hand-written forward / backward.  GPU only, no fallback.

Reference: transformer_decoder/mask2former_transformer_decoder.py:395-439 (the layer loop), :449-459 (prediction head)."""
import ctypes

import torch

from .. import lib as _lib

C, FF = 256, 2048
bf16 = torch.bfloat16


def _stream():
    return _lib.current_stream()


def _p(t):
    return None if t is None else t.data_ptr()


def supported(C_, ff, cdt):
    return C_ == C and ff == FF and cdt == bf16


class _PackDesc(ctypes.Structure):                                  # PdDecPack (include/pd_declayer.h)
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32), ("transpose", ctypes.c_int32)]


_PK = {}


def pack(weights, transpose=False):
    """[N_i, K_i] bf16 row-major weights -> their packed copies (pd_dec_pack_grouped: block order of the fused kernels' weight stream), of
    the weights themselves (forward kernels) or of their transposes (backward kernels), all in ONE launch into one fresh buffer.
    -> list of flat bf16 tensors.  The descriptor table is cached per list of (address, shape)."""
    from .fused import PinnedRing
    key = (tuple((w.data_ptr(), tuple(w.shape)) for w in weights), bool(transpose))
    L = _lib.load()
    dev = weights[0].device
    hit = _PK.get(key)
    if hit is None:
        offs, total = [], 0
        for w in weights:
            assert w.dtype == bf16 and w.is_contiguous() and w.dim() == 2
            n, k = (w.shape[1], w.shape[0]) if transpose else w.shape
            assert n % 32 == 0 and k % 256 == 0, (n, k)
            offs.append(total)
            total += w.numel()
        tb = int(L.pd_dec_pack_table_bytes(len(weights)))
        hit = _PK[key] = ((_PackDesc * len(weights))(), offs, total, PinnedRing(tb, torch.uint8, pin=True), torch.empty(tb, dtype=torch.uint8, device=dev))
        if len(_PK) > 16:
            _PK.pop(next(iter(_PK)))
    descs, offs, total, ring, tdev = hit
    from .. import cmdbuf
    if cmdbuf.active() is not None:                          # recorded region: its own descriptor array + device table (arena)
        descs = (_PackDesc * len(weights))()
        tdev = torch.empty(tdev.numel(), dtype=torch.uint8, device=dev)
        for w in weights:
            cmdbuf.require_stable(w.data_ptr(), "weight to pack")
    buf = torch.empty(total, dtype=bf16, device=dev)
    base = buf.data_ptr()
    for d, w, o in zip(descs, weights, offs):
        d.src, d.dst, d.rows, d.cols, d.transpose = w.data_ptr(), base + 2 * o, w.shape[0], w.shape[1], int(bool(transpose))
    host = ring.acquire()
    rc = L.pd_dec_pack_grouped(descs, len(weights), host.data_ptr(), tdev.data_ptr(), _stream())
    ring.release()
    _lib.check(rc)
    return [buf[o:o + w.numel()] for w, o in zip(weights, offs)]


def workspace(R, dev):
    """scratch of the two-launch FFN (fp32 slabs + one fp32 row block); no initial state"""
    return torch.empty(int(_lib.load().pd_dec_workspace_bytes(int(R))), dtype=torch.uint8, device=dev)


def _e(shape, dtype, dev):
    return torch.empty(shape, dtype=dtype, device=dev)


def fwd_a(o, res, qpos, pos_div, w_o, b_o, ln_w, ln_b, eps, w_qkv, b_qkv):
    """-> (z, stats [2, R], y, y_c, ypos_c, q, k, v): cross-attention output projection + residual + LayerNorm + the self-attention's
    q / k / v projections of the R = o.shape[0] rows"""
    R, dev = o.shape[0], o.device
    z, y = _e((R, C), torch.float32, dev), _e((R, C), torch.float32, dev)
    stats = _e((2, R), torch.float32, dev)
    y_c, ypos_c, q, k, v = (_e((R, C), bf16, dev) for _ in range(5))
    _lib.check(_lib.load().pd_dec_fwd_a(o.data_ptr(), res.data_ptr(), qpos.data_ptr(), int(pos_div), w_o.data_ptr(), b_o.data_ptr(), ln_w.data_ptr(),
                                        ln_b.data_ptr(), float(eps), w_qkv.data_ptr(), b_qkv.data_ptr(), z.data_ptr(), stats.data_ptr(), y.data_ptr(),
                                        y_c.data_ptr(), ypos_c.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), R, _stream()))
    return z, stats, y, y_c, ypos_c, q, k, v


def fwd_b(o, res, qpos, pos_div, lay, dn_w, dn_b, mlp, q_next, eps, dec_out, ws=None):
    """(all weights PACKED: pack())  lay = (w_o, b_o, ln2_w, ln2_b, w_1, b_1, w_2, b_2, ln3_w, ln3_b) or None (the head in front of the first layer: y3 = res);
    mlp = the six mask-embedding MLP tensors and q_next = (W_q, b_q) of the next layer's cross-attention, or None / None after the last layer.
    dec_out: the [R, C] fp32 row block of the stacked decoder outputs this head writes.
    -> dict(z2, stats2, y2_c, h, z3, stats3, y3 | None ..., ypos_c, hstats, ef | None, qc | None)"""
    R, dev = res.shape[0], res.device
    layer, head = lay is not None, mlp is not None
    out = {}
    if layer:
        out.update(z2=_e((R, C), torch.float32, dev), stats2=_e((2, R), torch.float32, dev), y2_c=_e((R, C), bf16, dev), h=_e((R, FF), bf16, dev),
                   z3=_e((R, C), torch.float32, dev), stats3=_e((2, R), torch.float32, dev), y3=_e((R, C), torch.float32, dev))
    out.update(ypos_c=_e((R, C), bf16, dev), hstats=_e((2, R), torch.float32, dev))
    if head:
        out.update(ef=_e((pos_div, R // pos_div, C), bf16, dev), qc=_e((R, C), bf16, dev))
    g = out.get
    L = lay if layer else (None,) * 10
    M = mlp if head else (None,) * 6
    Qn = q_next if head else (None, None)
    _lib.check(_lib.load().pd_dec_fwd_b(_p(o), res.data_ptr(), qpos.data_ptr(), int(pos_div), *[_p(t) for t in L], dn_w.data_ptr(), dn_b.data_ptr(),
                                        *[_p(t) for t in M], _p(Qn[0]), _p(Qn[1]), float(eps), _p(g("z2")), _p(g("stats2")), _p(g("y2_c")), _p(g("h")),
                                        _p(g("z3")), _p(g("stats3")), _p(g("y3")), out["ypos_c"].data_ptr(), dec_out.data_ptr(), out["hstats"].data_ptr(),
                                        _p(g("ef")), _p(g("qc")), _p(ws), R, (1 if layer else 0) | (2 if head else 0), _stream()))
    return out


def bwd_b(dqc_next, wqT_next, d_out, d_res, y3, hstats, dn_w, dgb_dn, z3, stats3, ln3_w, dgb3, db3, pos_acc, pos_div, w2T, h, w1T, z2, stats2,
          ln2_w, dgb2, db2, woT, ws=None):
    """-> (dz3_c, dh, dz2, dz2_c, d_o); accumulators (dgb_* = [dgamma | dbeta] fp32 [2C], db* [C], pos_acc [Q, C]) are added to"""
    R, dev = z3.shape[0], z3.device
    dz3_c, dz2_c, d_o = (_e((R, C), bf16, dev) for _ in range(3))
    dh = _e((R, FF), bf16, dev)
    dz2 = _e((R, C), torch.float32, dev)
    _lib.check(_lib.load().pd_dec_bwd_b(_p(dqc_next), _p(wqT_next), d_out.data_ptr(), _p(d_res), y3.data_ptr(), hstats.data_ptr(), dn_w.data_ptr(),
                                        dgb_dn.data_ptr(), z3.data_ptr(), stats3.data_ptr(), ln3_w.data_ptr(), dgb3.data_ptr(), db3.data_ptr(), _p(pos_acc),
                                        int(pos_div), w2T.data_ptr(), h.data_ptr(), w1T.data_ptr(), z2.data_ptr(), stats2.data_ptr(), ln2_w.data_ptr(),
                                        dgb2.data_ptr(), db2.data_ptr(), woT.data_ptr(), dz3_c.data_ptr(), dh.data_ptr(), dz2.data_ptr(), dz2_c.data_ptr(),
                                        d_o.data_ptr(), _p(ws), R, _stream()))
    return dz3_c, dh, dz2, dz2_c, d_o


def bwd_a(dq, dk, dv, wqkvT, dz_in, z, stats, ln_w, dgb, db, pos_acc, pos_div, woT):
    """-> (dz1, dz1_c, d_o)"""
    R, dev = z.shape[0], z.device
    dz1 = _e((R, C), torch.float32, dev)
    dz1_c, d_o = _e((R, C), bf16, dev), _e((R, C), bf16, dev)
    _lib.check(_lib.load().pd_dec_bwd_a(dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), wqkvT.data_ptr(), dz_in.data_ptr(), z.data_ptr(), stats.data_ptr(),
                                        ln_w.data_ptr(), dgb.data_ptr(), db.data_ptr(), pos_acc.data_ptr(), int(pos_div), woT.data_ptr(), dz1.data_ptr(),
                                        dz1_c.data_ptr(), d_o.data_ptr(), R, _stream()))
    return dz1, dz1_c, d_o
