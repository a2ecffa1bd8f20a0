"""Fused query-side decoder layer kernels (include/pd_declayer.h) against the unfused kernels they replace (csrc/smallgemm.hip,
csrc/rowwise.hip — themselves checked against the reference goldens by tests/test_product_gpu.py) and against plain torch fp32
on the same bf16-rounded operands.  Reference: mask2former_transformer_decoder.py:44-54, 102-114, 167-171, 198-204, 449-459."""
import pytest
import torch

pytestmark = pytest.mark.gpu

C, FF = 256, 2048
bf = torch.bfloat16


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda", 0)


def _close(a, b, what, ulps=2.0, mean_tol=2e-3):
    """a vs b where both went through bf16 roundings whose last bit the accumulation order may flip: every element within `ulps`
    bf16 steps of the tensor's scale, and the MEAN deviation far below one step (a systematic error would show there)"""
    a, b = a.float(), b.float()
    assert a.shape == b.shape, what
    scale = b.abs().max().item() + 1e-30
    d = (a - b).abs()
    assert torch.isfinite(a).all(), what
    assert d.max().item() <= ulps * 2.0 ** -8 * scale, f"{what}: max dev {d.max().item():.3e} of scale {scale:.3e}"
    assert d.mean().item() <= mean_tol * b.abs().mean().item() + 1e-12, f"{what}: mean dev {d.mean().item():.3e} vs mean {b.abs().mean().item():.3e}"


def _params(dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)

    def w(n, k, s=None):
        return (torch.randn(n, k, generator=g) * (s or k ** -0.5)).to(dev).to(bf).contiguous()

    def b(n):
        return (torch.randn(n, generator=g) * 0.1).to(dev).to(bf).contiguous()

    def ln():
        return (1 + 0.2 * torch.randn(C, generator=g)).to(dev).contiguous(), (0.1 * torch.randn(C, generator=g)).to(dev).contiguous()

    p = {"co_w": w(C, C), "co_b": b(C), "cn": ln(), "si_w": w(3 * C, C), "si_b": b(3 * C), "so_w": w(C, C), "so_b": b(C), "sn": ln(),
         "w1": w(FF, C), "b1": b(FF), "w2": w(C, FF), "b2": b(C), "fn": ln(), "dn": ln(),
         "mlp": [w(C, C), b(C), w(C, C), b(C), w(C, C), b(C)], "cq_w": w(C, C), "cq_b": b(C)}
    return p, g


@pytest.mark.parametrize("split", [False, True])          # True: the FFN as two launches (PD_DEC_SPLIT workgroups per row block, then one)
@pytest.mark.parametrize("Q,B", [(100, 2), (100, 3), (7, 1), (16, 1)])
def test_forward_kernels_vs_unfused(Q, B, split):
    dev = _dev()
    from partdistillation_amd.functions import declayer as dl, rowwise as rw, smallgemm as sg
    R, eps = Q * B, 1e-5
    p, g = _params(dev, 1 + Q + B)
    o = torch.randn(R, C, generator=g).to(dev).to(bf)
    tgt = torch.randn(R, C, generator=g).to(dev)
    qpos = torch.randn(Q, C, generator=g).to(dev)
    # ---- A: unfused
    z_r, y_r, yc_r, yp_r, m_r, r_r = rw.add_ln_fwd(sg.linear(o, p["co_w"], p["co_b"]), tgt, p["cn"][0], p["cn"][1], eps, c_dtype=bf, want_yc=True, pos=qpos,
                                                    pos_div=B, want_ypos=True)
    q_r = sg.linear(yp_r, p["si_w"][:C], p["si_b"][:C])
    k_r = sg.linear(yp_r, p["si_w"][C:2 * C], p["si_b"][C:2 * C])
    v_r = sg.linear(yc_r, p["si_w"][2 * C:], p["si_b"][2 * C:])
    pk = dict(zip(("co_w", "si_w", "so_w", "w1", "w2", "m0", "m1", "m2", "cq_w"),
                  dl.pack([p["co_w"], p["si_w"], p["so_w"], p["w1"], p["w2"], p["mlp"][0], p["mlp"][2], p["mlp"][4], p["cq_w"]])))
    z, st, y, yc, yp, q, k, v = dl.fwd_a(o, tgt, qpos, B, pk["co_w"], p["co_b"], p["cn"][0], p["cn"][1], eps, pk["si_w"], p["si_b"])
    _close(z, z_r, "z1"); _close(y, y_r, "y1", ulps=4); _close(yc, yc_r, "y1_c", ulps=4); _close(yp, yp_r, "y1pos_c", ulps=4)
    _close(st[0], m_r, "mean1"); _close(st[1], r_r, "rstd1", ulps=4)
    _close(q, q_r, "q", ulps=6); _close(k, k_r, "k", ulps=6); _close(v, v_r, "v", ulps=6)
    # torch fp32 on the same operands (not via the unfused kernels)
    x = (o.float() @ p["co_w"].float().t() + p["co_b"].float()).to(bf).float() + tgt
    _close(z, x, "z1 vs torch")
    y_t = torch.nn.functional.layer_norm(x, (C,), p["cn"][0], p["cn"][1], eps)
    _close(y, y_t, "y1 vs torch", ulps=4)
    # ---- B: unfused, fed with the SAME inputs (a synthetic attention output)
    o_s = torch.randn(R, C, generator=g).to(dev).to(bf)
    z2_r, y2_r, y2c_r, _, m2_r, r2_r = rw.add_ln_fwd(sg.linear(o_s, p["so_w"], p["so_b"]), y_r, p["sn"][0], p["sn"][1], eps, c_dtype=bf, want_yc=True)
    h_r = sg.linear(y2c_r, p["w1"], p["b1"], True)
    z3_r, y3_r, _, y3p_r, m3_r, r3_r = rw.add_ln_fwd(sg.linear(h_r, p["w2"], p["b2"]), y2_r, p["fn"][0], p["fn"][1], eps, c_dtype=bf, pos=qpos, pos_div=B,
                                                    want_ypos=True)
    d_t = torch.nn.functional.layer_norm(y3_r, (C,), p["dn"][0], p["dn"][1], eps)
    e = d_t.to(bf)
    for j in range(3):
        e = sg.linear(e, p["mlp"][2 * j], p["mlp"][2 * j + 1], j < 2)
    qc_r = sg.linear(y3p_r, p["cq_w"], p["cq_b"])
    lay = (pk["so_w"], p["so_b"], p["sn"][0], p["sn"][1], pk["w1"], p["b1"], pk["w2"], p["b2"], p["fn"][0], p["fn"][1])
    mlp_p = [pk["m0"], p["mlp"][1], pk["m1"], p["mlp"][3], pk["m2"], p["mlp"][5]]
    ws = dl.workspace(R, dev) if split else None
    dec_out = torch.empty(R, C, device=dev)
    r = dl.fwd_b(o_s, y_r, qpos, B, lay, p["dn"][0], p["dn"][1], mlp_p, (pk["cq_w"], p["cq_b"]), eps, dec_out, ws)
    _close(r["z2"], z2_r, "z2"); _close(r["y2_c"], y2c_r, "y2_c", ulps=4); _close(r["stats2"][0], m2_r, "mean2"); _close(r["stats2"][1], r2_r, "rstd2", ulps=4)
    _close(r["h"], h_r, "h", ulps=6); _close(r["z3"], z3_r, "z3", ulps=6); _close(r["y3"], y3_r, "y3", ulps=8)
    _close(r["ypos_c"], y3p_r, "y3pos_c", ulps=8); _close(dec_out, d_t, "dec_out", ulps=8)
    _close(r["ef"], e.view(Q, B, C).transpose(0, 1), "ef", ulps=16, mean_tol=1e-2)
    _close(r["qc"], qc_r, "qc", ulps=12, mean_tol=5e-3)
    # the head in front of the first layer: y3 = the input rows
    dec0 = torch.empty(R, C, device=dev)
    r0 = dl.fwd_b(None, tgt, qpos, B, None, p["dn"][0], p["dn"][1], mlp_p, (pk["cq_w"], p["cq_b"]), eps, dec0)
    d0 = torch.nn.functional.layer_norm(tgt, (C,), p["dn"][0], p["dn"][1], eps)
    _close(dec0, d0, "dec_out 0")
    e = d0.to(bf)
    for j in range(3):
        e = sg.linear(e, p["mlp"][2 * j], p["mlp"][2 * j + 1], j < 2)
    _close(r0["ef"], e.view(Q, B, C).transpose(0, 1), "ef 0", ulps=8, mean_tol=5e-3)
    tp0 = (tgt + qpos.repeat_interleave(B, 0)).to(bf)
    _close(r0["ypos_c"], tp0, "ypos 0")
    _close(r0["qc"], sg.linear(tp0, p["cq_w"], p["cq_b"]), "qc 0", ulps=4)
    # the last layer: no head MLP
    dec9 = torch.empty(R, C, device=dev)
    r9 = dl.fwd_b(o_s, y_r, qpos, B, lay, p["dn"][0], p["dn"][1], None, None, eps, dec9, ws)
    assert torch.equal(dec9, dec_out) and torch.equal(r9["y3"], r["y3"]) and torch.equal(r9["h"], r["h"])


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("Q,B,last", [(100, 2, False), (100, 2, True), (100, 3, False), (7, 1, False)])
def test_backward_kernels_vs_unfused(Q, B, last, split):
    dev = _dev()
    from partdistillation_amd.functions import declayer as dl, igemm, rowwise as rw, smallgemm as sg
    R, eps = Q * B, 1e-5
    p, g = _params(dev, 11 + Q + B)

    def rnd(*s, scale=1.0):
        return (torch.randn(*s, generator=g) * scale).to(dev)

    def stats():
        return torch.stack([rnd(R, scale=0.1), 1 + rnd(R, scale=0.2).abs()]).contiguous()

    z3, z2, z1, y3 = rnd(R, C), rnd(R, C), rnd(R, C), rnd(R, C)
    st3, st2, st1, hst = stats(), stats(), stats(), stats()
    h = rnd(R, FF).clamp_min(0).to(bf)
    d_out, d_res = rnd(R, C, scale=0.05), (None if last else rnd(R, C, scale=0.05))
    dqc = None if last else rnd(R, C, scale=0.05).to(bf)
    cqT, w2T, w1T, soT, siT, coT = dl.pack([p["cq_w"], p["w2"], p["w1"], p["so_w"], p["si_w"], p["co_w"]], transpose=True)
    # the transposing pack equals the plain pack of the explicitly transposed weight
    tr = igemm.transposed([p["w2"], p["si_w"]])
    chk = dl.pack([t.contiguous() for t in tr])
    assert torch.equal(chk[0], w2T) and torch.equal(chk[1], siT)

    def accs():
        return {k_: torch.zeros(n, device=dev) for k_, n in (("dn", 2 * C), ("g3", 2 * C), ("b3", C), ("g2", 2 * C), ("b2", C), ("g1", 2 * C), ("b1", C), ("pos", Q * C))}

    # ---- unfused B
    A = accs()
    d_pos_c = None if last else sg.dgrad(dqc, p["cq_w"])
    dzh, _ = rw.add_ln_bwd(y3, hst[0], hst[1], p["dn"][0], dy=d_out, dgamma=A["dn"][:C], dbeta=A["dn"][C:])
    dz3_r, dz3c_r = rw.add_ln_bwd(z3, st3[0], st3[1], p["fn"][0], dy=dzh, dy2=d_res, dypos_c=d_pos_c, dz_c_dtype=bf, dgamma=A["g3"][:C], dbeta=A["g3"][C:],
                                  dbias=A["b3"], dpos_acc=A["pos"].view(Q, C) if d_pos_c is not None else None, pos_div=B)
    dh_r = sg.dgrad(dz3c_r, p["w2"], relu_ref=h)
    dx_r = sg.dgrad(dh_r, p["w1"])
    dz2_r, dz2c_r = rw.add_ln_bwd(z2, st2[0], st2[1], p["sn"][0], dy=dz3_r.clone(), dy_c=dx_r, dz_c_dtype=bf, dgamma=A["g2"][:C], dbeta=A["g2"][C:], dbias=A["b2"])
    do_r = sg.dgrad(dz2c_r, p["so_w"])
    # ---- fused B
    F = accs()
    dz3c, dh, dz2, dz2c, do = dl.bwd_b(dqc, None if last else cqT, d_out, d_res, y3, hst, p["dn"][0], F["dn"], z3, st3, p["fn"][0], F["g3"], F["b3"],
                                       None if last else F["pos"], B, w2T, h, w1T, z2, st2, p["sn"][0], F["g2"], F["b2"], soT, dl.workspace(R, dev) if split else None)
    _close(dz3c, dz3c_r, "dz3_c", ulps=3); _close(dh, dh_r, "dh", ulps=4); _close(dz2, dz2_r, "dz2", ulps=4); _close(dz2c, dz2c_r, "dz2_c", ulps=4)
    _close(do, do_r, "d_o", ulps=6, mean_tol=5e-3)
    for k_ in ("dn", "g3", "b3", "g2", "b2", "pos"):
        _close(F[k_], A[k_], "acc " + k_, ulps=4, mean_tol=5e-3)
    # ---- A
    dq, dk, dv = (rnd(R, C, scale=0.05).to(bf) for _ in range(3))
    d_tp = sg.dgrad(dq, p["si_w"][:C])
    sg.dgrad(dk, p["si_w"][C:2 * C], out=d_tp, accumulate=True)
    d_tc = sg.dgrad(dv, p["si_w"][2 * C:])
    dz1_r, dz1c_r = rw.add_ln_bwd(z1, st1[0], st1[1], p["cn"][0], dy=dz2_r.clone(), dy_c=d_tc, dypos_c=d_tp, dz_c_dtype=bf, dgamma=A["g1"][:C], dbeta=A["g1"][C:],
                                  dbias=A["b1"], dpos_acc=A["pos"].view(Q, C), pos_div=B)
    doc_r = sg.dgrad(dz1c_r, p["co_w"])
    if last:
        F["pos"].zero_(); A["pos"].zero_()
        rw.add_ln_bwd(z1, st1[0], st1[1], p["cn"][0], dy=dz2_r.clone(), dy_c=d_tc, dypos_c=d_tp, dz_c_dtype=bf, dpos_acc=A["pos"].view(Q, C), pos_div=B)
    dz1, dz1c, doc = dl.bwd_a(dq, dk, dv, siT, dz2_r, z1, st1, p["cn"][0], F["g1"], F["b1"], F["pos"], B, coT)
    _close(dz1, dz1_r, "dz1", ulps=4); _close(dz1c, dz1c_r, "dz1_c", ulps=4); _close(doc, doc_r, "d_o cross", ulps=6, mean_tol=5e-3)
    for k_ in ("g1", "b1", "pos"):
        _close(F[k_], A[k_], "acc " + k_, ulps=4, mean_tol=5e-3)
    # torch fp32 on the same operands: the first product and LayerNorm backward of the B kernel
    xh = (z3 - st3[0][:, None]) * st3[1][:, None]
    xhh = (y3 - hst[0][:, None]) * hst[1][:, None]
    gh = d_out * p["dn"][0]
    dzh_t = hst[1][:, None] * (gh - gh.mean(1, keepdim=True) - xhh * (gh * xhh).mean(1, keepdim=True))
    tt = dzh_t + (d_res if d_res is not None else 0) + (0 if last else (dqc.float() @ p["cq_w"].float()).to(bf).float())
    gg = tt * p["fn"][0]
    dz3_t = st3[1][:, None] * (gg - gg.mean(1, keepdim=True) - xh * (gg * xh).mean(1, keepdim=True))
    _close(dz3c, dz3_t, "dz3 vs torch", ulps=3)
