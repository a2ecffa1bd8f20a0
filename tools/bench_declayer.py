"""Isolated timing of the fused decoder layer kernels against the unfused launch chains they replace (R = Q * B = 200 rows)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_declayer_gpu import _params, C, FF, bf
from partdistillation_amd.functions import declayer as dl, igemm, rowwise as rw, smallgemm as sg

dev = torch.device("cuda", 0)
Q, B = 100, int(os.environ.get("B", "2"))
R, eps = Q * B, 1e-5
p, g = _params(dev, 3)
o = torch.randn(R, C, generator=g).to(dev).to(bf)
tgt = torch.randn(R, C, generator=g).to(dev)
qpos = torch.randn(Q, C, generator=g).to(dev)
pk = dict(zip(("co_w", "si_w", "so_w", "w1", "w2", "m0", "m1", "m2", "cq_w"),
              dl.pack([p["co_w"], p["si_w"], p["so_w"], p["w1"], p["w2"], p["mlp"][0], p["mlp"][2], p["mlp"][4], p["cq_w"]])))
lay = (pk["so_w"], p["so_b"], p["sn"][0], p["sn"][1], pk["w1"], p["b1"], pk["w2"], p["b2"], p["fn"][0], p["fn"][1])
mlp_p = [pk["m0"], p["mlp"][1], pk["m1"], p["mlp"][3], pk["m2"], p["mlp"][5]]
dec_out = torch.empty(R, C, device=dev)
ws = dl.workspace(R, dev) if os.environ.get("WS", "1") == "1" else None


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def unf_a():
    z, y, yc, yp, m, r = rw.add_ln_fwd(sg.linear(o, p["co_w"], p["co_b"]), tgt, p["cn"][0], p["cn"][1], eps, c_dtype=bf, want_yc=True, pos=qpos, pos_div=B, want_ypos=True)
    return sg.linear_multi([(yp, p["si_w"][:C], p["si_b"][:C]), (yp, p["si_w"][C:2 * C], p["si_b"][C:2 * C]), (yc, p["si_w"][2 * C:], p["si_b"][2 * C:])])


def unf_b():
    z2, y2, y2c, _, m2, r2 = rw.add_ln_fwd(sg.linear(o, p["so_w"], p["so_b"]), tgt, p["sn"][0], p["sn"][1], eps, c_dtype=bf, want_yc=True)
    h = sg.linear(y2c, p["w1"], p["b1"], True)
    z3, y3, _, y3p, m3, r3 = rw.add_ln_fwd(sg.linear(h, p["w2"], p["b2"]), y2, p["fn"][0], p["fn"][1], eps, c_dtype=bf, pos=qpos, pos_div=B, want_ypos=True)
    return y3


print("fwd_a fused   us", round(timeit(lambda: dl.fwd_a(o, tgt, qpos, B, pk["co_w"], p["co_b"], p["cn"][0], p["cn"][1], eps, pk["si_w"], p["si_b"])), 1))
print("fwd_a unfused us (3 launches)", round(timeit(unf_a), 1))
print("fwd_b fused   us", round(timeit(lambda: dl.fwd_b(o, tgt, qpos, B, lay, p["dn"][0], p["dn"][1], mlp_p, (pk["cq_w"], p["cq_b"]), eps, dec_out, ws)), 1))
print("fwd_b fused, no head MLP us", round(timeit(lambda: dl.fwd_b(o, tgt, qpos, B, lay, p["dn"][0], p["dn"][1], None, None, eps, dec_out, ws)), 1))
print("fwd_b head only us", round(timeit(lambda: dl.fwd_b(None, tgt, qpos, B, None, p["dn"][0], p["dn"][1], mlp_p, (pk["cq_w"], p["cq_b"]), eps, dec_out, ws)), 1))
print("fwd_b unfused us (5 launches, without head + q)", round(timeit(unf_b), 1))
cqT, w2T, w1T, soT, siT, coT = dl.pack([p["cq_w"], p["w2"], p["w1"], p["so_w"], p["si_w"], p["co_w"]], transpose=True)
allw = [p["co_w"], p["si_w"], p["so_w"], p["w1"], p["w2"], p["mlp"][0], p["mlp"][2], p["mlp"][4], p["cq_w"]] * 9
print("pack 9 layers forward us", round(timeit(lambda: dl.pack(allw), 50), 1), " backward (transposing) us", round(timeit(lambda: dl.pack(allw, True), 50), 1))
z3, z2, z1, y3 = (torch.randn(R, C, device=dev) for _ in range(4))
st = torch.stack([torch.zeros(R, device=dev), torch.ones(R, device=dev)]).contiguous()
h = torch.randn(R, FF, device=dev).clamp_min(0).to(bf)
d_out, d_res = torch.randn(R, C, device=dev), torch.randn(R, C, device=dev)
dqc = torch.randn(R, C, device=dev).to(bf)
F = {k_: torch.zeros(n, device=dev) for k_, n in (("dn", 2 * C), ("g3", 2 * C), ("b3", C), ("g2", 2 * C), ("b2", C), ("g1", 2 * C), ("b1", C), ("pos", Q * C))}
print("bwd_b fused us", round(timeit(lambda: dl.bwd_b(dqc, cqT, d_out, d_res, y3, st, p["dn"][0], F["dn"], z3, st, p["fn"][0], F["g3"], F["b3"], F["pos"], B, w2T, h, w1T, z2, st,
                                                      p["sn"][0], F["g2"], F["b2"], soT, ws)), 1))
print("bwd_a fused us", round(timeit(lambda: dl.bwd_a(dqc, dqc, dqc, siT, d_res, z1, st, p["cn"][0], F["g1"], F["b1"], F["pos"], B, coT)), 1))


def unf_bwd_b():
    d_pos_c = sg.dgrad(dqc, p["cq_w"])
    dzh, _ = rw.add_ln_bwd(y3, st[0], st[1], p["dn"][0], dy=d_out, dgamma=F["dn"][:C], dbeta=F["dn"][C:])
    dz3, dz3c = rw.add_ln_bwd(z3, st[0], st[1], p["fn"][0], dy=dzh, dy2=d_res, dypos_c=d_pos_c, dz_c_dtype=bf, dgamma=F["g3"][:C], dbeta=F["g3"][C:], dbias=F["b3"],
                              dpos_acc=F["pos"].view(Q, C), pos_div=B)
    dh = sg.dgrad(dz3c, p["w2"], relu_ref=h)
    dx = sg.dgrad(dh, p["w1"])
    dz2, dz2c = rw.add_ln_bwd(z2, st[0], st[1], p["sn"][0], dy=dz3, dy_c=dx, dz_c_dtype=bf, dgamma=F["g2"][:C], dbeta=F["g2"][C:], dbias=F["b2"])
    return sg.dgrad(dz2c, p["so_w"])


def unf_bwd_a():
    d_tp = sg.dgrad(dqc, p["si_w"][:C])
    sg.dgrad(dqc, p["si_w"][C:2 * C], out=d_tp, accumulate=True)
    d_tc = sg.dgrad(dqc, p["si_w"][2 * C:])
    dz1, dz1c = rw.add_ln_bwd(z1, st[0], st[1], p["cn"][0], dy=d_res, dy_c=d_tc, dypos_c=d_tp, dz_c_dtype=bf, dgamma=F["g1"][:C], dbeta=F["g1"][C:], dbias=F["b1"],
                              dpos_acc=F["pos"].view(Q, C), pos_div=B)
    return sg.dgrad(dz1c, p["co_w"])


print("bwd_b unfused us (7 launches)", round(timeit(unf_bwd_b), 1))
print("bwd_a unfused us (5 launches)", round(timeit(unf_bwd_a), 1))
