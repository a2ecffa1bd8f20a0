"""where the R50 backbone's convolution time goes at 2 x 1024^2: every distinct conv of the network (torchvision-style
R50: stride on the 3x3), bf16 channels_last, timed through torch (MIOpen) forward / dgrad / wgrad, with the HBM and MFMA
floors beside it.  python tools/bench_r50_convs.py [--size 1024] [--batch 2]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
import torch.nn.functional as F
sys.path.insert(0, ROOT)
from partdistillation_amd.functions import conv_bf16 as OC

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--batch", type=int, default=2)
a = ap.parse_args()
B, S = a.batch, a.size
convs = [("stem7x7", 3, 64, 7, 2, S, 1)]
h = S // 4
cin = 64
for stage, (mid, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]):
    out = mid * 4
    for b in range(blocks):
        s = stride if b == 0 else 1
        name = f"res{stage + 2}.{b}"
        convs.append((name + ".c1 1x1", cin, mid, 1, 1, h, 1))
        convs.append((name + ".c2 3x3", mid, mid, 3, s, h, 1))
        if b == 0:
            convs.append((name + ".sc 1x1", cin, out, 1, s, h, 1))
        h = h // s
        convs.append((name + ".c3 1x1", mid, out, 1, 1, h, 1))
        cin = out
# merge identical configs
uniq = {}
for name, ci, co, k, s, hh, n in convs:
    key = (ci, co, k, s, hh)
    if key in uniq:
        uniq[key][1] += 1
    else:
        uniq[key] = [name, 1]


def timeit(fn, n=6):
    """device time per call (sum of the kernels' durations; the host cannot issue these small launches back to back)"""
    from torch.profiler import profile, ProfilerActivity
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return sum(e.device_time_total for e in prof.key_averages()) / n       # us


tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0, "floor": 0.0}
tot.update({"own_fwd": 0.0, "own_dgrad": 0.0, "own_wgrad": 0.0})
print(f"{'conv':18s} {'cfg':28s} cnt   fwd_us dgrad_us wgrad_us | own_fwd own_dgrad own_wgrad | GF(one dir)  MB(fwd)  floor_us(one dir: max(hbm@6TB/s, mfma@1.5PF))")
for (ci, co, k, s, hh), (name, cnt) in uniq.items():
    x = torch.randn(B, ci, hh, hh, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(co, ci, k, k, device="cuda", dtype=torch.bfloat16) * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    pad = k // 2
    y = F.conv2d(x, w, None, s, pad)
    gy = torch.randn_like(y)
    t_f = timeit(lambda: F.conv2d(x, w, None, s, pad))
    cb = torch.ops.aten.convolution_backward
    xd, wd = x.detach(), w.detach()
    t_d = timeit(lambda: cb(gy, xd, wd, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False])) if ci > 3 else 0.0
    t_w = timeit(lambda: cb(gy, xd, wd, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False]))
    ho = y.shape[2]
    o_f = o_d = o_w = 0.0
    if ci % 8 == 0:
        o_w = timeit(lambda: OC.conv_wgrad(gy, xd, k, s, pad))
    if OC.supported(xd, wd, s, pad):
        sc, bi = torch.rand(co, device="cuda") + 0.5, torch.randn(co, device="cuda")
        resid = torch.randn_like(y)
        o_f = timeit(lambda: OC.conv_fwd(xd, wd, sc, bi, resid, True, s, pad))
        wt = OC.transposed_filter(wd)
        o_d = timeit(lambda: OC.conv_dgrad(gy, wt, xd.shape, k, s, pad, addend=xd))
    gf = 2.0 * B * ho * ho * co * ci * k * k / 1e9
    mb = (x.numel() + y.numel() + w.numel()) * 2 / 1e6
    floor = max(mb / 6.0e6 * 1e6, gf / 1.5e6 * 1e6)
    print(f"{name:18s} {f'{ci}->{co} k{k} s{s} @{hh}':28s} {cnt:3d} {t_f:8.1f} {t_d:8.1f} {t_w:8.1f} | {o_f:7.1f} {o_d:8.1f} {o_w:8.1f} | {gf:9.1f} {mb:8.1f} {floor:8.1f}")
    tot["fwd"] += cnt * t_f; tot["dgrad"] += cnt * t_d; tot["wgrad"] += cnt * t_w; tot["floor"] += cnt * floor
    tot["own_fwd"] += cnt * (o_f or t_f); tot["own_dgrad"] += cnt * (o_d or t_d); tot["own_wgrad"] += cnt * (o_w or t_w)
print("totals (us per step):", {k: round(v, 1) for k, v in tot.items()}, "sum", round(tot["fwd"] + tot["dgrad"] + tot["wgrad"], 1),
      "floor x3 dirs", round(3 * tot["floor"], 1))
