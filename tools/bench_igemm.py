"""pd_igemm_bf16 against the library on the shapes it is meant for (development tool, GPU box):
    python tools/bench_igemm.py [r50] [swin] [sweep]
GPU time per launch: own kernel through pd_igemm_bf16_time (events around back-to-back launches issued from C++), library calls behind a
blocker kernel so that the host's issue time does not enter (events around N calls the host has finished enqueuing before the GPU reaches them)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from partdistillation_amd import lib
L = lib.load()
from partdistillation_amd.functions import igemm as ig

torch.backends.cudnn.benchmark = True
dev = "cuda"
_BLOCK = None


def gpu_time(fn, n=20):
    """us per call of a torch / library op, measured on the device behind a blocker"""
    global _BLOCK
    if _BLOCK is None:
        _BLOCK = torch.randn((8192, 8192), device=dev).to(torch.bfloat16)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        torch.mm(_BLOCK, _BLOCK)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def own_time(desc, iters=30, **knobs):
    for k, v in knobs.items():
        lib.check(L.pd_debug_set(k.encode(), int(v)))
    need = int(L.pd_igemm_bf16_workspace_bytes(ctypes.byref(desc)))
    ws = ig.workspace(torch.device(dev), need) if need > 0 else None
    us = ctypes.c_float(0)
    lib.check(L.pd_igemm_bf16_time(ctypes.byref(desc), ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, iters,
                                   ctypes.byref(us), lib.current_stream()))
    for k in knobs:
        L.pd_debug_set(k.encode(), 0)
    return us.value


def desc_of(src, w, out, **kw):
    p = lambda t: t.data_ptr() if t is not None else None
    g = lambda k, d=None: kw.get(k, d)
    return ig.PdIgemm(p(src), p(w), p(g("scale")), p(g("bias")), p(g("res")), None, p(g("gate")), p(out), p(g("out_pre")), kw["batch"], kw["hs"], kw["ws"],
                      kw["cs"], kw["ho"], kw["wo"], kw["n"], g("k", 1), g("stride", 1), g("pad", 0), int(g("dgrad", False)), g("act", 0),
                      g("gate_mode", 0) if g("gate") is not None else 0, g("res_mode", 0))


def r50_layers(S=1024):
    out, h, cin = [], S // 4, 64
    for stage, (mid, blocks) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3))):
        cout = mid * 4
        for b in range(blocks):
            s = 2 if (b == 0 and stage > 0) else 1
            out.append((f"res{stage + 2}.{b}.conv1", cin, mid, 1, 1, h))
            out.append((f"res{stage + 2}.{b}.conv2", mid, mid, 3, s, h))
            if b == 0:
                out.append((f"res{stage + 2}.{b}.shortcut", cin, cout, 1, s, h))
            out.append((f"res{stage + 2}.{b}.conv3", mid, cout, 1, 1, h // s))
            cin, h = cout, h // s
    return out


VARIANTS = [("warmup", {}), ("auto", {}), ("128/1", dict(ig_bn=128, ig_nst=1, ig_splits=1)), ("128/2", dict(ig_bn=128, ig_nst=2, ig_splits=1)),
            ("128/3", dict(ig_bn=128, ig_nst=3, ig_splits=1)), ("64/1", dict(ig_bn=64, ig_nst=1, ig_splits=1)), ("64/2", dict(ig_bn=64, ig_nst=2, ig_splits=1)),
            ("64/3", dict(ig_bn=64, ig_nst=3, ig_splits=1)), ("64/3/s4", dict(ig_bn=64, ig_nst=3, ig_splits=4))]


def r50(sweep):
    B = 2
    tot = {}
    seen = {}
    names = [v[0] for v in (VARIANTS if sweep else VARIANTS[:2])]
    print(f"{'layer':18s} {'ci':>5s} {'co':>5s} k s {'H':>4s} | fwd: " + " ".join(f"{n:>8s}" for n in names) + "      lib | dgrad: " + " ".join(f"{n:>8s}" for n in names) + "      lib   (us)")
    for name, ci, co, k, s, H in r50_layers():
        key = (ci, co, k, s, H)
        if key not in seen:
            pad = k // 2
            x = torch.randn((B, H, H, ci), device=dev).to(torch.bfloat16)
            w = (torch.randn((co, k, k, ci), device=dev) * (k * k * ci) ** -0.5).to(torch.bfloat16)
            Ho = (H + 2 * pad - k) // s + 1
            res = torch.randn((B, Ho, Ho, co), device=dev).to(torch.bfloat16)
            y = torch.empty_like(res)
            scale, bias = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev)
            xn, wn = x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2)
            d = desc_of(x, w, y, batch=B, hs=H, ws=H, cs=ci, ho=Ho, wo=Ho, n=co, k=k, stride=s, pad=pad, scale=scale, bias=bias, res=res, act=ig.ACT_RELU)
            f = [own_time(d, **kn) for _, kn in (VARIANTS if sweep else VARIANTS[:2])]
            t_lib = gpu_time(lambda: torch.ops.aten.convolution(xn, wn, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1))
            dz = torch.randn((B, Ho, Ho, co), device=dev).to(torch.bfloat16)
            wt = w.permute(3, 1, 2, 0).contiguous()
            dzn = dz.permute(0, 3, 1, 2)
            dx = torch.empty_like(x)
            if k == 1 and s == 2:
                dc = torch.empty((B, Ho, Ho, ci), device=dev, dtype=torch.bfloat16)
                dd = desc_of(dz, wt, dc, batch=B, hs=Ho, ws=Ho, cs=co, ho=Ho, wo=Ho, n=ci)
            else:
                dd = desc_of(dz, wt, dx, batch=B, hs=Ho, ws=Ho, cs=co, ho=H, wo=H, n=ci, k=k, stride=s, pad=pad, dgrad=k > 1, res=x, gate=x, gate_mode=ig.GATE_RELU)
            g = [own_time(dd, **kn) for _, kn in (VARIANTS if sweep else VARIANTS[:2])]
            t_ld = gpu_time(lambda: torch.ops.aten.convolution_backward(dzn, xn, wn, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False]))
            seen[key] = (f, t_lib, g, t_ld)
        f, t_lib, g, t_ld = seen[key]
        for i, n in enumerate(names):
            tot[("f", n)] = tot.get(("f", n), 0) + f[i]
            tot[("g", n)] = tot.get(("g", n), 0) + g[i]
        tot["bestf"] = tot.get("bestf", 0) + min(f)
        tot["bestg"] = tot.get("bestg", 0) + min(g)
        tot["lf"] = tot.get("lf", 0) + t_lib
        tot["lg"] = tot.get("lg", 0) + t_ld
        print(f"{name:18s} {ci:5d} {co:5d} {k} {s} {H:4d} |      " + " ".join(f"{v:8.1f}" for v in f) + f" {t_lib:8.1f} |        " + " ".join(f"{v:8.1f}" for v in g) + f" {t_ld:8.1f}")
    print("R50 body totals (ms): fwd " + ", ".join(f"{n} {tot[('f', n)] / 1e3:.3f}" for n in names) + f", best-of {tot['bestf'] / 1e3:.3f}, library (no epilogue) {tot['lf'] / 1e3:.3f}")
    print("                      dgrad " + ", ".join(f"{n} {tot[('g', n)] / 1e3:.3f}" for n in names) + f", best-of {tot['bestg'] / 1e3:.3f}, library (no epilogue) {tot['lg'] / 1e3:.3f}")


def swin(sweep):
    names = [v[0] for v in (VARIANTS if sweep else VARIANTS[:2])]
    print(f"{'Linear':16s} {'tokens':>7s} {'K':>5s} {'N':>5s} | " + " ".join(f"{n:>8s}" for n in names) + "    addmm | TF/s best / lib")
    shapes = []
    for nm, C0, g in (("swinB", 128, [264, 132, 72, 36]), ("swinL", 192, [324, 168, 84, 48])):
        for st in range(4):
            C, T = C0 << st, 2 * g[st] * g[st]
            shapes += [(f"{nm} s{st} qkv", T, C, 3 * C), (f"{nm} s{st} proj", T, C, C), (f"{nm} s{st} fc1", T, C, 4 * C), (f"{nm} s{st} fc2", T, 4 * C, C)]
    shapes += [("dec kv 16384", 32768, 256, 512), ("dec kv 4096", 8192, 256, 512), ("dec kv 1024", 2048, 256, 512)]
    tb = tl = 0.0
    for nm, M, K, N in shapes:
        x = torch.randn((M, K), device=dev).to(torch.bfloat16)
        w = (torch.randn((N, K), device=dev) * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev).to(torch.bfloat16)
        bf = b.float()
        y = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
        d = desc_of(x, w, y, batch=1, hs=M, ws=1, cs=K, ho=M, wo=1, n=N, bias=bf)
        f = [own_time(d, **kn) for _, kn in (VARIANTS if sweep else VARIANTS[:2])]
        t_lib = gpu_time(lambda: torch.addmm(b, x, w.t()))
        gf = 2.0 * M * K * N / 1e9
        tb += min(f); tl += t_lib
        print(f"{nm:16s} {M:7d} {K:5d} {N:5d} | " + " ".join(f"{v:8.1f}" for v in f) + f" {t_lib:8.1f} | {gf / min(f) * 1e-3:5.0f} / {gf / t_lib * 1e-3:5.0f}")
    print(f"sum of best {tb / 1e3:.3f} ms, library {tl / 1e3:.3f} ms")


if __name__ == "__main__":
    what = set(sys.argv[1:]) or {"r50", "swin"}
    sweep = "sweep" in what
    if "r50" in what or what == {"sweep"}:
        r50(sweep)
    if "swin" in what or what == {"sweep"}:
        swin(sweep)
