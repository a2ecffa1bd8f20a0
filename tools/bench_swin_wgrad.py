"""dW = dY^T X of the Swin Linears (rows = the stage's tokens): the split-rows skinny kernel (pd_sgemm_wgrad_split_bf16), the transpose-read
filter-gradient kernel as a single launch (pd_conv_bf16_wgrad, a 1 x 1 "convolution" over the tokens) and the library (torch.mm(dy.t(), x)).
GPU time per call behind a blocker (development tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from partdistillation_amd import lib
L = lib.load()
from partdistillation_amd.functions import conv_bf16, igemm as ig, smallgemm as sg
import ctypes


def own_time(dy, x, iters=20, **knobs):
    for k, v in knobs.items():
        lib.check(L.pd_debug_set(k.encode(), int(v)))
    M, N = dy.shape
    K = x.shape[1]
    dw = torch.empty((N, K), dtype=torch.bfloat16, device=dy.device)
    db = torch.zeros(N, dtype=torch.float32, device=dy.device)
    d = ig.PdWgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), None, M, N, K, dy.stride(0), x.stride(0), K)
    need = int(L.pd_wgrad_bf16_workspace_bytes(ctypes.byref(d)))
    ws = ig.workspace(dy.device, need) if need > 0 else None
    us = ctypes.c_float(0)
    lib.check(L.pd_wgrad_bf16_time(ctypes.byref(d), ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, iters, ctypes.byref(us),
                                   lib.current_stream()))
    for k in knobs:
        L.pd_debug_set(k.encode(), 0)
    return us.value, dw

dev = "cuda"
_B = torch.randn((8192, 8192), device=dev).to(torch.bfloat16)


def gpu_time(fn, n=15):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        torch.mm(_B, _B)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def tr_single(dy, x):
    M, N = dy.shape
    K = x.shape[1]
    return conv_bf16.conv_wgrad(dy.view(1, M, 1, N).permute(0, 3, 1, 2), x.view(1, M, 1, K).permute(0, 3, 1, 2), 1)


if __name__ == "__main__":
    shapes = []
    for nm, C0, g in (("swinB", 128, [264, 132, 72, 36]), ("swinL", 192, [324, 168, 84, 48])):
        for st in range(4):
            C, T = C0 << st, 2 * g[st] * g[st]
            shapes += [(f"{nm} s{st} qkv", T, C, 3 * C), (f"{nm} s{st} proj", T, C, C), (f"{nm} s{st} fc1", T, C, 4 * C), (f"{nm} s{st} fc2", T, 4 * C, C)]
    VAR = [("auto", {}), ("2st/s2", dict(wg_nst=2, wg_splits=2)), ("2st/s4", dict(wg_nst=2, wg_splits=4)),
           ("2st/s8", dict(wg_nst=2, wg_splits=8)), ("2st/s16", dict(wg_nst=2, wg_splits=16)),
           ("1st/s16", dict(wg_nst=1, wg_splits=16)), ("2st/s32", dict(wg_nst=2, wg_splits=32)), ("2st/s64", dict(wg_nst=2, wg_splits=64)),
           ("1st/s64", dict(wg_nst=1, wg_splits=64)), ("2st/128", dict(wg_nst=2, wg_splits=128)), ("2st/256", dict(wg_nst=2, wg_splits=256))]
    print(f"{'Linear':16s} {'tokens':>7s} {'K(in)':>6s} {'N(out)':>6s} |    split   tr-conv   library | own: " + " ".join(f"{n:>8s}" for n, _ in VAR) + "  (us)   TF/s best own")
    tot = [0.0, 0.0, 0.0]
    tot_own = 0.0
    for nm, M, K, N in shapes:
        x = torch.randn((M, K), device=dev).to(torch.bfloat16)
        dy = torch.randn((M, N), device=dev).to(torch.bfloat16)
        ts = []
        try:
            ts.append(gpu_time(lambda: sg.wgrad_split(dy, x, True)))
        except Exception as e:
            ts.append(float("nan"))
        try:
            ref = torch.mm(dy.float().t(), x.float())
            got = tr_single(dy, x).reshape(N, K).float()
            err = ((got - ref).abs().max() / ref.abs().max()).item()
            ts.append(gpu_time(lambda: tr_single(dy, x)))
        except Exception as e:
            err = float("nan"); ts.append(float("nan"))
        ts.append(gpu_time(lambda: torch.mm(dy.t(), x)))
        for i in range(3):
            tot[i] += ts[i]
        gf = 2.0 * M * K * N / 1e9
        own = []
        for _, kn in VAR:
            t_, dw_ = own_time(dy, x, **kn)
            own.append(t_)
        oerr = ((dw_.float() - ref).abs().max() / ref.abs().max()).item()
        tot_own += min(own)
        print(f"{nm:16s} {M:7d} {K:6d} {N:6d} | {ts[0]:8.1f} {ts[1]:8.1f} {ts[2]:8.1f} |      " + " ".join(f"{v:8.1f}" for v in own) + f"   {gf / min(own) * 1e-3:6.0f}   (own err {oerr:.1e})")
    print("sums (ms): split / tr-conv / library", [round(t / 1e3, 3) for t in tot], "own best", round(tot_own / 1e3, 3))
