"""fp32 weight gradients dW = dY^T X over M = 43 008 tokens (config 2's encoder): exact-fp32 MFMA kernel vs the 3 x bf16 split
kernels (scalar transposed staging / ds_read_b64_tr_b16 transpose reads); back-to-back device time and error against fp64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
sys.path.insert(0, ROOT)
from partdistillation_amd import lib as L
from partdistillation_amd.functions import gemm as G


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


M = int(os.environ.get("M", "43008"))
for N, K in [(1024, 256), (256, 1024), (256, 256), (512, 256), (192, 256), (256, 2304)]:
    torch.manual_seed(0)
    dy = torch.randn(M, N, device="cuda"); x = torch.randn(M, K, device="cuda")
    ref = (dy.double().t() @ x.double())
    refb = dy.double().sum(0)
    row = []
    for name, x3, abl in (("fp32 mfma", False, 0), ("x3 tr-read+ws", True, 0)):
        L.load().pd_debug_set(b"x3_ablate", abl)
        dw = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
        G.gemm_wgrad_acc(dy, x, dw, db, x3=x3)
        err = float((dw.double() - ref).abs().max() / ref.abs().max())
        errb = float((db.double() - refb).abs().max() / refb.abs().max())
        us = timeit(lambda: G.gemm_wgrad_acc(dy, x, dw, db, x3=x3))
        row.append(f"{name}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF  err {err:.1e} bias {errb:.1e}")
    L.load().pd_debug_set(b"x3_ablate", 0)
    ya, xa = G.row_amax(dy), G.row_amax(x)
    dw = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
    G.gemm_wgrad_acc(dy, x, dw, db, h2=True, y_amax=ya, x_amax=xa)
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    us = timeit(lambda: G.gemm_wgrad_acc(dy, x, dw, db, h2=True, y_amax=ya, x_amax=xa))
    row.append(f"f16x2 tr-read+ws: {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF  err {err:.1e}")
    L.load().pd_debug_set(b"wgrad_wide", 0)
    us0 = timeit(lambda: G.gemm_wgrad_acc(dy, x, dw, db, h2=True, y_amax=ya, x_amax=xa))
    L.load().pd_debug_set(b"wgrad_wide", 1)
    row.append(f"f16x2 128x128 tiles: {us0:7.1f} us")
    print(f"N={N:5d} K={K:5d} | " + " | ".join(row), flush=True)
