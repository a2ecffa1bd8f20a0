"""fp32 Linear of the deformable-attention encoder: library fp32 GEMM vs own exact-fp32 MFMA kernel vs the 3-way bf16 split
kernel (pd_gemm_tn_f32x3); time and error against fp64."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from partdistillation_amd import lib; lib.load()
from partdistillation_amd.functions import gemm

def t(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

torch.manual_seed(0)
for M, N, K in [(43008, 256, 256), (43008, 288, 256), (43008, 1024, 256), (43008, 256, 1024), (67200, 1024, 256), (1344, 256, 256)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
    ref = torch.addmm(b.double(), a.double(), w.double().t())
    y_lib, y_own, y_x3 = torch.addmm(b, a, w.t()), gemm.gemm_tn(a, w, b), gemm.gemm_tn_x3(a, w, b)
    scale = ref.abs().max().item()
    e = [((y.double() - ref).abs().max().item() / scale) for y in (y_lib, y_own, y_x3)]
    rms = [((y.double() - ref).pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item()) for y in (y_lib, y_own, y_x3)]
    tl, to, tx = t(lambda: torch.addmm(b, a, w.t())), t(lambda: gemm.gemm_tn(a, w, b)), t(lambda: gemm.gemm_tn_x3(a, w, b))
    abl = []
    for k in (1, 2, 3, 4):
        lib.load().pd_debug_set(b"x3_ablate", k)
        abl.append("%d:%.0f" % (k, t(lambda: gemm.gemm_tn_x3(a, w, b))))
    lib.load().pd_debug_set(b"x3_ablate", 0)
    print("      x3 ablations (1 no MFMA, 2 hi*hi only, 3 no split, 4 no stores):", " ".join(abl))
    gf = 2.0 * M * N * K / 1e9
    print(f"M={M:6d} N={N:4d} K={K:4d}: library {tl:6.1f} us ({gf/tl*1e-3:5.1f} TF) | exact MFMA {to:6.1f} us | x3 {tx:6.1f} us ({gf/tx*1e-3:5.1f} TF)"
          f" | max err/scale lib {e[0]:.2e} own {e[1]:.2e} x3 {e[2]:.2e} | rel rms lib {rms[0]:.2e} x3 {rms[2]:.2e}")

print("weight gradients (accumulating): exact-fp32 MFMA kernel vs 3-way split kernel")
for M, N, K in [(43008, 1024, 256), (43008, 256, 1024), (43008, 256, 256), (43008, 288, 256), (67200, 1024, 256)]:
    dy = torch.randn(M, N, device="cuda"); x = torch.randn(M, K, device="cuda")
    dw = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
    t0 = t(lambda: gemm.gemm_wgrad_acc(dy, x, dw, db, x3=False))
    t1 = t(lambda: gemm.gemm_wgrad_acc(dy, x, dw, db, x3=True))
    gf = 2.0 * M * N * K / 1e9
    print(f"M={M:6d} N={N:4d} K={K:4d}: exact {t0:6.1f} us ({gf/t0*1e-3:5.1f} TF) | x3 {t1:6.1f} us ({gf/t1*1e-3:5.1f} TF)")

