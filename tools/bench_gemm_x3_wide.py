import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from partdistillation_amd import lib; L = lib.load()
from partdistillation_amd.functions import gemm
def t(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
torch.manual_seed(0)
for M, N, K in [(43008, 256, 256), (43008, 1024, 256), (43008, 256, 1024), (43008, 512, 256), (67200, 1024, 256), (5000, 256, 1000)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
    ref = torch.addmm(b.double(), a.double(), w.double().t())
    y_lib = torch.addmm(b, a, w.t())
    L.pd_debug_set(b"x3_narrow", 1); y_n = gemm.gemm_tn_x3(a, w, b); tn = t(lambda: gemm.gemm_tn_x3(a, w, b))
    L.pd_debug_set(b"x3_narrow", 2); y_w2 = gemm.gemm_tn_x3(a, w, b); tw2 = t(lambda: gemm.gemm_tn_x3(a, w, b))
    L.pd_debug_set(b"x3_narrow", 3); y_w3 = gemm.gemm_tn_x3(a, w, b); tw3 = t(lambda: gemm.gemm_tn_x3(a, w, b))
    assert (y_w3 - y_w2).abs().max().item() <= 1e-5 * ref.abs().max().item()
    L.pd_debug_set(b"x3_narrow", 0); y_w = gemm.gemm_tn_x3(a, w, b); tw = t(lambda: gemm.gemm_tn_x3(a, w, b))
    yr = gemm.gemm_tn_x3(a, w, b, relu=True)
    tl = t(lambda: torch.addmm(b, a, w.t()))
    scale = ref.abs().max().item()
    e = [((y.double() - ref).abs().max().item() / scale) for y in (y_lib, y_n, y_w)]
    er = ((yr.double() - ref.relu()).abs().max().item() / scale)
    gf = 2.0 * M * N * K / 1e9
    print(f"M={M:6d} N={N:4d} K={K:4d}: library {tl:6.1f} us ({gf/tl*1e-3:5.1f} TF) | x3 128x128 {tn:6.1f} us | x3 256x256 {tw2:6.1f} us | x3 128x256 2/CU {tw3:6.1f} us | x3 default {tw:6.1f} us ({gf/tw*1e-3:5.1f} TF fp32-equiv, {6*gf/tw*1e-3:5.0f} TF bf16)"
          f" | max err/scale lib {e[0]:.2e} narrow {e[1]:.2e} wide {e[2]:.2e} relu {er:.2e}")

print("pre-split weight planes (pd_gemm_tn_f32x3_pre) vs in-kernel split, 256 x 256 kernel, us")
for M, N, K in [(43008, 1024, 256), (43008, 256, 1024)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5; bb = torch.randn(N, device="cuda")
    L.pd_debug_set(b"x3_narrow", 2)
    t0 = t(lambda: gemm.gemm_tn_x3(a, w, bb))
    L.pd_debug_set(b"x3_narrow", 0)
    pl = gemm.split3(w)
    t1 = t(lambda: gemm.gemm_tn_x3_pre(a, pl, bb)); ts = t(lambda: gemm.split3(w))
    print(M, N, K, f"in-kernel {t0:.1f} | pre-split {t1:.1f} (+ split kernel {ts:.1f})")
print("wide-kernel ablations (us): 11 no MFMA, 12 no output stores, 13 no operand split")
for M, N, K in [(43008, 1024, 256), (43008, 256, 1024)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
    r = []
    for k in (0, 11, 12, 13):
        L.pd_debug_set(b"x3_ablate", k)
        r.append("%d: %.0f" % (k, t(lambda: gemm.gemm_tn_x3(a, w, b))))
    L.pd_debug_set(b"x3_ablate", 0)
    print(M, N, K, " | ".join(r))
