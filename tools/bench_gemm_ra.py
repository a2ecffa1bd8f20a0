"""register-operand K = 256 GEMM (gemm_ra_f16x2_k256, pd_debug_set f16x2_tile 0) vs the tiled kernel (80) and the row stream (61)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from partdistillation_amd import lib; L = lib.load()
from partdistillation_amd.functions import gemm
def t(f, n=40):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
torch.manual_seed(0)
for M, N, K in [(43008, 256, 256), (43008, 288, 256), (131072, 256, 256), (43008, 1024, 256), (32768, 256, 256)]:
    a = torch.randn(M, K, device="cuda") * (1 + 10 * torch.rand(M, 1, device="cuda")); w = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
    aa, wa = gemm.row_amax(a), gemm.row_amax(w)
    cm = torch.zeros(M, device="cuda")
    out = []
    for tile in (90, 0, 61):
        L.pd_debug_set(b"f16x2_tile", tile)
        out.append(f"tile{tile} {t(lambda: gemm.gemm_tn_h2(a, w, b, a_amax=aa, b_amax=wa)):6.1f} / {t(lambda: gemm.gemm_tn_h2(a, w, b, a_amax=aa, b_amax=wa, c_amax=cm)):6.1f} us")
    L.pd_debug_set(b"f16x2_tile", 0)
    mb = (M * K + M * N + N * K) * 4 / 1e6
    print(f"M={M} N={N} K={K} ({mb:.0f} MB): " + " | ".join(out))
