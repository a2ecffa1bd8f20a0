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
M, N, K = 32768, 256, 256
a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
aa, wa = gemm.row_amax(a), gemm.row_amax(w)
for abl in (0, 1, 2, 4, 6, 8, 16, 23, 31):
    L.pd_debug_set(b"f16x2_tile", 100 + abl if abl else 90)
    print(f"abl {abl:2d}: {t(lambda: gemm.gemm_tn_h2(a, w, b, a_amax=aa, b_amax=wa)):6.1f} us")
