"""the encoder FFN's ReLU-epilogue GEMMs (N = 1024 <- K = 256, modes 1 / 2) on the row stream (default) vs the tiled kernel (f16x2_tile 62)"""
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
M, N, K = 43008, 1024, 256
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
xa, wa = gemm.row_amax(x), gemm.row_amax(w)
cm = torch.zeros(M, device="cuda"); acc = torch.zeros(N, device="cuda")
for tile in (0, 62):
    L.pd_debug_set(b"f16x2_tile", tile)
    h, bits = gemm.gemm_tn_h2(x, w, b, mode=1, want_bits=True, a_amax=xa, b_amax=wa, c_amax=cm)
    f1 = t(lambda: gemm.gemm_tn_h2(x, w, b, mode=1, want_bits=True, a_amax=xa, b_amax=wa, c_amax=cm))
    f2 = t(lambda: gemm.gemm_tn_h2(x, w, None, mode=2, bits=bits, colsum=acc, a_amax=xa, b_amax=wa, c_amax=cm))
    print(f"tile {tile}: mode 1 (ReLU + bits) {f1:6.1f} us, mode 2 (mask + column sums) {f2:6.1f} us")
L.pd_debug_set(b"f16x2_tile", 0)
