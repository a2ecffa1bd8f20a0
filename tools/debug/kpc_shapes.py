import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from partdistillation_amd import lib; L = lib.load()
from partdistillation_amd.functions import gemm
def t(f, n=30):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N, K in [(43008, 256, 256), (43008, 1024, 256), (43008, 256, 1024), (43008, 512, 1024), (131072, 256, 512), (131072, 256, 256)]:
    a, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * K ** -0.5, torch.randn(N, device="cuda")
    aa, wa = gemm.row_amax(a), gemm.row_amax(w)
    ref = torch.addmm(b.double(), a.double(), w.double().t())
    out = []
    for name, v in (("default", 0), ("tiled", 80), ("kpc", 92)):
        L.pd_debug_set(b"f16x2_tile", v)
        y = gemm.gemm_tn_h2(a, w, b, a_amax=aa, b_amax=wa)
        err = ((y.double() - ref).abs() / ref.abs().amax(1, keepdim=True)).max().item()
        out.append(f"{name} {t(lambda: gemm.gemm_tn_h2(a, w, b, a_amax=aa, b_amax=wa)):6.1f} us (err {err:.1e})")
    L.pd_debug_set(b"f16x2_tile", 0)
    print(f"M={M} N={N} K={K}: " + " | ".join(out))
