"""fp16 two-plane fp32 GEMM (pd_gemm_tn_f16x2) vs the 3-plane bf16 kernel and the library, config-2 encoder shapes."""
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
for M, N, K in [(43008, 256, 256), (43008, 288, 256), (43500, 256, 256), (131072, 256, 256), (8200, 512, 256), (43520, 1024, 256), (43520, 256, 1024), (43520, 384, 256), (5000, 256, 1000)]:
    for scale_a in (1.0,):
        a = torch.randn(M, K, device="cuda") * scale_a * (1 + 10 * torch.rand(M, 1, device="cuda")); w = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda") * scale_a
        ref = torch.addmm(b.double(), a.double(), w.double().t()); scale = ref.abs().max().item()
        aa, wa = gemm.row_amax(a), gemm.row_amax(w)
        assert torch.equal(aa, a.abs().amax(1))
        err = lambda y: ((y.double() - ref).abs().max().item() / scale)
        y_lib = torch.addmm(b, a, w.t()); y_x3 = gemm.gemm_tn_x3(a, w, b)
        res = []
        for tile in (0, 90, 70, 61, 13, 4):
            if tile in (1, 3, 13, 5, 15, 51, 52) and (N % 256 or M < 1024): continue
            # 0: the product's choice (interleaved interior step); 90: the experimental register-operand kernel where it applies (K = 256); 70: the guarded step by shape; 61: the row stream where it applies; 13 / 4: forced wide / narrow tiles, guarded step
            L.pd_debug_set(b"f16x2_tile", tile)
            y = gemm.gemm_tn_h2(a, w, b, a_amax=aa, b_amax=wa)
            cm = torch.zeros(M, device="cuda"); y2 = gemm.gemm_tn_h2(a, w, b, a_amax=aa, b_amax=wa, c_amax=cm)
            assert torch.equal(y, y2) and torch.equal(cm, y.abs().amax(1)), (tile, (cm - y.abs().amax(1)).abs().max())
            res.append(f"tile{tile} {t(lambda: gemm.gemm_tn_h2(a, w, b, a_amax=aa, b_amax=wa)):6.1f} us (+amax out {t(lambda: gemm.gemm_tn_h2(a, w, b, a_amax=aa, b_amax=wa, c_amax=cm)):6.1f}) err {err(y):.2e}")
        L.pd_debug_set(b"f16x2_tile", 0)
        y_ns = gemm.gemm_tn_h2(a, w, b)
        gf = 2.0 * M * N * K / 1e9
        print(f"M={M} N={N} K={K} |a|~{scale_a:g}: library {t(lambda: torch.addmm(b, a, w.t())):6.1f} us err {err(y_lib):.2e} | x3 {t(lambda: gemm.gemm_tn_x3(a, w, b)):6.1f} us err {err(y_x3):.2e} | "
              + " | ".join(res) + f" | unscaled err {err(y_ns):.2e} | row_amax {t(lambda: gemm.row_amax(a)):5.1f} us")
