"""Probe: library bf16 GEMM time of the Swin stage-3 shapes in the layouts a Linear's forward / input gradient can be
posed in (weight as stored [N, K], or a transposed copy), to see whether a per-step weight transpose would pay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

def t(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

for M, N, K in [(14112, 2304, 768), (14112, 768, 768), (12800, 3072, 768), (12800, 768, 3072), (8192, 1536, 512), (8192, 2048, 512), (8192, 512, 2048),
                (56448, 1152, 384), (51200, 1536, 384), (51200, 384, 1536)]:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
    wt = w.t().contiguous()
    dy = torch.randn(M, N, device="cuda").bfloat16()
    f1 = t(lambda: torch.addmm(b, x, w.t()))
    f2 = t(lambda: torch.addmm(b, x, wt))
    d1 = t(lambda: torch.mm(dy, w))
    d2 = t(lambda: torch.mm(dy, wt.t()))
    tr = t(lambda: w.t().contiguous())
    gf = 2.0 * M * N * K / 1e9
    print(f"M={M:6d} N={N:4d} K={K:4d}: fwd w.t() {f1:6.1f} us ({gf / f1 * 1e-3:.0f} TF)  fwd wt {f2:6.1f} | dgrad w {d1:6.1f} us ({gf / d1 * 1e-3:.0f} TF)  dgrad wt.t() {d2:6.1f} | transpose {tr:4.1f} us")
