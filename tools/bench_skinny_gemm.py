"""bf16 forward / input-gradient GEMMs of the Swin stages with small K: own 32 x 128-tile kernels vs the library."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from partdistillation_amd import lib; lib.load()
from partdistillation_amd.functions import smallgemm

def t(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

for M, N, K in [(131072, 384, 128), (131072, 128, 128), (131072, 512, 128), (131072, 128, 512), (32768, 768, 256), (32768, 1024, 256), (32768, 256, 1024),
                (204800, 576, 192), (204800, 192, 192), (204800, 768, 192), (204800, 192, 768), (51200, 1152, 384), (51200, 1536, 384), (51200, 384, 1536)]:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
    dy = torch.randn(M, N, device="cuda").bfloat16()
    tl = t(lambda: torch.addmm(b, x, w.t()))
    ts = t(lambda: smallgemm.linear(x, w, b)) if K % 64 == 0 else float("nan")
    tld = t(lambda: torch.mm(dy, w))
    tsd = t(lambda: smallgemm.dgrad(dy, w)) if N % 64 == 0 else float("nan")
    print(f"M={M:6d} N={N:4d} K={K:4d}: fwd library {tl:6.1f} us own {ts:6.1f} us | dgrad library {tld:6.1f} us own {tsd:6.1f} us | min bytes {(M*(N+K)*2)/1e6:.0f} MB")
