"""bf16 weight-gradient GEMM over many rows: pd_sgemm_wgrad_split_bf16 against the library (torch.mm + column sum)."""
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

for M, N, K in [(8192, 1536, 512), (8192, 512, 512), (8192, 2048, 512), (8192, 512, 2048), (32768, 768, 256), (131072, 384, 128),
                (131072, 512, 128), (2048, 3072, 1024), (12800, 2304, 768), (12800, 3072, 768), (51200, 1152, 384), (204800, 576, 192), (204800, 768, 192)]:
    dy = torch.randn(M, N, device="cuda").bfloat16(); x = torch.randn(M, K, device="cuda").bfloat16()
    tl = t(lambda: (torch.mm(dy.t(), x), dy.sum(0, dtype=torch.float32)))
    ts = t(lambda: smallgemm.wgrad_split(dy, x, True))
    gf = 2.0 * M * N * K / 1e9
    print(f"M={M:6d} N={N:4d} K={K:4d}: library mm + sum {tl:7.1f} us, split kernel {ts:7.1f} us ({gf / ts:.0f} TFLOP/s), bytes {(M*(N+K)*2)/1e6:.0f} MB")
