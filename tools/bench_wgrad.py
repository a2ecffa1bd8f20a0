"""Sweep of the split-K policy of pd_gemm_wgrad_f32 at the encoder's weight-gradient shapes (development tool)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from partdistillation_amd import lib
from partdistillation_amd.functions import gemm
from bench_gemm import timeit  # noqa

M = 43008
for N, K in [(256, 256), (192, 256), (96, 256), (1024, 256), (256, 1024)]:
    x = torch.randn(M, K, device="cuda"); dy = torch.randn(M, N, device="cuda")
    fl = 2.0 * M * N * K
    row = {"shape": [N, K]}
    for wgs in (128, 256, 512, 1024, 2048):
        lib.load().pd_debug_set(b"wgrad_wgs", wgs)
        for bias in (False, True):
            t = timeit(lambda: gemm.gemm_wgrad(dy, x, with_bias=bias))
            row[f"{wgs}{'b' if bias else ''}"] = f"{t * 1e3:.0f}us/{fl / t / 1e9:.0f}TF"
    t = timeit(lambda: dy.t() @ x)
    row["torch"] = f"{t * 1e3:.0f}us/{fl / t / 1e9:.0f}TF"
    print(json.dumps(row))
