"""gemm_kres_f16x2 with phases of its loop removed (pd_debug_set("f16x2_tile", 200 + bits): 1 no split / LDS stores, 2 no products, 4 no fragment
reads, 8 no global loads in the loop; timing only): where the time of a launch goes.  GPU box: python tools/debug/kres_ablate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from partdistillation_amd import lib; L = lib.load()
from partdistillation_amd.functions import gemm
M, N, K = 43008, 256, 1024
a, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * K ** -0.5
aa, wa = gemm.row_amax(a), gemm.row_amax(w)
def t(n=30):
    for _ in range(3): gemm.gemm_tn_h2(a, w, None, a_amax=aa, b_amax=wa)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): gemm.gemm_tn_h2(a, w, None, a_amax=aa, b_amax=wa)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, v in [("tiled 128 x 128", 80), ("resident accumulators", 91), ("  no split / LDS stores", 201), ("  no products", 202), ("  no fragment reads", 204),
                ("  no fragment reads, no products", 206), ("  no global loads", 208), ("  no loads, no stores to LDS", 209), ("  nothing in the loop", 215), ("producer / consumer", 92), ("producer / consumer, weight planes", 93), ("  no split / LDS stores", 221),
                ("  no products", 222), ("  no fragment reads", 224), ("  no fragment reads, no products", 226), ("  no global loads", 228), ("  no loads, no stores to LDS", 229),
                ("  nothing in the loop", 235)]:
    L.pd_debug_set(b"f16x2_tile", v)
    print(f"{name:36s} {t():7.1f} us")
L.pd_debug_set(b"f16x2_tile", 0)
