"""Micro-benchmark of the fp32 MFMA GEMMs vs torch (rocBLAS/hipBLASLt heuristic) at the encoder's shapes."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from partdistillation_amd.functions import gemm


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return ts[len(ts) // 2]


if __name__ == "__main__":
    M = 43008
    for N, K in [(1024, 256), (256, 1024), (256, 256), (288, 256)]:
        a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
        dy = torch.randn(M, N, device="cuda"); wt = w.t().contiguous()
        fl = 2.0 * M * N * K
        r = {"shape": [M, N, K]}
        for name, ours, ref in (("fwd", lambda: gemm.gemm_tn(a, w, b), lambda: torch.nn.functional.linear(a, w, b)),
                                ("dgrad", lambda: gemm.gemm_tn(dy, wt), lambda: dy @ w),
                                ("wgrad", lambda: gemm.gemm_wgrad(dy, a), lambda: dy.t() @ a)):
            t1, t2 = timeit(ours), timeit(ref)
            r[name] = {"ours_ms": round(t1, 4), "ours_TF": round(fl / t1 / 1e9, 1), "torch_ms": round(t2, 4), "torch_TF": round(fl / t2 / 1e9, 1)}
        print(json.dumps(r))
