"""pd_kmeans_assign at the shapes of BASELINE config 4 (4 images x 5431 object pixels x C channels, K = 4) with ablations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from partdistillation_amd import lib; L = lib.load()
from partdistillation_amd.functions.fused import upload_small
for C in (1536, 1152):
    B, N, K = 4, 5431, 4
    X = torch.randn(B * N, C, device="cuda")
    centers = torch.randn(B, K, C, device="cuda"); cnorm = (centers * centers).sum(-1).contiguous()
    for slab in (64, 32):
        table = [(b, b * N + s0, min(slab, N - s0)) for b in range(B) for s0 in range(0, N, slab)]
        blocks = upload_small([v for r in table for v in r], torch.int32, "cuda").view(-1, 3)
        labels = torch.zeros(B * N, dtype=torch.int32, device="cuda"); sums = torch.zeros(B, K, C, device="cuda")
        counts = torch.zeros(B, K, device="cuda"); flags = torch.zeros(3, B, dtype=torch.int32, device="cuda")
        st = lib.current_stream()
        def run():
            lib.check(L.pd_kmeans_assign(X.data_ptr(), blocks.data_ptr(), len(table), centers.data_ptr(), cnorm.data_ptr(), flags[1].data_ptr(),
                                         labels.data_ptr(), sums.data_ptr(), counts.data_ptr(), flags[0].data_ptr(), C, K, st))
        out = []
        for abl in (0, 1, 2, 4, 3):
            L.pd_debug_set(b"kmeans_ablate", abl)
            for _ in range(3): run()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): run()
            torch.cuda.synchronize(); out.append("%d:%.0f" % (abl, (time.perf_counter() - t0) / 30 * 1e6))
        L.pd_debug_set(b"kmeans_ablate", 0)
        print(f"C={C} slab={slab} WGs={len(table)}: us per launch by ablation (0 full, 1 no label pass, 2 no sums pass, 4 no atomics, 3 neither pass):", " ".join(out),
              f"| X = {X.numel() * 4 / 1e6:.0f} MB")
