"""fp32 256-wide projections of the encoder (M = 43 008) through hipBLASLt vs rocBLAS (torch.backends.cuda.preferred_blas_library)."""
import torch
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 43008
x = torch.randn(M, 256, device="cuda"); w = torch.randn(256, 256, device="cuda"); b = torch.randn(256, device="cuda")
w2 = torch.randn(288, 256, device="cuda"); b2 = torch.randn(288, device="cuda"); g2 = torch.randn(M, 288, device="cuda")
for lib in ("hipblaslt", "hipblas"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print(lib, "unavailable", e); continue
    print(lib, "addmm 256<-256 %.1f us | mm dgrad 256 %.1f | addmm 288<-256 %.1f | mm dgrad 288 %.1f" % (
        timeit(lambda: torch.addmm(b, x, w.t())), timeit(lambda: torch.mm(x, w)),
        timeit(lambda: torch.addmm(b2, x, w2.t())), timeit(lambda: torch.mm(g2, w2))))
