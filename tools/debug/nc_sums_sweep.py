"""nc_sums (GroupNorm statistics) at the FPN's four map sizes, forward and backward mode; PD_NC_PPB=<pixels per block> overrides the plan"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from partdistillation_amd import lib
L = lib.load()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
st = torch.cuda.current_stream().cuda_stream
for hw in (32, 64, 128, 256):
    N, P, C = 2, hw * hw, 256
    x, dy, y = (torch.randn(N, P, C, device="cuda") for _ in range(3))
    a, b = torch.randn(N, C, device="cuda"), torch.randn(N, C, device="cuda")
    out = torch.zeros(N, C, 2, dtype=torch.float64, device="cuda")
    big = [torch.randn(64 << 20, device="cuda") for _ in range(2)]      # flush between timed calls is skipped: isolated numbers are cache-warm for the small maps
    t0 = timeit(lambda: L.pd_nc_sums_f32(x.data_ptr(), None, None, None, None, out.data_ptr(), N, P, C, 0, 0, st))
    t1 = timeit(lambda: L.pd_nc_sums_f32(x.data_ptr(), dy.data_ptr(), y.data_ptr(), a.data_ptr(), b.data_ptr(), out.data_ptr(), N, P, C, 1, 1, st))
    print(f"{hw:4d}^2  fwd {t0:7.1f} us  bwd {t1:7.1f} us   ({x.numel() * 4 / 1e6:.0f} MB per operand)")
