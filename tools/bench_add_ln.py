"""pd_add_layernorm_bwd at the encoder's shape ([43 008, 256] fp32): with its three column-sum outputs (dgamma, dbeta, dbias: every
workgroup ends with 3 C atomic adds into the same addresses) and without them.  python tools/bench_add_ln.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from partdistillation_amd import lib
from partdistillation_amd.functions import rowwise as rw


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


lib.load()
rows, C = int(sys.argv[1]) if len(sys.argv) > 1 else 43008, 256
x, res = torch.randn(rows, C, device="cuda"), torch.randn(rows, C, device="cuda")
g, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
z, y, _, _, mean, rstd = rw.add_ln_fwd(x, res, g, b, 1e-5)
dy = torch.randn(rows, C, device="cuda")
acc = [torch.zeros(C, device="cuda") for _ in range(3)]
t_f = timeit(lambda: rw.add_ln_fwd(x, res, g, b, 1e-5))
t_all = timeit(lambda: rw.add_ln_bwd(z, mean, rstd, g, dy=dy, dgamma=acc[0], dbeta=acc[1], dbias=acc[2]))
t_two = timeit(lambda: rw.add_ln_bwd(z, mean, rstd, g, dy=dy, dgamma=acc[0], dbeta=acc[1]))
t_none = timeit(lambda: rw.add_ln_bwd(z, mean, rstd, g, dy=dy))
caps = {}
for cap in (128, 256, 512, 1024, 2048):
    lib.load().pd_debug_set(b"ln_bwd_cap", cap)
    caps[cap] = timeit(lambda: rw.add_ln_bwd(z, mean, rstd, g, dy=dy, dgamma=acc[0], dbeta=acc[1], dbias=acc[2]))
lib.load().pd_debug_set(b"ln_bwd_cap", 0)
print(f"rows {rows}: fwd {t_f:.1f} us | bwd with 3 column sums {t_all:.1f} us, 2: {t_two:.1f}, none: {t_none:.1f} | by workgroup cap: " + "  ".join(f"{c}: {t:.1f}" for c, t in caps.items()))
