"""pd_swin_ln_fwd / pd_swin_ln_bwd at the stage shapes of BASELINE configs 3 / 5 (bs 2): microseconds and achieved TB/s on the
algorithmic bytes (fwd: x + r read, s + y written; bwd: dy + s + dsup read, ds + dr written), with and without the MX copy,
and the backward without its column-sum atomics (pd_debug_set swin_ln_abl 1).  python tools/bench_swin_rows.py [swinl|swinb]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from partdistillation_amd import lib
from partdistillation_amd.functions import swin_rows as rows


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "swinl"
    L_ = lib.load()
    stages = [(324 * 324, 192), (168 * 168, 384), (84 * 84, 768), (48 * 48, 1536)] if which == "swinl" else \
             [(264 * 264, 128), (132 * 132, 256), (72 * 72, 512), (36 * 36, 1024)]
    B = 2
    for L, C in stages:
        x = torch.randn(B * L, C, device="cuda")
        r = torch.randn(B * L, C, device="cuda").to(torch.bfloat16)
        g, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
        dy = torch.randn(B * L, C, device="cuda").to(torch.bfloat16)
        dsup = torch.randn(B * L, C, device="cuda")
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        s, y, st = rows.ln_fwd(x, r, None, L, None, g, b, 1e-5, None, L, None, B, L)
        n = B * L * C
        fb, bb = n * (4 + 2 + 4 + 2), n * (2 + 4 + 4 + 4 + 2)
        tf = timeit(lambda: rows.ln_fwd(x, r, None, L, None, g, b, 1e-5, None, L, None, B, L))
        tfm = timeit(lambda: rows.ln_fwd(x, r, None, L, None, g, b, 1e-5, None, L, None, B, L, mx=0)) if C % 128 == 0 else float("nan")
        tb = timeit(lambda: rows.ln_bwd(dy, None, L, dsup, s, st, g, True, None, L, None, None, dg, db, B, L))
        tbm = timeit(lambda: rows.ln_bwd(dy, None, L, dsup, s, st, g, True, None, L, None, None, dg, db, B, L, mx=1)) if C % 128 == 0 else float("nan")
        dg8 = torch.zeros(8, 2, C, device="cuda")
        tb8 = timeit(lambda: rows.ln_bwd(dy, None, L, dsup, s, st, g, True, None, L, None, None, dg8[0, 0], dg8[0, 1], B, L, n_rep=8, rep_stride=2 * C))
        L_.pd_debug_set(b"swin_ln_abl", 1)
        tba = timeit(lambda: rows.ln_bwd(dy, None, L, dsup, s, st, g, True, None, L, None, None, dg, db, B, L))
        L_.pd_debug_set(b"swin_ln_abl", 0)
        print(f"rows {B * L:7d} C {C:5d} | fwd {tf:6.1f} us {fb / tf / 1e6:5.2f} TB/s (+MX {tfm:6.1f}) | bwd {tb:6.1f} us {bb / tb / 1e6:5.2f} TB/s (+MX {tbm:6.1f}; 8 copies of the sums {tb8:6.1f}; no atomics {tba:6.1f})")


if __name__ == "__main__":
    main()
