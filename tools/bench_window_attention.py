"""Per-stage timing of the fused Swin window attention (pd_window_attn_*_w12) at the shapes of BASELINE configs 3 / 5."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from partdistillation_amd import lib
from partdistillation_amd.functions import window_attention as wa
lib.load()

def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

for name, stages in (("config 3 Swin-B 1024^2 bs2", [(264, 4), (132, 8), (72, 16), (36, 32)]),
                     ("config 5 Swin-L 1280^2 bs2", [(324, 6), (168, 12), (84, 24), (48, 48)])):
    print(name)
    for side, heads in stages:
        nW = (side // 12) ** 2
        B_, C = 2 * nW, heads * 32
        qkv = torch.randn(B_, 144, 3 * C, device="cuda").bfloat16()
        table = torch.randn(529, heads, device="cuda") * 0.1
        go = torch.randn(B_, 144, C, device="cuda").bfloat16()
        for shift in (0, 6):
            reg = wa.shifted_window_regions(side, side, shift, "cuda") if shift else None
            out, lse = wa.fwd_raw(qkv, table, reg, 32 ** -0.5, nW)
            tf = t(lambda: wa.fwd_raw(qkv, table, reg, 32 ** -0.5, nW))
            tb = t(lambda: wa.bwd_raw(qkv, table, reg, out, go, lse, 32 ** -0.5, nW))
            abl = []
            for a in (1, 2, 16, 4, 8, 12, 14, 28, 13):
                lib.load().pd_debug_set(b"wattn_ablate", a)
                abl.append("%d:%.0f" % (a, t(lambda: wa.bwd_raw(qkv, table, reg, out, go, lse, 32 ** -0.5, nW))))
            lib.load().pd_debug_set(b"wattn_ablate", 0)
            print("      ablations", " ".join(abl))
            units = B_ * heads
            gf = units * 4 * 144 * 144 * 32 / 1e9          # QK^T + PV forward flops
            print(f"  grid {side}^2 heads {heads:2d} shift {shift}: units {units:6d}  fwd {tf:7.1f} us ({gf / tf * 1e-3 * 1e3:6.1f} TF)"
                  f"  bwd {tb:7.1f} us ({2.5 * gf / tb * 1e-3 * 1e3:6.1f} TF)  qkv {qkv.numel() * 2 / 1e6:.0f} MB")
