import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from partdistillation_amd import lib
from partdistillation_amd.functions import window_attention as wa
lib.load()
def t(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for side, heads in [(264, 4), (72, 16), (36, 32)]:
    nW = (side // 12) ** 2
    B_, C = 2 * nW, heads * 32
    qkv = torch.randn(B_, 144, 3 * C, device="cuda").bfloat16()
    table = torch.randn(529, heads, device="cuda") * 0.1
    res = []
    for a in (0, 32, 64, 96):
        lib.load().pd_debug_set(b"wattn_ablate", a)
        res.append("%d:%.1f" % (a, t(lambda: wa.fwd_raw(qkv, table, None, 32 ** -0.5, nW))))
    lib.load().pd_debug_set(b"wattn_ablate", 0)
    print(side, heads, " ".join(res))
