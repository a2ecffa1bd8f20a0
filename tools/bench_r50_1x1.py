"""1x1 convolutions of R50 at 2 x 1024^2 as plain GEMMs on the NHWC rows: hipBLASLt (torch matmul) and the in-tree
bf16 kernels (sgemm_tn / sgemm_nn / wgrad_split) against MIOpen's conv (tools/bench_r50_convs.py), device time per call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
import torch.nn.functional as F
sys.path.insert(0, ROOT)
from partdistillation_amd.functions import smallgemm as sg


def timeit(fn, n=6):
    from torch.profiler import profile, ProfilerActivity
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return sum(e.device_time_total for e in prof.key_averages()) / n


shapes = [(64, 64, 256, 1), (64, 256, 256, 4), (256, 64, 256, 2), (256, 128, 256, 1), (128, 512, 128, 4), (512, 128, 128, 3),
          (512, 256, 128, 1), (256, 1024, 64, 6), (1024, 256, 64, 5), (1024, 512, 64, 1), (512, 2048, 32, 3), (2048, 512, 32, 2)]
print(f"{'cin->cout @h':20s} cnt |  M      | lt_fwd own_fwd | lt_dgrad own_dgrad | lt_wgrad own_wgrad_split | hbm floor(us @6TB/s)")
tot = [0.0] * 6
for ci, co, h, cnt in shapes:
    M = 2 * h * h
    x = torch.randn(M, ci, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(co, ci, device="cuda", dtype=torch.bfloat16) * 0.05
    dy = torch.randn(M, co, device="cuda", dtype=torch.bfloat16)
    r = [timeit(lambda: F.linear(x, w)), timeit(lambda: sg.linear(x, w)),
         timeit(lambda: dy @ w), timeit(lambda: sg.dgrad(dy, w)),
         timeit(lambda: dy.t() @ x), timeit(lambda: sg.wgrad_split(dy, x, want_bias=False))]
    floor = (M * (ci + co) + ci * co) * 2 / 6.0e6
    print(f"{f'{ci}->{co} @{h}':20s} {cnt:3d} | {M:7d} | " + " ".join(f"{v:8.1f}" for v in r) + f" | {floor:6.1f}")
    for i in range(6):
        tot[i] += cnt * r[i]
print("totals us/step (x count):", [round(v, 1) for v in tot])
