"""one shape of pd_wgrad_bf16 under rocprofv3 --pmc (development probe): PD_WG_NST / PD_WG_SPLITS / PD_WG_MODE pick the schedule."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tools.bench_swin_wgrad import own_time
M, K, N = 10368, 512, 2048
x = torch.randn((M, K), device="cuda").to(torch.bfloat16)
dy = torch.randn((M, N), device="cuda").to(torch.bfloat16)
for nst, sp, mode in ((2, 4, 0), (2, 4, 2), (2, 1, 2), (1, 12, 0)):
    t, _ = own_time(dy, x, iters=3, wg_nst=nst, wg_splits=sp, wg_mode=mode)
    print(nst, sp, mode, t)
