"""pd_wgrad_bf16: full kernel / loads only / arithmetic only per shape and schedule (development probe)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
sys.argv = sys.argv[:1]
from tools.bench_swin_wgrad import own_time  # noqa
dev = "cuda"
for nm, M, K, N in (("swinB s2 fc1", 10368, 512, 2048), ("swinB s1 fc1", 34848, 256, 1024), ("swinB s3 fc1", 2592, 1024, 4096)):
    x = torch.randn((M, K), device=dev).to(torch.bfloat16)
    dy = torch.randn((M, N), device=dev).to(torch.bfloat16)
    for nst, sp in ((2, 1), (3, 1), (2, 4), (3, 4), (2, 8), (3, 8), (1, 12), (1, 16)):
        row = []
        for mode in (0, 4, 1 | 4, 2 | 4):
            t, _ = own_time(dy, x, wg_nst=nst, wg_splits=sp, wg_mode=mode)
            row.append(t)
        print(f"{nm:14s} nst {nst} splits {sp:3d}: full {row[0]:7.1f}  main loop {row[1]:7.1f}  its loads only {row[2]:7.1f}  its arithmetic only {row[3]:7.1f} us", flush=True)
