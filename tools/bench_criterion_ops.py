"""The criterion kernels (include/pd_criterion.h) in isolation at BASELINE config 2's shapes: python tools/bench_criterion_ops.py
(GPU box; HIP-event time per launch, 50 launches after 5).  `crit_abl` bits ablate phases of pd_uncertain_points (timing only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from partdistillation_amd import lib
from partdistillation_amd.functions import criterion_ops as cops
L = lib.load()
dev = "cuda"
_BLOCK = torch.ones(64 << 20, device=dev)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(40):
        _BLOCK.mul_(1.0001)                     # ~4 ms of queued work: the host issues the timed launches while the GPU is still busy
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


torch.manual_seed(0)
B, H, Q, Pm, nt = 2, 10, 100, 12544, 4
x = (torch.randn(B * H, Q, Pm, device=dev) * 6).to(torch.bfloat16)
t = (torch.rand(B, nt, H * Pm, device=dev) > 0.5).float().view(B, nt, H, Pm)
prob = torch.rand(B * H, Q, 2, device=dev); labels = torch.zeros(B, nt, dtype=torch.long, device=dev)
print(f"matcher_costs [{B * H}, {Q}, {Pm}] bf16, {nt} targets: {timed(lambda: cops.matcher_costs(x, t, prob, labels, H, 5., 2., 5.)):.1f} us")
N, K, k, R = 80, 37632, 9408, 3136
v = torch.randn(N, K, device=dev) * 5; c = torch.rand(N, K, 2, device=dev); r = torch.rand(N, R, 2, device=dev)
for abl, what in ((0, "full"), (1, "plain atomics in the first pass"), (2, "no counting"), (4, "no compaction"), (6, "no counting, no compaction")):
    L.pd_debug_set(b"crit_abl", abl)
    print(f"uncertain_points [{N}, {K}] k={k} ({what}): {timed(lambda: cops.uncertain_points(v, c, k, r)):.1f} us")
L.pd_debug_set(b"crit_abl", 0)
pl = torch.randn(N, 12544, device=dev, requires_grad=True); y = torch.rand(N, 12544, device=dev)
print(f"mask_point_losses fwd: {timed(lambda: cops.mask_point_losses(pl, y)):.1f} us")
masks = torch.rand(2, 4, 1024, 1024, device=dev) > 0.5
mc = torch.rand(2, H * Pm, 2, device=dev)
print(f"point_sample_masks all targets at the matcher's points: {timed(lambda: cops.point_sample_masks(masks, mc, None, 4)):.1f} us")
idx = torch.randint(0, 8, (N,), device=dev); pc = torch.rand(N, 12544, 2, device=dev)
print(f"point_sample_masks one target per pair: {timed(lambda: cops.point_sample_masks(masks, pc, idx)):.1f} us")
