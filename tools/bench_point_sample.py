import torch, sys
sys.path.insert(0, "/root/repo")
import torch.nn.functional as F
from partdistillation_amd.functions import rowwise as rw
def timeit(fn, n=10):
    from torch.profiler import profile, ProfilerActivity
    for _ in range(3): fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(n): fn()
        torch.cuda.synchronize()
    return {e.key[:40]: round(e.device_time_total / n, 1) for e in prof.key_averages()}
for (N, C, H, W, P) in [(80, 1, 256, 256, 37632), (80, 1, 256, 256, 12544), (1, 4, 1024, 1024, 501760)]:
    x = torch.randn(N, C, H, W, device="cuda", requires_grad=True)
    co = torch.rand(N, P, 2, device="cuda")
    go = torch.randn(N, C, P, device="cuda")
    print((N, C, H, W, P), "own fwd", timeit(lambda: rw.point_sample_planar(x, co)))
    print("   torch fwd", timeit(lambda: F.grid_sample(x, 2 * co.unsqueeze(2) - 1, mode="bilinear", padding_mode="zeros", align_corners=False)))
    y = rw.point_sample_planar(x, co)
    print("   own bwd", timeit(lambda: torch.autograd.grad(y, x, go, retain_graph=True)))
    y2 = F.grid_sample(x, 2 * co.unsqueeze(2) - 1, mode="bilinear", padding_mode="zeros", align_corners=False).squeeze(3)
    print("   torch bwd", timeit(lambda: torch.autograd.grad(y2, x, go, retain_graph=True)))
