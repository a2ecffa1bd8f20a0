"""a chain of one-thread spin kernels (torch.cuda._sleep) on a side stream next to full-chip kernels on the main stream, host ahead of the GPU"""
import torch
dev = "cuda"
big = torch.randn(64 << 20, device=dev)
A = torch.randn(8192, 4096, device=dev, dtype=torch.bfloat16); B = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
small = lambda n=300: [torch.cuda._sleep(20000) for _ in range(n)]
large_mem = lambda n=30: [big.mul_(1.0001) for _ in range(n)]
large_mm = lambda n=15: [A @ B for _ in range(n)]


def timed(fn):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    large_mm(180)
    s.record(); fn(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)


for name, large in (("HBM-bound", large_mem), ("GEMM", large_mm)):
    for prio in (0, -1):
        side = torch.cuda.Stream(priority=prio)

        def two():
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                small()
            large()
            main.wait_stream(side)

        for _ in range(3):
            small(); large(); two()
        ts, tl = min(timed(small) for _ in range(3)), min(timed(large) for _ in range(3))
        tser = min(timed(lambda: (small(), large())) for _ in range(3))
        ttwo = min(timed(two) for _ in range(3))
        print(f"{name:10s} side priority {prio:2d}: spin chain {ts:.3f} ms, large {tl:.3f} ms, one stream {tser:.3f} ms, two streams {ttwo:.3f} ms")
