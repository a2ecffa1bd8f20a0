"""pd_mx8_gemm against pd_igemm_bf16 at the Swin Linear shapes of BASELINE configs 3 / 5 (one GPU, bs 2): microseconds per launch and
TFLOP/s on 2 M N K, plus the standalone quantisation pass of the activation.  python tools/bench_mx8.py [swinl|swinb]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from partdistillation_amd import lib
from partdistillation_amd.functions import igemm, mx8


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
    L = lib.load()
    if which == "swinl":      # 1280^2, window-padded token counts per stage, C = 192 .. 1536
        stages = [(2 * 324 * 324, 192), (2 * 168 * 168, 384), (2 * 84 * 84, 768), (2 * 48 * 48, 1536)]
    else:                     # Swin-B 1024^2
        stages = [(2 * 264 * 264, 128), (2 * 132 * 132, 256), (2 * 72 * 72, 512), (2 * 36 * 36, 1024)]
    print(f"{'M':>7} {'N':>5} {'K':>5} | bf16 us (TF/s) | mx8 us (TF/s) | quant(x) us | schedules bn/nst: us")
    for M, C in stages:
        for (N, K, tag) in ((3 * C, C, "qkv"), (C, C, "proj"), (4 * C, C, "fc1"), (C, 4 * C, "fc2"), (C, 3 * C, "dqkv")):
            if not mx8.supported(M, N, K):
                continue
            x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
            w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
            b = torch.randn(N, device="cuda").to(torch.bfloat16)
            tb = timeit(lambda: igemm.linear(x, w, b))
            a, wq = mx8.quantize(x), mx8.quantize(w)
            tm = timeit(lambda: mx8.linear(a, wq, b))
            tq = timeit(lambda: mx8.quantize(x))
            fl = 2.0 * M * N * K
            sched = []
            for bn, nst in ((64, 1), (64, 2), (128, 1), (128, 2)):
                if bn == 128 and N % 128:
                    continue
                L.pd_debug_set(b"mx_bn", bn); L.pd_debug_set(b"mx_nst", nst)
                sched.append(f"{bn}/{nst}: {timeit(lambda: mx8.linear(a, wq, b), 15):.1f}")
            L.pd_debug_set(b"mx_bn", 0); L.pd_debug_set(b"mx_nst", 0)
            print(f"{M:7d} {N:5d} {K:5d} | {tb:7.1f} ({fl / tb / 1e6:5.0f}) | {tm:7.1f} ({fl / tm / 1e6:5.0f}) | {tq:6.1f} | {tag:5s} " + "  ".join(sched))


if __name__ == "__main__":
    main()
