"""Probe: does the library fp8 GEMM (hipBLASLt through torch._scaled_mm) run on gfx950 for the operand mixes an fp8
training step needs, and how does it compare with the bf16 GEMM at the Swin-L / decoder shapes of config 5."""
import sys, time, torch
dev = "cuda"
E4, E5 = torch.float8_e4m3fn, torch.float8_e5m2
one = torch.ones((), device=dev)

def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

def mm8(a, b):  # a [M,K] row-major fp8, b [N,K] row-major fp8 -> [M,N] bf16
    return torch._scaled_mm(a, b.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16)

for da, db in ((E4, E4), (E5, E4), (E4, E5)):
    try:
        a = torch.randn(256, 128, device=dev).to(da); b = torch.randn(64, 128, device=dev).to(db)
        y = mm8(a, b); ref = a.float() @ b.float().t()
        print(da, db, "ok, max err", float((y.float() - ref).abs().max()), "ref max", float(ref.abs().max()))
    except Exception as e:
        print(da, db, "FAILED", type(e).__name__, str(e)[:200])

shapes = [(204800, 576, 192), (204800, 192, 192), (204800, 768, 192), (204800, 192, 768),
          (51200, 1152, 384), (51200, 1536, 384), (12800, 2304, 768), (12800, 3072, 768), (3200, 6144, 1536),
          (67200, 1024, 256), (200, 2048, 256), (8192, 8192, 8192)]
for M, N, K in shapes:
    xa = torch.randn(M, K, device=dev, dtype=torch.bfloat16); wb = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    x8, w8 = xa.to(E4), wb.to(E4)
    tb = t(lambda: xa @ wb.t())
    try:
        t8 = t(lambda: mm8(x8, w8))
    except Exception as e:
        t8 = float("nan"); print("fp8 failed", M, N, K, str(e)[:120])
    tq = t(lambda: xa.to(E4))
    gf = 2 * M * N * K / 1e9
    print(f"M={M} N={N} K={K}: bf16 {tb:.1f} us ({gf/tb*1e-3:.0f} TF)  fp8 {t8:.1f} us ({gf/t8*1e-3:.0f} TF)  cast {tq:.1f} us")
