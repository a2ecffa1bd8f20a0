"""Development tool: error ladder of the bf16 MFMA attention kernels against an fp64 reference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from partdistillation_amd.functions.attention import masked_attention_d32
from test_attention_gpu import _ref

for (Lq, Lk, B, H, masked) in [(32, 32, 1, 1, False), (32, 64, 1, 1, False), (32, 32, 1, 1, True), (100, 100, 2, 8, False), (100, 256, 1, 2, True),
                               (100, 512, 1, 2, True), (100, 1024, 2, 8, True), (7, 70, 3, 2, True)]:
    g = torch.Generator(device="cuda").manual_seed(Lq + Lk)
    C = H * 32
    q = torch.randn(Lq, B, C, device="cuda", generator=g).bfloat16().requires_grad_()
    k = torch.randn(Lk, B, C, device="cuda", generator=g).bfloat16().requires_grad_()
    v = torch.randn(Lk, B, C, device="cuda", generator=g).bfloat16().requires_grad_()
    mask = None
    if masked:
        mask = torch.rand(B, Lq, Lk, device="cuda", generator=g) < 0.7
        mask[:, :, 0] = False
    go = torch.randn(Lq, B, C, device="cuda", generator=g).bfloat16()
    o = masked_attention_d32(q, k, v, mask, H)
    dq, dk, dv = torch.autograd.grad(o, (q, k, v), go)
    qr, kr, vr = [t.detach().double().requires_grad_() for t in (q, k, v)]
    ro = _ref(qr, kr, vr, mask, H)
    rq, rk, rv = torch.autograd.grad(ro, (qr, kr, vr), go.double())
    err = lambda a, b: f"{(a.double() - b).abs().max().item():.3e}"
    print((Lq, Lk, B, H, masked), "o", err(o, ro), "dq", err(dq, rq), "dk", err(dk, rk), "dv", err(dv, rv), flush=True)
