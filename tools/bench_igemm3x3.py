"""The stride-1 3 x 3 convolutions of R50 at 2 x 1024^2 (forward with frozen-BN + ReLU, input gradient with the ReLU gate): the gathered
kernel (pd_debug_set("ig_patch", 0)) against the patch form (default), GPU time per launch through pd_igemm_bf16_time."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from partdistillation_amd import lib
L = lib.load()
from partdistillation_amd.functions import igemm as ig
dev = "cuda"


def t_us(desc, patch, iters=50):
    lib.check(L.pd_debug_set(b"ig_patch", patch))
    need = int(L.pd_igemm_bf16_workspace_bytes(ctypes.byref(desc)))
    ws = ig.workspace(torch.device(dev), need) if need > 0 else None
    us = ctypes.c_float(0)
    lib.check(L.pd_igemm_bf16_time(ctypes.byref(desc), ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, iters, ctypes.byref(us),
                                   lib.current_stream()))
    L.pd_debug_set(b"ig_patch", 1)
    return us.value


B = 2
tot = [0.0, 0.0]
for c, h, n in ((64, 256, 3), (128, 128, 3), (256, 64, 5), (512, 32, 2)):
    x = torch.randn(B, h, h, c, device=dev).to(torch.bfloat16)
    w = (torch.randn(c, 3, 3, c, device=dev) * (9 * c) ** -0.5).to(torch.bfloat16)
    y = torch.empty(B, h, h, c, device=dev, dtype=torch.bfloat16)
    scale, bias = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    p = lambda t: t.data_ptr() if t is not None else None
    for dgrad in (0, 1):
        d = ig.PdIgemm(p(x), p(w), p(scale) if not dgrad else None, p(bias) if not dgrad else None, None, None, p(x) if dgrad else None, p(y), None, B, h, h, c, h, h, c, 3, 1, 1,
                       dgrad, 0 if dgrad else 1, 1 if dgrad else 0, 0)
        a, b_ = t_us(d, 0), t_us(d, 1)
        tot[0] += a * n; tot[1] += b_ * n
        print(f"C = {c:3d}, {h:3d}^2, {'input gradient' if dgrad else 'forward       '}: gathered {a:6.1f} us, patch form {b_:6.1f} us   (x {n} layers)")
print(f"all 26 launches of a step: gathered {tot[0]:.0f} us, patch form {tot[1]:.0f} us")

# the stride-2 3 x 3 input gradients (first block of res3 / res4 / res5): nine-tap walk (pd_debug_set("ig_pcls", 0)) against parity classes
print()
tot = [0.0, 0.0]
for c, ho in ((128, 128), (256, 64), (512, 32)):
    hi = 2 * ho
    dz = torch.randn(B, ho, ho, c, device=dev).to(torch.bfloat16)
    wt = (torch.randn(c, 3, 3, c, device=dev) * (9 * c) ** -0.5).to(torch.bfloat16)
    dx = torch.empty(B, hi, hi, c, device=dev, dtype=torch.bfloat16)
    gate = torch.randn(B, hi, hi, c, device=dev).to(torch.bfloat16)
    p = lambda t: t.data_ptr() if t is not None else None
    d = ig.PdIgemm(p(dz), p(wt), None, None, None, None, p(gate), p(dx), None, B, ho, ho, c, hi, hi, c, 3, 2, 1, 1, 0, 1, 0)
    r = []
    for cls in (0, 1):
        lib.check(L.pd_debug_set(b"ig_pcls", cls))
        need = int(L.pd_igemm_bf16_workspace_bytes(ctypes.byref(d)))
        ws = ig.workspace(torch.device(dev), need) if need > 0 else None
        us = ctypes.c_float(0)
        lib.check(L.pd_igemm_bf16_time(ctypes.byref(d), ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, 50, ctypes.byref(us), lib.current_stream()))
        r.append(us.value)
    L.pd_debug_set(b"ig_pcls", 1)
    tot[0] += r[0]; tot[1] += r[1]
    print(f"C = {c:3d}, {ho:3d}^2 -> {hi:3d}^2 stride-2 input gradient: nine taps {r[0]:6.1f} us, parity classes {r[1]:6.1f} us")
print(f"the three launches of a step: {tot[0]:.0f} -> {tot[1]:.0f} us")
