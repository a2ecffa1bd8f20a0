"""3x3 FPN convolution of the pixel decoder (B=2, 256 -> 256 channels at 256^2, fp32): MIOpen vs pd_conv3x3_nhwc_f32x3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from partdistillation_amd import lib; lib.load()
from partdistillation_amd.functions import conv_x3
torch.backends.cudnn.benchmark = True

def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for B, S, C in [(2, 256, 256), (2, 320, 256)]:
    x = torch.randn(B, C, S, S, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_()
    w = (torch.randn(C, C, 3, 3, device="cuda") * 0.02).contiguous(memory_format=torch.channels_last).requires_grad_()
    go = torch.randn(B, C, S, S, device="cuda").contiguous(memory_format=torch.channels_last)
    gf = 2.0 * B * S * S * C * C * 9 / 1e9
    for h2 in (True, False):
        conv_x3.H2 = h2
        yy = conv_x3.conv3x3(x, w)
        print(f"  H2={h2}: forward {t(lambda: conv_x3.conv3x3(x, w)):.3f} ms, backward (dx + dw) {t(lambda: torch.autograd.grad(yy, (x, w), go, retain_graph=True)):.3f} ms, dw alone {t(lambda: torch.autograd.grad(yy, (w,), go, retain_graph=True)):.3f} ms")
    for tile in (70, 21, 0):                                   # the guarded step, the (tap, channel) order, the product (interleaved step)
        lib.load().pd_debug_set(b"f16x2_tile", tile); conv_x3.H2 = True
        print(f"  H2 tile {tile}: forward {t(lambda: conv_x3.conv3x3(x, w)):.3f} ms")
    lib.load().pd_debug_set(b"f16x2_tile", 0)
    conv_x3.H2 = False
    f_lib = t(lambda: F.conv2d(x, w, None, padding=1)); f_x3 = t(lambda: conv_x3.conv3x3(x, w))
    y1 = F.conv2d(x, w, None, padding=1); y2 = conv_x3.conv3x3(x, w)
    b_lib = t(lambda: torch.autograd.grad(y1, (x, w), go, retain_graph=True)); b_x3 = t(lambda: torch.autograd.grad(y2, (x, w), go, retain_graph=True))
    print(f"B={B} {S}^2 C={C}: forward library {f_lib:.3f} ms ({gf/f_lib:.0f} GF/ms) x3 {f_x3:.3f} ms ({gf/f_x3:.0f}) | backward (dx + dw) library {b_lib:.3f} ms x3 {b_x3:.3f} ms")
    wo = t(lambda: torch.autograd.grad(y2, (w,), go, retain_graph=True))
    conv_x3.WGRAD_X3 = False
    wl = t(lambda: torch.autograd.grad(y2, (w,), go, retain_graph=True))
    conv_x3.WGRAD_X3 = True
    print(f"    weight gradient alone: library {wl:.3f} ms, transpose-read split kernel {wo:.3f} ms ({gf/wo:.0f} GF/ms)")
    from partdistillation_amd.functions.gemm import _wgrad_workspace
    L = lib.load()
    dwk = torch.zeros(C, 3, 3, C, device="cuda"); ws = _wgrad_workspace(x.device, int(L.pd_gemm_wgrad_f32x3_ws_floats(C, 9 * C)))
    xd = x.detach()
    def direct():
        lib.check(L.pd_conv3x3_wgrad_nhwc_f32x3(go.data_ptr(), xd.data_ptr(), dwk.data_ptr(), None, ws.data_ptr(), ws.numel(), B, S, S, C, C, lib.current_stream()))
    def direct_atomics():
        lib.check(L.pd_conv3x3_wgrad_nhwc_f32x3(go.data_ptr(), xd.data_ptr(), dwk.data_ptr(), None, None, 0, B, S, S, C, C, lib.current_stream()))
    from partdistillation_amd.functions import gemm as G
    dy2, x2 = go.permute(0, 2, 3, 1).reshape(-1, C), torch.randn(B * S * S, 9 * C, device="cuda")
    dw2 = torch.zeros(C, 9 * C, device="cuda")
    print(f"    direct C call: workspace {t(direct):.3f} ms, atomics {t(direct_atomics):.3f} ms; plain GEMM of the same size (unfolded X) {t(lambda: G.gemm_wgrad_acc(dy2, x2, dw2, None, x3=True)):.3f} ms")
