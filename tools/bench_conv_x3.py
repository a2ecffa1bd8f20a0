"""3x3 FPN convolution of the pixel decoder (B=2, 256 -> 256 channels at 256^2, fp32): MIOpen vs pd_conv3x3_nhwc_f32x3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "partdistillation_amd", "miopen_db"))
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
    f_lib = t(lambda: F.conv2d(x, w, None, padding=1)); f_x3 = t(lambda: conv_x3.conv3x3(x, w))
    y1 = F.conv2d(x, w, None, padding=1); y2 = conv_x3.conv3x3(x, w)
    b_lib = t(lambda: torch.autograd.grad(y1, (x, w), go, retain_graph=True)); b_x3 = t(lambda: torch.autograd.grad(y2, (x, w), go, retain_graph=True))
    print(f"B={B} {S}^2 C={C}: forward library {f_lib:.3f} ms ({gf/f_lib:.0f} GF/ms) x3 {f_x3:.3f} ms ({gf/f_x3:.0f}) | backward (dx + dw) library {b_lib:.3f} ms x3 {b_x3:.3f} ms")
