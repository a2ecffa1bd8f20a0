"""the stem convolution: own kernels (pd_stem7x7_*) vs the library path (MIOpen conv + pd_affine_act + autograd backward), 2 x 1024^2"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from partdistillation_amd.modeling.backbone import resnet as R
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
torch.manual_seed(0)
m = R.BasicStem(3, 64, "FrozenBN").cuda()
m.conv1.weight.data = m.conv1.weight.data.bfloat16().contiguous(memory_format=torch.channels_last)
x = torch.randn(2, 3, 1024, 1024, device="cuda").contiguous(memory_format=torch.channels_last)
g = None
def fwd():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return m(x)
def fb():
    global g
    y = fwd()
    if g is None: g = torch.randn_like(y)
    m.conv1.weight.grad = None
    y.backward(g)
for own in (True, False):
    R.OWN_STEM = own
    print(f"own={own}: forward (conv + BN + ReLU + max pool) {t(fwd):7.1f} us, forward + backward {t(fb):7.1f} us")
