import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
import test_r50_fused_gpu as T
from partdistillation_amd.modeling.backbone import resnet_core as rc
net = T._backbone()
x = torch.randn((2, 3, 128, 160), device="cuda")
xx = x.clone().requires_grad_()
with torch.autocast("cuda", dtype=torch.bfloat16):
    outs = net(xx)
loss = sum((v.float() * torch.randn(v.shape, device="cuda")).sum() for v in outs.values())
loss.backward()
plan = list(rc._PLANS.values())[0]
B = plan.shape[0]
names = ["conv1", "conv2", "conv3", "shortcut"]
for bi in (0, 1, 3, 15):
    d = plan.info[bi]
    prev = plan.info[bi - 1] if bi else None
    for ci_, nm in enumerate(names):
        c = plan.convs[4 * bi + ci_]
        if c is None:
            continue
        if nm == "conv1":
            dz, xo, hi, wi, ho, wo = d["g_a"], (prev["out"] if prev else None), d["h"], d["w"], d["h"], d["w"]
        elif nm == "conv2":
            dz, xo, hi, wi, ho, wo = d["g_b"], d["a"], d["h"], d["w"], d["ho"], d["wo"]
        elif nm == "conv3":
            dz, xo, hi, wi, ho, wo = d["g_out"], d["b"], d["ho"], d["wo"], d["ho"], d["wo"]
        else:
            dz, xo, hi, wi, ho, wo = d["g_out"], (prev["out"] if prev else None), d["h"], d["w"], d["ho"], d["wo"]
        if xo is None:
            continue
        g = plan.view(dz, B, ho, wo, c.co).float()
        xin = plan.view(xo, B, hi, wi, c.ci).float()
        w = c.mod.weight.float().detach().requires_grad_()
        z = F.conv2d(xin, w, None, c.stride, c.pad)
        (dw,) = torch.autograd.grad(z, w, g * c.scale.view(1, -1, 1, 1))
        got = plan.arena.as_strided(c.mod.weight.shape, c.mod.weight.stride(), c.dw_off).float()
        e = ((got - dw).abs().max() / dw.abs().max()).item()
        r = (got.norm() / dw.norm()).item()
        print(f"block {bi} {nm}: rel err {e:.3e} norm ratio {r:.3f}  max ref {dw.abs().max().item():.3e} got {got.abs().max().item():.3e}")
