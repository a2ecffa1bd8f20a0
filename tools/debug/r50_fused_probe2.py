import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_r50_fused_gpu as T
from partdistillation_amd.modeling.backbone import resnet_core as rc
net = T._backbone()
net32 = copy.deepcopy(net).float()
x = torch.randn((2, 3, 128, 160), device="cuda")
gs = None
def run(n, fused, amp):
    global gs
    rc.ENABLED = fused
    for p in n.parameters(): p.grad = None
    xx = x.clone().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        outs = n(xx)
    if gs is None:
        gs = {k: torch.randn(v.shape, device="cuda") for k, v in outs.items()}
    sum((v.float() * gs[k]).sum() for k, v in outs.items()).backward()
    return {k: v.detach().float() for k, v in outs.items()}, xx.grad.float(), {k: p.grad.float() for k, p in n.named_parameters() if p.grad is not None}
o_ref, gx_ref, gw_ref = run(net32, False, False)
o_m, gx_m, gw_m = run(net, False, True)
o_f, gx_f, gw_f = run(net, True, True)
err = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
for k in o_ref: print(k, "module", f"{err(o_m[k], o_ref[k]):.3e}", "fused", f"{err(o_f[k], o_ref[k]):.3e}", "max", o_ref[k].abs().max().item())
print("gx module", err(gx_m, gx_ref), "fused", err(gx_f, gx_ref))
for k in list(gw_ref)[:14] + list(gw_ref)[-6:]:
    print(f"{k:30s} module {err(gw_m[k], gw_ref[k]):.3e} fused {err(gw_f[k], gw_ref[k]):.3e}  fused-vs-module {err(gw_f[k], gw_m[k]):.3e}")
