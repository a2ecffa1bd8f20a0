import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from partdistillation_amd.functions.attention import masked_attention_d32
def timeit(fn, iters=20):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev); return ts[len(ts)//2]
for Lk in (16384, 4096, 1024, 100):
    B, H, Lq = 2, 8, 100
    q = torch.randn(Lq, B, 256, device="cuda").bfloat16().requires_grad_(); k = torch.randn(Lk, B, 256, device="cuda").bfloat16().requires_grad_(); v = torch.randn(Lk, B, 256, device="cuda").bfloat16().requires_grad_()
    mask = (torch.rand(B, Lq, Lk, device="cuda") < 0.6) if Lk > 100 else None
    if mask is not None: mask[:, :, 0] = False
    go = torch.randn(Lq, B, 256, device="cuda").bfloat16()
    def ours_f(): return masked_attention_d32(q, k, v, mask, H)
    o = ours_f()
    def ours_b(): torch.autograd.grad(o, (q, k, v), go, retain_graph=True)
    qh = lambda t, L: t.reshape(L, B, H, 32).permute(1, 2, 0, 3)
    fm = None if mask is None else torch.zeros(B, 1, Lq, Lk, device="cuda", dtype=torch.bfloat16).masked_fill_(mask[:, None], float("-inf"))
    def ref_f(): return F.scaled_dot_product_attention(qh(q, Lq), qh(k, Lk), qh(v, Lk), attn_mask=fm)
    ro = ref_f()
    def ref_b(): torch.autograd.grad(ro, (q, k, v), qh(go, Lq), retain_graph=True)
    print(json.dumps({"Lk": Lk, "ours_fwd_us": round(timeit(ours_f)*1e3, 1), "sdpa_fwd_us": round(timeit(ref_f)*1e3, 1), "ours_bwd_us": round(timeit(ours_b)*1e3, 1), "sdpa_bwd_us": round(timeit(ref_b)*1e3, 1)}))
