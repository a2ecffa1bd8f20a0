import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, "partdistillation_amd", "miopen_db"))
from partdistillation_amd import lib; lib.load()
from partdistillation_amd.config import setup_cfg
from partdistillation_amd.engine.synthetic import make_batch
from partdistillation_amd.engine.trainer import TrainStep
torch.backends.cudnn.benchmark = True
cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd", "configs", "part_distillation", "swinb_mask2former.yaml"), ["INPUT.IMAGE_SIZE", "1024"])
step = TrainStep(cfg)
m = step.model
blk = m.backbone.layers[0].blocks[0]
print("qkv.weight", blk.attn.qkv.weight.dtype, "bias", blk.attn.qkv.bias.dtype, "table", blk.attn.relative_position_bias_table.dtype, "norm", blk.norm1.weight.dtype)
batches = [make_batch(2, 1024, seed=1234 + 1000 * i, device="cuda", part_distillation=True) for i in range(2)]
for i in range(4): step(batches[i % 2])
torch.cuda.synchronize()
# host time of the backbone alone, forward + backward
from partdistillation_amd.modeling.backbone import swin, swin_core
_orig = swin_core.supported
def _sup(layer, x):
    r = _orig(layer, x)
    print("   stage dim", x.shape[-1], "depth", len(layer.blocks), "x", x.dtype, "supported", r)
    return r
swin_core.supported = _sup
x = torch.randn(2, 3, 1024, 1024, device="cuda")
for fused in (True, False):
    swin.FUSED_STAGE = fused
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = m.backbone(x)
        loss = sum(v.float().mean() for v in out.values())
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print("fused", fused, "host fwd %.1f ms bwd %.1f ms, total incl. GPU %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t0) * 1e3))
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = m.backbone(x)
        sum(v.float().mean() for v in out.values()).backward()
        torch.cuda.synchronize()
    ev = prof.key_averages()
    print("   launches", sum(e.count for e in ev), "gpu ms %.1f" % (sum(e.self_device_time_total for e in ev) / 1e3))
    print(ev.table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))
