"""Where the ATen glue of the training step comes from: torch.profiler over two steps, device time of every aten operator that
launches a copy / cast / fill / add / cat / reduce kernel, grouped by (operator, input shapes, first frame of this package on the
Python stack).  Development tool (GPU box): python tools/glue_ops.py"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
from partdistillation_amd import lib; lib.load()
from partdistillation_amd.config import setup_cfg
from partdistillation_amd.engine.synthetic import make_batch
from partdistillation_amd.engine.trainer import TrainStep
torch.backends.cudnn.benchmark = True
name = os.environ.get("PD_CONFIG", "")                # "" = config 2 (R50 proposal learning); swinb = config 3, swinl = config 5 (SIZE=1280)
size = int(os.environ.get("SIZE", "1024"))
if name:
    cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd/configs/part_distillation", name + "_mask2former.yaml"), ["INPUT.IMAGE_SIZE", str(size)])
else:
    cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd/configs/proposal_learning/r50_mask2former.yaml"), ["INPUT.IMAGE_SIZE", str(size)])
torch.manual_seed(0)
step = TrainStep(cfg)
batches = [make_batch(2, size, seed=1234 + 1000 * i, device="cuda", part_distillation=bool(name)) for i in range(2)]
for i in range(6):
    step(batches[i % 2])
torch.cuda.synchronize()
N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for i in range(N):
        step(batches[i % 2])
    torch.cuda.synchronize()
WANT = ("aten::copy_", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::cat", "aten::sum", "aten::mul", "aten::div", "aten::clone",
        "aten::contiguous", "aten::_to_copy", "aten::stack", "aten::softplus", "aten::sigmoid", "aten::index", "aten::gather", "aten::neg", "aten::sub")
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    ALL = os.environ.get("ALL_ATEN", "0") == "1"               # ALL_ATEN=1: every aten operator with device time (library GEMMs, convolutions, sampling ...)
    want = (lambda n: n.startswith("aten::")) if ALL else (lambda n: n in WANT)
    if not want(e.name) or e.device_time_total <= 0 or e.cpu_children and any(want(c.name) and c.device_time_total > 0 for c in e.cpu_children):
        continue
    frame = next((f for f in (e.stack or []) if "partdistillation_amd" in f and "torch/" not in f), None)
    if frame is None:                                   # no Python stack recorded: the chain of enclosing operators / autograd nodes instead
        chain, q = [], e.cpu_parent
        while q is not None and len(chain) < 4:
            chain.append(q.name)
            q = q.cpu_parent
        frame = " < ".join(chain) or "?"
    frame = frame.replace(ROOT + "/", "")
    key = (e.name, str(e.input_shapes)[:90], frame[:110])
    agg[key][0] += 1
    agg[key][1] += e.device_time_total
tot = sum(v[1] for v in agg.values())
print(f"device time of the listed aten operators: {tot / N / 1e3:.2f} ms / step")
for (name, shapes, frame), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('TOP', '60'))]:
    print(f"{t / N:8.1f} us  x{c / N:5.1f}  {name:12s} {shapes:60.60s} {frame}")
# memsets / device copies (runtime calls, not aten operators with device time of their own): which operator or autograd node issues them
rt = collections.defaultdict(int)
for e in prof.events():
    if not any(s in e.name for s in ("hipMemset", "hipMemcpy", "Memset", "Memcpy")) or e.name.startswith("aten::"):
        continue
    chain, q = [], e.cpu_parent
    while q is not None and len(chain) < 5:
        chain.append(q.name + (" " + str(q.input_shapes)[:50] if q.input_shapes else ""))
        q = q.cpu_parent
    rt[(e.name, " < ".join(chain) or "(no operator: libpd_hip.so / recorded region)")] += 1
print("memsets and copies per step, by issuing operator:")
for (name, chain), c in sorted(rt.items(), key=lambda kv: -kv[1]):
    print(f"  x{c / N:5.1f}  {name:22s} {chain[:200]}")
