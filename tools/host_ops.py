"""Host (CPU) time of the step by operator: torch.profiler CPU activity over a few steps at a size where the host is the limiter.
Development tool (GPU box): SIZE=512 python tools/host_ops.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
from partdistillation_amd import lib; lib.load()
from partdistillation_amd.config import setup_cfg
from partdistillation_amd.engine.synthetic import make_batch
from partdistillation_amd.engine.trainer import TrainStep
S = int(os.environ.get("SIZE", "512"))
torch.backends.cudnn.benchmark = True
cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd/configs/proposal_learning/r50_mask2former.yaml"), ["INPUT.IMAGE_SIZE", str(S)])
torch.manual_seed(0)
step = TrainStep(cfg)
batches = [make_batch(2, S, seed=1234 + 1000 * i, device="cuda") for i in range(4)]
for i in range(8):
    step(batches[i % 4])
torch.cuda.synchronize()
N = 5
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for i in range(N):
        step(batches[i % 4])
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.self_cpu_time_total)
tot = sum(e.self_cpu_time_total for e in ka)
print("total self CPU %.1f ms / step over %d steps" % (tot / N / 1e3, N))
for e in rows[:45]:
    print("%8.2f ms/step  x %6.1f  %s" % (e.self_cpu_time_total / N / 1e3, e.count / N, e.key[:90]))
