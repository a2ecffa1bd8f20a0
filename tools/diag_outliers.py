"""Which steps of a run are slow, and why: per-step HIP-event times next to garbage-collector activity (gc.callbacks), caching-allocator
growth (num_alloc_retries / reserved bytes) and host wall time per step.  Development tool (GPU box): python tools/diag_outliers.py [steps]"""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from partdistillation_amd import lib; lib.load()
from partdistillation_amd.config import setup_cfg
from partdistillation_amd.engine.synthetic import make_batch
from partdistillation_amd.engine.trainer import TrainStep
torch.backends.cudnn.benchmark = True
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd/configs/proposal_learning/r50_mask2former.yaml"), ["INPUT.IMAGE_SIZE", "1024"])
torch.manual_seed(0)
step = TrainStep(cfg)
batches = [make_batch(2, 1024, seed=1234 + 1000 * i, device="cuda") for i in range(4)]
for i in range(8):
    step(batches[i % 4])
torch.cuda.synchronize()
if os.environ.get("GC_FREEZE"):
    gc.collect(); gc.freeze()
log = []
t_gc = {}
def cb(phase, info):
    if phase == "start":
        t_gc["t"] = time.perf_counter()
    else:
        log.append(("gc", cur[0], info["generation"], (time.perf_counter() - t_gc["t"]) * 1e3, info.get("collected")))
gc.callbacks.append(cb)
cur = [0]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
host = []
ev[0].record()
for i in range(steps):
    cur[0] = i
    t0 = time.perf_counter()
    step(batches[i % 4])
    host.append((time.perf_counter() - t0) * 1e3)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
med = sorted(ms)[steps // 2]
print("median %.2f ms, mean %.2f ms; reserved %.2f GiB, alloc retries %d" % (med, sum(ms) / steps, torch.cuda.memory_reserved() / 2**30, torch.cuda.memory_stats().get("num_alloc_retries", 0)))
for i, m in enumerate(ms):
    if m > 1.15 * med or host[i] > 1.3 * sorted(host)[steps // 2]:
        print("step %3d: gpu %.1f ms, host issue %.1f ms" % (i, m, host[i]), [e for e in log if e[1] == i])
print("gen-2 collections:", [(e[1], round(e[3], 1)) for e in log if e[2] == 2])
