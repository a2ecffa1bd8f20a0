"""Step time of BASELINE config 3 on ONE GPU (Swin-B part-distillation training, 1024^2, bs 2, K = 8 x 1000 object
classes): development measurement, not the bench line (config 3 is quoted for 8 GPUs)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from partdistillation_amd import lib
lib.load()
from partdistillation_amd.config import setup_cfg
from partdistillation_amd.engine.synthetic import make_batch
from partdistillation_amd.engine.trainer import TrainStep
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
name = os.environ.get("PD_CONFIG", "swinb")          # swinb = config 3, swinl (+ size 1280) = config 5
extra = os.environ.get("PD_OPTS", "").split()
torch.backends.cudnn.benchmark = True
cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd", "configs", "part_distillation", (name if name.endswith("fp8") else name + "_mask2former") + ".yaml"), ["INPUT.IMAGE_SIZE", str(size)] + extra)
torch.manual_seed(0)
step = TrainStep(cfg)
batches = [make_batch(2, size, seed=1234 + 1000 * i, device="cuda", part_distillation=True) for i in range(2)]
for i in range(4):
    step(batches[i % 2])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    losses = step(batches[i % 2])
issue = time.perf_counter() - t0
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(json.dumps({"workload": "%s part-distillation step, bs 2, %d^2, 1 GPU %s" % (name, size, " ".join(extra)), "ms_per_step": el / steps * 1e3,
                  "host_issue_ms": issue / steps * 1e3, "images_per_s": 2 * steps / el,
                  "loss": float(sum(v.detach() for v in losses.values())), "max_mem_GiB": torch.cuda.max_memory_allocated() / 2**30}))
if len(sys.argv) > 3:
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for i in range(2):
            step(batches[i % 2])
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=70, max_name_column_width=90))
