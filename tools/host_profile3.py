"""Development tool: host-side cProfile of the Swin-B part-distillation step (config 3) at a small image size."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(ROOT, "partdistillation_amd", "miopen_db"))
from partdistillation_amd import lib
lib.load()
from partdistillation_amd.config import setup_cfg
from partdistillation_amd.engine.synthetic import make_batch
from partdistillation_amd.engine.trainer import TrainStep
size = int(sys.argv[1]) if len(sys.argv) > 1 else 384

cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd", "configs", "part_distillation", "swinb_mask2former.yaml"), ["INPUT.IMAGE_SIZE", str(size)])
torch.manual_seed(0)
step = TrainStep(cfg)
batches = [make_batch(2, size, seed=1234 + 1000 * i, device="cuda", part_distillation=True) for i in range(2)]
for i in range(5):
    step(batches[i % 2])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    step(batches[i % 2])
issue = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"size {size}: host issue {issue * 100:.2f} ms/step, wall {(time.perf_counter() - t0) * 100:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    step(batches[i % 2])
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
