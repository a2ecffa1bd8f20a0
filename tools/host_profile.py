import os, sys, cProfile, pstats, io, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from partdistillation_amd import lib; lib.load()
from partdistillation_amd.config import setup_cfg
from partdistillation_amd.engine.synthetic import make_batch
from partdistillation_amd.engine.trainer import TrainStep
S = int(os.environ.get("SIZE", "512"))
cfg = setup_cfg(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "partdistillation_amd/configs/proposal_learning/r50_mask2former.yaml"), ["INPUT.IMAGE_SIZE", str(S)])
torch.manual_seed(0)
step = TrainStep(cfg)
batches = [make_batch(2, S, seed=1234 + 1000 * i, device="cuda") for i in range(4)]
for i in range(8): step(batches[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10): step(batches[i % 4])
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("issue ms/step", (t1 - t0) / 10 * 1e3, "total", (t2 - t0) / 10 * 1e3)
pr = cProfile.Profile(); pr.enable()
for i in range(10): step(batches[i % 4])
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("tottime"); ps.print_stats(45); print(s.getvalue()[:9000])
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("cumulative"); ps.print_stats(60); print(s.getvalue()[:12000])
