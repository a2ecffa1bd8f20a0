"""Host issue time of the training step by section (no device synchronisation inside the step: wall time the calling thread spends
issuing each part), plus cProfile's top Python functions by own time.  Development tool (GPU box): [PD_CONFIG=swinb] python tools/host_sections.py"""
import cProfile, os, pstats, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from partdistillation_amd import lib; lib.load()
from partdistillation_amd.config import setup_cfg
from partdistillation_amd.engine.synthetic import make_batch
from partdistillation_amd.engine.trainer import TrainStep
S = int(os.environ.get("SIZE", "1024"))
torch.backends.cudnn.benchmark = True
NAME = os.environ.get("PD_CONFIG", "")             # "": BASELINE config 2 (R50 proposal learning); swinb / swinl: configs 3 / 5 (part distillation)
YAML = "partdistillation_amd/configs/proposal_learning/r50_mask2former.yaml" if not NAME else \
    "partdistillation_amd/configs/part_distillation/%s.yaml" % (NAME if NAME.endswith("fp8") else NAME + "_mask2former")
cfg = setup_cfg(os.path.join(ROOT, YAML), ["INPUT.IMAGE_SIZE", str(S)])
torch.manual_seed(0)
step = TrainStep(cfg)
batches = [make_batch(2, S, seed=1234 + 1000 * i, device="cuda", part_distillation=bool(NAME)) for i in range(4)]
for i in range(8):
    step(batches[i % 4])
torch.cuda.synchronize()
acc = collections.defaultdict(float)
model = step.model
mods = {"backbone": model.backbone, "pixel_decoder": model.sem_seg_head.pixel_decoder, "decoder": model.sem_seg_head.predictor, "criterion": model.criterion}
t_in = {}
hooks = []
for name, m in mods.items():
    hooks.append(m.register_forward_pre_hook(lambda mod, a, n=name: t_in.__setitem__(n, time.perf_counter())))
    hooks.append(m.register_forward_hook(lambda mod, a, o, n=name: acc.__setitem__(n, acc[n] + time.perf_counter() - t_in[n])))
fb, opt = step._forward_backward, step.optimizer.step
def fb_timed(b):
    t0 = time.perf_counter(); r = fb(b); acc["forward+backward"] += time.perf_counter() - t0; return r
def opt_timed():
    t0 = time.perf_counter(); r = opt(); acc["optimizer.step"] += time.perf_counter() - t0; return r
step._forward_backward, step.optimizer.step = fb_timed, opt_timed
N = 10
t0 = time.perf_counter()
for i in range(N):
    step(batches[i % 4])
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host issue {t_host / N * 1e3:.2f} ms / step, wall {t_all / N * 1e3:.2f} ms / step")
fwd = sum(acc[k] for k in mods)
for k in list(mods) + ["forward+backward", "optimizer.step"]:
    print(f"  {k:18s} {acc[k] / N * 1e3:7.2f} ms")
print(f"  {'backward (+glue)':18s} {(acc['forward+backward'] - fwd) / N * 1e3:7.2f} ms   (forward+backward minus the four forward sections)")
for h in hooks:
    h.remove()
pr = cProfile.Profile()
pr.enable()
for i in range(3):
    step(batches[i % 4])
pr.disable()
torch.cuda.synchronize()
import io
for key, n in (("tottime", 45), ("cumtime", 60)):
    st = pstats.Stats(pr); st.sort_stats(key)
    buf = io.StringIO(); st.stream = buf; st.print_stats(n)
    print("\n".join(l[:170] for l in buf.getvalue().splitlines()[:n + 12]))
