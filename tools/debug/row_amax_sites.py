"""Which operands still get their row maxima from a pass of their own (pd_row_amax_f32)?  One training step with gemm.row_amax wrapped:
shape, count and the calling site.  python tools/debug/row_amax_sites.py"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from partdistillation_amd.config import setup_cfg
from partdistillation_amd.engine.synthetic import make_batch
from partdistillation_amd.engine.trainer import TrainStep
from partdistillation_amd.functions import gemm
cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd/configs/proposal_learning/r50_mask2former.yaml"), ["INPUT.IMAGE_SIZE", "1024"])
torch.manual_seed(0)
step = TrainStep(cfg)
batch = make_batch(2, 1024, seed=1234, device="cuda")
for _ in range(3):
    step(batch)
seen = collections.Counter()
real = gemm.row_amax
def wrapped(x):
    fr = [f for f in traceback.extract_stack()[:-1] if "partdistillation_amd" in f.filename][-2:]
    seen[(tuple(x.shape), " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr)))] += 1
    return real(x)
gemm.row_amax = wrapped
import partdistillation_amd.functions.conv_x3 as cx, partdistillation_amd.functions.encoder_core as ec
for m in (cx, ec):
    if hasattr(m, "row_amax"):
        m.row_amax = wrapped
os.environ["PD_CMDBUF"] = "0"
step(batch)
for (shape, site), n in sorted(seen.items(), key=lambda kv: -kv[0][0][0] * kv[0][0][1] * kv[1]):
    print(f"{n:3d} x {str(shape):18s} {site}")
