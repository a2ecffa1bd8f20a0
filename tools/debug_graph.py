import os, sys
ROOT0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(ROOT0, 'partdistillation_amd', 'miopen_db'))
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from partdistillation_amd.config import setup_cfg
from partdistillation_amd.engine.synthetic import make_batch
from partdistillation_amd.engine.trainer import TrainStep
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
opts = [] if size >= 512 else ["MODEL.MASK_FORMER.NUM_OBJECT_QUERIES", "20", "MODEL.MASK_FORMER.DEC_LAYERS", "4", "MODEL.SEM_SEG_HEAD.TRANSFORMER_ENC_LAYERS", "2", "MODEL.MASK_FORMER.TRAIN_NUM_POINTS", "256"]
cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd", "configs", "proposal_learning", "r50_mask2former.yaml"), opts)
torch.manual_seed(0)
step = TrainStep(cfg)
batches = [make_batch(2, size, n_parts=3, seed=5 + i, device="cuda") for i in range(2)]
for i in range(3):
    step(batches[i % 2])
torch.cuda.synchronize(); print("eager ok", flush=True)
if mode == "fwd":
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ld = step.model(batches[0])
    torch.cuda.synchronize(); print("captured fwd", flush=True)
    for i in range(3):
        g.replay(); torch.cuda.synchronize(); print("replay fwd ok", i, float(ld.total), flush=True)
elif mode == "fwdbwd":
    if os.environ.get("DBG_CLONE"):
        from partdistillation_amd.compat.structures import BitMasks, Instances
        st = []
        for x in batches[0]:
            inst = Instances(x["instances"].image_size)
            inst.gt_masks = BitMasks(x["instances"].gt_masks.tensor.clone())
            inst.gt_classes = x["instances"].gt_classes.clone()
            st.append({**{k: v for k, v in x.items() if k not in ("image", "instances")}, "image": x["image"].clone(), "instances": inst})
        batches[0] = st
    g = torch.cuda.CUDAGraph()
    step.optimizer.zero_grad()
    with torch.cuda.graph(g):
        ld = step._forward_backward(batches[0])
    torch.cuda.synchronize(); print("captured fwdbwd", flush=True)
    for i in range(3):
        g.replay(); torch.cuda.synchronize(); print("replay fwdbwd ok", i, float(ld.total), flush=True)
else:
    if os.environ.get("DBG_NO_ADAMW"):
        import partdistillation_amd.functions.optim as O
        O.adamw_clipped_ = lambda *a, **k: None
    if os.environ.get("DBG_NO_GATHER"):
        from partdistillation_amd.engine.flat_params import FlatGroup
        FlatGroup.gather = lambda self, *a, **k: None
    if os.environ.get("DBG_NO_LAUNCH"):
        step.optimizer.launch_step = lambda: None
    step.capture(batches[0], warmup=0 if os.environ.get("DBG_NO_WARM") else 3)
    torch.cuda.synchronize(); print("captured full", flush=True)
    if os.environ.get("DBG_NO_PREP"):
        step.optimizer.prepare_step = lambda: None
    if os.environ.get("DBG_STATIC"):
        batches = [step._static, step._static]
    if os.environ.get("DBG_RAW"):
        for i in range(4):
            step._graph.replay(); torch.cuda.synchronize(); print("replay raw", i, float(step._static_losses.total.detach()), flush=True)
        sys.exit(0)
    for i in range(4):
        ld = step(batches[0 if os.environ.get('DBG_SAME') else i % 2]); torch.cuda.synchronize()
        fl = step.optimizer.flat
        print("replay", i, float(ld.total.detach()), "gradnorm", float(step.optimizer.grad_norm()),
              "finite", [bool(torch.isfinite(g.param).all()) for g in fl.groups], "dyn", step.optimizer._dyn_dev[:, :3].tolist(),
              "pmax", [float(g.param.abs().max()) for g in fl.groups], flush=True)
