"""list every host->device (and device->host) copy one training step issues, with pinned/pageable source and the python
line that issued it.  A pageable source inside a hipGraph-captured region is a bug: the memcpy node re-reads host memory
that was a temporary."""
import os, sys, traceback, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_flatten
sys.path.insert(0, ROOT)
from partdistillation_amd.config import setup_cfg
from partdistillation_amd.engine.synthetic import make_batch
from partdistillation_amd.engine.trainer import TrainStep

found = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = func(*args, **kwargs)
        ins = [a for a in tree_flatten((args, kwargs))[0] if isinstance(a, torch.Tensor)]
        outs = [a for a in tree_flatten(out)[0] if isinstance(a, torch.Tensor)]
        cpu_in = [a for a in ins if a.device.type == "cpu"]
        cuda_any = any(a.is_cuda for a in ins + outs)
        if cpu_in and cuda_any:
            frames = [f for f in traceback.extract_stack() if "partdistillation_amd" in f.filename]
            where = f"{os.path.relpath(frames[-1].filename, ROOT)}:{frames[-1].lineno}" if frames else "?"
            pinned = all(a.is_pinned() for a in cpu_in if a.numel() > 0)
            found[(str(func), "pinned" if pinned else "PAGEABLE", where, tuple(cpu_in[0].shape))] += 1
        return out


which = sys.argv[1] if len(sys.argv) > 1 else "proposal"
if which == "proposal":
    cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd", "configs", "proposal_learning", "r50_mask2former.yaml"), [])
    pd = False
else:
    cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd", "configs", "part_distillation", "swinb_IN21k_384_mask2former.yaml"), [])
    pd = True
torch.manual_seed(0)
step = TrainStep(cfg)
size = 512
batches = [make_batch(2, size, n_parts=4, seed=5 + i, device="cuda", part_distillation=pd) for i in range(2)]
for i in range(3):
    step(batches[i % 2])
torch.cuda.synchronize()
with Spy():
    step._forward_backward(batches[0])
    step.optimizer.launch_step()
torch.cuda.synchronize()
for k, n in sorted(found.items(), key=lambda kv: kv[0][2]):
    print(n, *k)
print("total host<->device ops:", sum(found.values()))
