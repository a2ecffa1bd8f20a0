"""world_size-2 gloo tests (CPU) of the data-parallel path: the bucketed gradient reducer over the flat
gradient buffers must give every rank the mean of the per-rank gradients == the single-process gradient of the
concatenated batch / world, including parameters that receive no gradient, and broadcast_parameters must make the
replicas identical.  The scalar num_masks all-reduce of the criterion is covered too."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(7, 13)
        self.conv = torch.nn.Conv2d(3, 5, 3, padding=1)
        self.b = torch.nn.Linear(13, 3)
        self.unused = torch.nn.Linear(4, 4)          # never receives a gradient
        self.big = torch.nn.Linear(64, 300)           # forces several buckets at bucket_mb = 0.01

    def forward(self, x, img):
        h = self.b(torch.relu(self.a(x)))
        return h.sum() + self.conv(img).pow(2).mean() + self.big(x.repeat(1, 10)[:, :64]).tanh().sum()


def _entries(model):
    return [{"param": p, "name": n, "lr": 1e-3 if "big" not in n else 1e-4, "weight_decay": 0.0 if n.endswith("bias") else 0.05}
            for n, p in model.named_parameters()]


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from partdistillation_amd.engine.ddp import BucketedGradReducer, broadcast_parameters
    from partdistillation_amd.engine.flat_params import FlatParams
    from partdistillation_amd.modeling.criterion import SetCriterion
    torch.manual_seed(100 + rank)                    # different init per rank on purpose
    model = Net()
    groups = {}
    for e in reversed(_entries(model)):
        g = groups.setdefault((e["lr"], e["weight_decay"]), {"params": [], "names": [], "lr": e["lr"], "weight_decay": e["weight_decay"]})
        g["params"].append(e["param"]); g["names"].append(e["name"])
    flat = FlatParams(list(groups.values()))
    broadcast_parameters(flat, 0)
    reducer = BucketedGradReducer(flat, bucket_mb=0.01)
    assert len(reducer.buckets) >= 3
    torch.manual_seed(7)
    xs, imgs = torch.randn(2 * world, 7), torch.randn(2 * world, 3, 6, 6)
    for it in range(2):                               # two iterations: hooks/buckets must re-arm
        flat.zero_grad()
        model(xs[2 * rank:2 * rank + 2], imgs[2 * rank:2 * rank + 2]).backward()
        reducer.finish()
    crit = SetCriterion(1, None, {}, 0.1, [], 4, 3.0, 0.75)
    nm = crit.num_masks([{"labels": torch.zeros(3 + 2 * rank)}], torch.device("cpu"))
    flat_grads = {n: g._view(g.grad, p, off).detach().clone() for g in flat.groups for n, p, off in zip(g.names, g.params, g.offsets)}
    torch.save({"params": {n: p.detach().clone() for n, p in model.named_parameters()},
                "grads": flat_grads, "num_masks": nm,
                "xs": xs, "imgs": imgs}, os.path.join(tmp, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucketed_reducer_two_ranks_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    for n in r0["params"]:
        assert torch.equal(r0["params"][n], r1["params"][n]), f"replicas differ after broadcast: {n}"
        torch.testing.assert_close(r0["grads"][n], r1["grads"][n], rtol=0, atol=0)
    # single-process reference: same weights, per-rank losses summed / world
    model = Net()
    model.load_state_dict(r0["params"])
    xs, imgs = r0["xs"], r0["imgs"]
    loss = sum(model(xs[2 * r:2 * r + 2], imgs[2 * r:2 * r + 2]) for r in range(world)) / world
    loss.backward()
    for n, p in model.named_parameters():
        want = p.grad if p.grad is not None else torch.zeros_like(p)
        torch.testing.assert_close(r0["grads"][n], want, rtol=1e-5, atol=1e-6, msg=lambda m: f"{n}: {m}")
    # num_masks = clamp(sum_over_ranks / world, 1) (criterion.py:248-254): (3 + 5) / 2
    assert float(r0["num_masks"]) == 4.0 and float(r1["num_masks"]) == 4.0


# ----------------------------------------------------------------------------- row-sparse class head, prefetched num_masks, divergent graphs
class SparseNet(torch.nn.Module):
    """a float64 [rows, C] head of which a step uses a few rows (the part-distillation class head), a dense trunk, and a
    branch that only SOME ranks take in a step (data-dependent graph)"""

    def __init__(self, rows=40, c=6):
        super().__init__()
        self.trunk = torch.nn.Linear(5, c)
        self.side = torch.nn.Linear(5, c)
        self.head = torch.nn.Linear(c, rows).double()
        self.head.weight._pd_row_sparse = self.head.bias._pd_row_sparse = True

    def forward(self, x, rows, use_side):
        h = self.trunk(x)
        if use_side:
            h = h + self.side(x)
        self.head.weight._pd_rows = self.head.bias._pd_rows = rows
        w, b = self.head.weight[rows], self.head.bias[rows]
        return (h.double() @ w.t() + b).pow(2).sum()


def _sparse_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from partdistillation_amd.engine.ddp import BucketedGradReducer, broadcast_parameters
    from partdistillation_amd.engine.flat_params import FlatParams
    from partdistillation_amd.modeling.criterion import SetCriterion
    torch.manual_seed(5)
    model = SparseNet()
    names = dict((id(p), n) for n, p in model.named_parameters())
    groups = [{"params": [p for p in reversed(list(model.parameters())) if p.dtype == dt],
               "names": [names[id(p)] for p in reversed(list(model.parameters())) if p.dtype == dt], "lr": 1e-3, "weight_decay": 0.0}
              for dt in (torch.float32, torch.float64)]
    flat = FlatParams(groups)
    broadcast_parameters(flat, 0)
    reducer = BucketedGradReducer(flat, bucket_mb=0.00005)
    assert len(reducer.sparse_groups) == 1 and len(reducer.buckets) >= 2
    crit = SetCriterion(1, None, {}, 0.1, [], 4, 3.0, 0.75)
    torch.manual_seed(50 + rank)
    x = torch.randn(4, 5)
    # duplicates inside a rank (two images of one class, the shared last row) and rows shared across the ranks
    rows = torch.tensor([3, 4, 5, 39, 3, 4, 5, 39]) if rank == 0 else torch.tensor([17, 18, 19, 39, 3, 4, 5, 39])
    targets = [{"labels": torch.zeros(2 + 4 * rank)}]
    crit.prefetch_num_masks(targets, torch.device("cpu"))          # started before the "forward"
    flat.zero_grad()
    model(x, rows, use_side=(rank == 0)).backward()                # rank 1 never touches `side`: hooks fire for different sets
    reducer.finish()
    nm = crit.num_masks(targets, torch.device("cpu"))
    grads = {n: g._view(g.grad, p, off).detach().clone() for g in flat.groups for n, p, off in zip(g.names, g.params, g.offsets)}
    torch.save({"grads": grads, "x": x, "rows": rows, "num_masks": nm, "params": {n: p.detach().clone() for n, p in model.named_parameters()}},
               os.path.join(tmp, f"sparse{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_row_sparse_exchange_prefetched_num_masks_and_divergent_graphs(tmp_path):
    """(1) the float64 class head's gradient is exchanged as touched rows only and equals the dense mean (duplicated rows
    inside a rank and shared rows across ranks included); (2) buckets are issued in index order although rank 1's graph never
    reaches the `side` layer; (3) the num_masks all-reduce started before the forward is the one the criterion consumes."""
    world = 2
    mp.spawn(_sparse_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"sparse{k}.pt") for k in range(world)]
    for n in r[0]["grads"]:
        torch.testing.assert_close(r[0]["grads"][n], r[1]["grads"][n], rtol=0, atol=0)
    model = SparseNet()
    model.load_state_dict(r[0]["params"])
    loss = sum(model(r[k]["x"], r[k]["rows"], use_side=(k == 0)) for k in range(world)) / world
    loss.backward()
    for n, p in model.named_parameters():
        torch.testing.assert_close(r[0]["grads"][n], p.grad, rtol=1e-6, atol=1e-7, msg=lambda m: f"{n}: {m}")
    g = r[0]["grads"]["head.weight"]
    assert sorted((g.abs().sum(1) > 0).nonzero().flatten().tolist()) == [3, 4, 5, 17, 18, 19, 39]
    assert float(r[0]["num_masks"]) == 4.0 and float(r[1]["num_masks"]) == 4.0            # (2 + 6) / 2
