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


# ----------------------------------------------------------------------------- row-sparse exchange when the ranks disagree
def _uneven_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from partdistillation_amd.engine.ddp import BucketedGradReducer, broadcast_parameters
    from partdistillation_amd.engine.flat_params import FlatParams
    torch.manual_seed(5)
    model = SparseNet()
    names = dict((id(p), n) for n, p in model.named_parameters())
    groups = [{"params": [p for p in reversed(list(model.parameters())) if p.dtype == dt],
               "names": [names[id(p)] for p in reversed(list(model.parameters())) if p.dtype == dt], "lr": 1e-3, "weight_decay": 0.0}
              for dt in (torch.float32, torch.float64)]
    flat = FlatParams(groups)
    broadcast_parameters(flat, 0)
    reducer = BucketedGradReducer(flat, bucket_mb=0.00005)
    torch.manual_seed(60 + rank)
    out = {"params": {n: p.detach().clone() for n, p in model.named_parameters()}, "steps": []}
    # step 0: rank 1 saw a smaller last batch (4 rows instead of 8); step 1: rank 1 has NO row record (its step skipped the
    # head's bookkeeping) while rank 0 has one; step 2: a stale record must not survive (both ranks set fresh rows again)
    plans = [(torch.tensor([3, 4, 5, 39, 7, 8, 9, 39]), torch.tensor([17, 18, 19, 39])),
             (torch.tensor([1, 2, 39]), None),
             (torch.tensor([30, 31, 39]), torch.tensor([30, 32, 39]))]
    for r0, r1 in plans:
        rows = r0 if rank == 0 else r1
        x = torch.randn(4, 5)
        flat.zero_grad()
        use = rows if rows is not None else torch.tensor([11, 12, 39])
        model(x, use, use_side=True).backward()
        if rows is None:
            model.head.weight._pd_rows = model.head.bias._pd_rows = None       # this rank kept no record
        reducer.finish()
        assert getattr(model.head.weight, "_pd_rows", None) is None             # consumed
        grads = {n: g._view(g.grad, p, off).detach().clone() for g in flat.groups for n, p, off in zip(g.names, g.params, g.offsets)}
        out["steps"].append({"x": x, "rows": use, "grads": grads})
    torch.save(out, os.path.join(tmp, f"uneven{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_row_sparse_exchange_uneven_rows_and_missing_record(tmp_path):
    """the ranks agree on ONE exchange path before any size-dependent collective: different row counts are padded
    with a sentinel, a rank without a row record sends every rank down the dense all-reduce, and a record is consumed by the
    step that used it.  In all three cases the result equals the dense mean and is identical on both ranks."""
    world = 2
    mp.spawn(_uneven_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"uneven{k}.pt") for k in range(world)]
    for step in range(3):
        for n in r[0]["steps"][step]["grads"]:
            torch.testing.assert_close(r[0]["steps"][step]["grads"][n], r[1]["steps"][step]["grads"][n], rtol=0, atol=0)
        model = SparseNet()
        model.load_state_dict(r[0]["params"])
        loss = sum(model(r[k]["steps"][step]["x"], r[k]["steps"][step]["rows"], use_side=True) for k in range(world)) / world
        loss.backward()
        for n, p in model.named_parameters():
            torch.testing.assert_close(r[0]["steps"][step]["grads"][n], p.grad, rtol=1e-6, atol=1e-7, msg=lambda m: f"step {step} {n}: {m}")


# ----------------------------------------------------------------------------- more touched rows than the cap: all ranks fail TOGETHER
def _overflow_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from partdistillation_amd.engine.ddp import BucketedGradReducer, broadcast_parameters
    from partdistillation_amd.engine.flat_params import FlatParams
    torch.manual_seed(5)
    model = SparseNet()
    names = dict((id(p), n) for n, p in model.named_parameters())
    groups = [{"params": [p for p in reversed(list(model.parameters())) if p.dtype == dt],
               "names": [names[id(p)] for p in reversed(list(model.parameters())) if p.dtype == dt], "lr": 1e-3, "weight_decay": 0.0}
              for dt in (torch.float32, torch.float64)]
    flat = FlatParams(groups)
    broadcast_parameters(flat, 0)
    reducer = BucketedGradReducer(flat, bucket_mb=0.00005, sparse_rows_cap=4)
    # step 0: ONLY rank 1's record (6 rows) exceeds the cap of 4.  Nobody may raise before the collectives (the other rank would hang in
    # all_gather); step 1: BOTH ranks raise at the check that opens their exchange, with the same message
    plans = [(torch.tensor([3, 4, 39]), torch.tensor([17, 18, 19, 20, 21, 39])), (torch.tensor([1, 39]), torch.tensor([2, 39]))]
    raised = []
    for it, (r0, r1) in enumerate(plans):
        flat.zero_grad()
        model(torch.randn(4, 5), r0 if rank == 0 else r1, use_side=True).backward()
        try:
            reducer.finish()
            raised.append(None)
        except RuntimeError as e:
            raised.append(str(e))
    torch.save(raised, os.path.join(tmp, f"overflow{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_row_sparse_overflow_is_raised_on_every_rank_together(tmp_path):
    """a rank whose step touched more rows than DDP_SPARSE_ROWS_CAP must not raise before its collectives (the others would
    block in all_gather until the process group's timeout): the flag travels with the gathered rows and every rank raises at the
    same point of the next step."""
    world = 2
    mp.spawn(_overflow_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"overflow{k}.pt") for k in range(world)]
    for k in range(world):
        assert r[k][0] is None, r[k]
        assert r[k][1] is not None and "DDP_SPARSE_ROWS_CAP = 4" in r[k][1], r[k]


# ----------------------------------------------------------------------------- bucket issue schedule on the real config-2 parameter list
def test_bucket_issue_schedule_config2():
    """VERDICT r2 item 7a.  The reducer issues buckets strictly in index order over ALL flat groups, so a group whose parameters are
    ready early could sit behind one that also holds late parameters.  On the real BASELINE config-2 parameter list (R50 proposal
    model, 44 M parameters, three flat groups) fire the gradient hooks in the order autograd produces them — the reverse of the
    forward: criterion -> decoder core -> pixel decoder (mask_features, FPN level, encoder core, input projections) -> backbone
    res5 .. stem — and check (1) buckets leave in index order, (2) a bucket leaves at the hook that completes it or, at worst,
    the one that completes its predecessor (no bucket waits for unrelated late parameters), (3) >= 80 % of the gradient bytes are on
    the wire before the last hook fires."""
    from partdistillation_amd.compat import build_model
    from partdistillation_amd.config import setup_cfg
    from partdistillation_amd.engine.ddp import BucketedGradReducer
    from partdistillation_amd.engine.optimizer import build_optimizer
    import partdistillation_amd.modeling, partdistillation_amd.proposal_model  # noqa: F401,E401
    cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd", "configs", "proposal_learning", "r50_mask2former.yaml"), ["MODEL.DEVICE", "cpu"])
    model = build_model(cfg)
    flat = build_optimizer(cfg, model).flat
    reducer = BucketedGradReducer(flat, bucket_mb=cfg.MODEL.AMD.DDP_BUCKET_MB)
    order = {id(p): i for i, (n, p) in enumerate(model.named_parameters())}
    params = sorted((p for g in flat.groups for p in g.params), key=lambda p: -order[id(p)])        # reverse registration order
    assert sum(p.numel() for p in params) > 40e6 and len(reducer.buckets) >= 6
    log, fired = [], [0]
    reducer._launch = lambda b: log.append((b.index, (b.end - b.start) * flat.groups[b.group].grad.element_size(), fired[0]))
    ready_at = {}
    for i, p in enumerate(params):
        fired[0] = i + 1
        b = reducer._param_bucket[p]
        ready_at[b.index] = i + 1                                                # the hook count at which bucket b became complete
        reducer._on_grad(p)
    assert [e[0] for e in log] == sorted(e[0] for e in log)                      # (1)
    assert len(log) == len(reducer.buckets), "every bucket is complete once every hook has fired"
    issued_at = {idx: at for idx, _, at in log}
    for idx in issued_at:                                                        # (2)
        assert issued_at[idx] == max(ready_at[j] for j in range(idx + 1)), (idx, issued_at[idx], ready_at[idx])
        late = issued_at[idx] - ready_at[idx]
        assert late <= 0.02 * len(params) or idx == 0, f"bucket {idx} waited {late} hooks behind an earlier bucket"
    total = sum(e[1] for e in log)
    before_last = sum(e[1] for e in log if e[2] < len(params))
    print(f"{len(log)} buckets, {total / 1e6:.1f} MB; {before_last / total:.1%} of the bytes issued before the last hook; issue points "
          f"{[round(e[2] / len(params), 2) for e in log]}")
    assert before_last >= 0.8 * total                                            # (3)


def test_bucket_issue_schedule_config2_at_node_granularity():
    """VERDICT r5 item 8.  The schedule test above fires one hook per parameter; in the product the fused cores are ONE autograd node each
    (functions/decoder_core.py DecoderCore, functions/encoder_core.py EncoderCore: their weight gradients are grouped launches at the end of
    the node's backward, so all of a node's gradients arrive when it returns) and the fused ResNet body publishes per stage
    (resnet_core.py -> BucketedGradReducer.publish).  Here the hooks fire at THAT granularity — a node's parameters all at once, nodes in
    backward order — and each event carries the share of the backward pass's kernel time that has elapsed when it fires (config 2 at
    1024 x 1024, profiles/r06_bench_n1_*_steady_kernel_stats.csv: criterion 0.7 ms, decoder core 2.9, mask features / FPN 1.5, encoder core
    5.6, input projections 0.3, res5 0.35, res4 0.55, res3 0.45, res2 + stem 0.9 of a 13.3 ms backward).  Checked: buckets leave in index
    order; >= 80 % of the gradient bytes are issued before the LAST node returns (measured 86.3 %: the last bucket, 24 MB, is the stem's).
    What the granularity costs is visible in the second figure: only 38.9 % of the bytes are on the wire with a fifth of the backward
    still to run, because the bucket that holds the encoder's 6 M parameters cannot leave before the encoder core returns at 0.80 and
    every later bucket queues behind it (index order) — per-layer hand-over from inside EncoderCore.backward is what would move it,
    at the price of splitting its two grouped weight-gradient launches."""
    from partdistillation_amd.compat import build_model
    from partdistillation_amd.config import setup_cfg
    from partdistillation_amd.engine.ddp import BucketedGradReducer
    from partdistillation_amd.engine.optimizer import build_optimizer
    import partdistillation_amd.modeling, partdistillation_amd.proposal_model  # noqa: F401,E401
    cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd", "configs", "proposal_learning", "r50_mask2former.yaml"), ["MODEL.DEVICE", "cpu"])
    model = build_model(cfg)
    flat = build_optimizer(cfg, model).flat
    reducer = BucketedGradReducer(flat, bucket_mb=cfg.MODEL.AMD.DDP_BUCKET_MB)
    named = [(n, p) for n, p in model.named_parameters() if p in reducer._param_bucket]

    def node_of(n):
        if ".predictor." in n:
            return "heads" if (".class_embed." in n or ".mask_embed." in n) else "decoder core"
        if ".pixel_decoder.transformer." in n:
            return "encoder core"
        if ".pixel_decoder.input_proj." in n:
            return "input projections"
        if ".pixel_decoder." in n:
            return "mask features / FPN"
        for st in ("res5", "res4", "res3", "res2", "stem"):
            if f"backbone.{st}." in n:
                return st
        raise AssertionError(n)

    # (node, elapsed share of the backward's kernel time when its gradients are handed over)
    schedule = [("heads", 0.05), ("decoder core", 0.27), ("mask features / FPN", 0.38), ("encoder core", 0.80), ("input projections", 0.83),
                ("res5", 0.86), ("res4", 0.90), ("res3", 0.93), ("res2", 0.97), ("stem", 1.00)]
    groups = {k: [p for n, p in named if node_of(n) == k] for k, _ in schedule}
    assert sum(len(v) for v in groups.values()) == len(named) and all(groups[k] for k, _ in schedule)
    log, now = [], [0.0]
    reducer._launch = lambda b: log.append((b.index, (b.end - b.start) * flat.groups[b.group].grad.element_size(), now[0]))
    for k, t in schedule:
        now[0] = t
        for p in groups[k]:
            reducer._on_grad(p)
    assert [e[0] for e in log] == sorted(e[0] for e in log) and len(log) == len(reducer.buckets)
    total = sum(e[1] for e in log)
    before_last = sum(e[1] for e in log if e[2] < 1.0)
    early = sum(e[1] for e in log if e[2] <= 0.80)
    print(f"node granularity: {len(log)} buckets, {total / 1e6:.1f} MB; issued before the last node returns {before_last / total:.1%}, with >= 20 % of the "
          f"backward still to run {early / total:.1%}; issue points {[e[2] for e in log]}")
    assert before_last >= 0.8 * total and early >= 0.35 * total


# ----------------------------------------------------------------------------- gradients handed over from INSIDE a fused node (per stage)
class _StagedBackward(torch.autograd.Function):
    """stand-in for the fused ResNet body (modeling/backbone/resnet_core.py): one autograd node over several layers that, in data-parallel
    runs, publishes each layer's weight gradient to the reducer from inside its backward — last layer first — and returns None for it"""

    @staticmethod
    def forward(ctx, x, publish, *ws):
        ctx.publish, ctx.ws = publish, ws
        acts = [x]
        for w in ws:
            acts.append(torch.tanh(acts[-1] @ w.t()))
        ctx.acts = acts
        return acts[-1]

    @staticmethod
    def backward(ctx, g):
        grads = [None] * len(ctx.ws)
        for i in reversed(range(len(ctx.ws))):
            g = g * (1 - ctx.acts[i + 1] ** 2)
            dw = g.t() @ ctx.acts[i]
            g = g @ ctx.ws[i]
            if ctx.publish is None or not ctx.publish(ctx.ws[i], dw):
                grads[i] = dw
        return (g, None, *grads)


def _staged_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from partdistillation_amd.engine.ddp import BucketedGradReducer, broadcast_parameters
    from partdistillation_amd.engine.flat_params import FlatParams
    from partdistillation_amd.modeling.backbone import resnet_core
    torch.manual_seed(3)
    ws = [torch.nn.Parameter(torch.randn(24, 24) * 0.3) for _ in range(4)]
    head = torch.nn.Parameter(torch.randn(5, 24) * 0.3)
    names = [f"body.{i}" for i in range(4)] + ["head"]
    flat = FlatParams([{"params": list(reversed(ws + [head])), "names": list(reversed(names)), "lr": 1e-3, "weight_decay": 0.0}])
    broadcast_parameters(flat, 0)
    reducer = BucketedGradReducer(flat, bucket_mb=0.002)
    assert len(reducer.buckets) >= 3 and resnet_core.PUBLISH is not None
    torch.manual_seed(20 + rank)
    x = torch.randn(6, 24)
    out = {}
    for mode in ("published", "hooks"):
        issued = []
        launch0 = reducer._launch
        reducer._launch = lambda b, _l=launch0: (issued.append(b.index), _l(b))[1]
        flat.zero_grad()
        y = _StagedBackward.apply(x, resnet_core.PUBLISH if mode == "published" else None, *ws)
        (y @ head.t()).pow(2).sum().backward()
        reducer.finish()
        reducer._launch = launch0
        g = flat.groups[0]
        out[mode] = ({n: g._view(g.grad, p, off).detach().clone() for n, p, off in zip(g.names, g.params, g.offsets)}, issued)
    torch.save(out, os.path.join(tmp, f"staged{rank}.pt"))
    reducer.remove()
    assert resnet_core.PUBLISH is None
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gradients_published_from_inside_a_fused_node_equal_the_hook_path(tmp_path):
    """the fused ResNet body hands its gradients to the reducer stage by stage from inside its backward
    (BucketedGradReducer.publish) instead of through autograd's hooks when the node returns.  The reduced gradients are identical to the
    hook path's on both ranks, every bucket is issued exactly once and in index order."""
    world = 2
    mp.spawn(_staged_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"staged{k}.pt") for k in range(world)]
    for k in range(world):
        for n in r[k]["published"][0]:
            torch.testing.assert_close(r[k]["published"][0][n], r[k]["hooks"][0][n], rtol=0, atol=0)
            torch.testing.assert_close(r[0]["published"][0][n], r[k]["published"][0][n], rtol=0, atol=0)
        for mode in ("published", "hooks"):
            idx = r[k][mode][1]
            assert idx == sorted(idx) and len(set(idx)) == len(idx) and len(idx) >= 3, (mode, idx)
