import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(ROOT, 'partdistillation_amd', 'miopen_db'))
import torch
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_graph_gpu import _cfg, _state
from partdistillation_amd.engine.synthetic import make_batch
from partdistillation_amd.engine.trainer import TrainStep
batches = [make_batch(2, 128, n_parts=3, seed=40 + i, device="cuda") for i in range(4)]
steps = {}
for name in ("eager", "graph", "eager2"):
    torch.manual_seed(0)
    steps[name] = TrainStep(_cfg())
    for i in range(2):
        torch.cuda.manual_seed(100 + i)
        steps[name](batches[i])
g, e, e2 = steps["graph"], steps["eager"], steps["eager2"]
print("state diff before capture", max(float((a.float() - b.float()).abs().max()) for a, b in zip(_state(e), _state(g))))
g.capture(batches[0])
for i in range(4):
    out = {}
    for name, s in (("eager", e), ("graph", g), ("eager2", e2)):
        torch.cuda.manual_seed(200 + i)
        ld = s(batches[(i + 1) % 4])
        out[name] = {k: float(v) for k, v in ld.items()}
        torch.cuda.synchronize()
    for other in ("graph", "eager2"):
        worst = max(out["eager"], key=lambda k: abs(out["eager"][k] - out[other][k]) / (abs(out["eager"][k]) + 1e-3))
        print(i, other, "worst", worst, out["eager"][worst], out[other][worst],
              "state diff", max(float((a.float() - b.float()).abs().max()) for a, b in zip(_state(e), _state(steps[other]))), flush=True)
print("---- replay vs eager from IDENTICAL state every step")
for i in range(4, 10):
    with torch.no_grad():
        for a, b in zip(e._flat_state(), g._flat_state()):
            b.copy_(a)
    g.optimizer.steps = e.optimizer.steps
    out = {}
    for name, s in (("eager", e), ("graph", g)):
        torch.cuda.manual_seed(200 + i)
        ld = s(batches[(i + 1) % 4])
        out[name] = {k: float(v) for k, v in ld.items()}
        torch.cuda.synchronize()
    ks = sorted(out["eager"])
    print(i, " ".join(f"{k}:{out['eager'][k]:.3f}/{out['graph'][k]:.3f}" for k in ks), flush=True)
print("---- state after ONE step from identical state")
names = []
for gg in e.optimizer.flat.groups:
    names.append(("param", gg))
for gg in e.optimizer.flat.groups:
    if gg.shadow is not None:
        names.append(("shadow", gg))
names += [("exp_avg", gg) for gg in e.optimizer.flat.groups] + [("exp_avg_sq", gg) for gg in e.optimizer.flat.groups]
for i in range(10, 13):
    with torch.no_grad():
        for a, b in zip(e._flat_state(), g._flat_state()):
            b.copy_(a)
    g.optimizer.steps = e.optimizer.steps
    prev = _state(e)
    for name, s in (("eager", e), ("graph", g)):
        torch.cuda.manual_seed(200 + i)
        s(batches[(i + 1) % 4])
        torch.cuda.synchronize()
    print("step", i, "sumsq", float(e.optimizer._sumsq), float(g.optimizer._sumsq), "dyn", e.optimizer._dyn_dev[0].tolist(), g.optimizer._dyn_dev[0].tolist())
    for (nm, gg), p0, a, b in zip(names, prev, _state(e), _state(g)):
        upd = float((a.float() - p0.float()).abs().max())
        dif = (a.float() - b.float()).abs()
        j = int(dif.argmax())
        # which parameter holds the worst element
        t = max(k for k, off in enumerate(gg.offsets) if off <= j)
        print(f"   {nm:10s} n={a.numel():9d} max|update|={upd:.3e} max|eager-graph|={float(dif.max()):.3e} at {gg.names[t]} frac_diff>{1e-6}: {float((dif > 1e-6).float().mean()):.4f}", flush=True)
