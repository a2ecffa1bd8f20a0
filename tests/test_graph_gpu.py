"""GPU tests of the hipGraph-captured training step (engine/trainer.py TrainStep.capture / _replay): what a replay
computes equals the eager step on the same weights, inputs and RNG state — also with eager steps of a second trainer
and of mismatching batches issued between the replays; capturing does not move the training trajectory.

capture() insists on DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (ROCm 7.2 packet-captured graphs go stale after eager launches),
and the HIP runtime reads that flag at its first call, so the checks run in a fresh interpreter: this file is its
own child script (``python tests/test_graph_gpu.py <check>``)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(extra=()):
    from partdistillation_amd.config import setup_cfg
    return setup_cfg(os.path.join(ROOT, "partdistillation_amd", "configs", "proposal_learning", "r50_mask2former.yaml"),
                     ["MODEL.MASK_FORMER.NUM_OBJECT_QUERIES", "20", "MODEL.MASK_FORMER.DEC_LAYERS", "4",
                      "MODEL.SEM_SEG_HEAD.TRANSFORMER_ENC_LAYERS", "2", "MODEL.MASK_FORMER.TRAIN_NUM_POINTS", "256",
                      "SOLVER.AMP.ENABLED", "True", "SOLVER.BASE_LR", "0.0001", "SOLVER.WARMUP_ITERS", "0",
                      "SOLVER.CLIP_GRADIENTS.CLIP_VALUE", "0.1"] + list(extra))


def _state(step):
    return [t.detach().clone() for t in step._flat_state()]


def check_replay_equals_eager():
    from partdistillation_amd.engine.synthetic import make_batch
    from partdistillation_amd.engine.trainer import TrainStep
    batches = [make_batch(2, 128, n_parts=3, seed=40 + i, device=DEV) for i in range(4)]
    steps = {}
    for name in ("eager", "graph"):
        torch.manual_seed(0)
        steps[name] = TrainStep(_cfg())
        for i in range(2):                                           # same two eager steps in both
            torch.cuda.manual_seed(100 + i)
            steps[name](batches[i])
    g, e = steps["graph"], steps["eager"]
    before = _state(g)
    n0 = g.optimizer.steps
    g.capture(batches[0])
    assert g.optimizer.steps == n0
    for a, b in zip(before, _state(g)):
        assert torch.equal(a, b)                                     # capture left weights, bf16 copies, moments bit-identical
    for i in range(8):
        # same state, same batch, same philox seed/offset in both -> the replayed step must do what the eager step does.
        # (The toy model is chaotic — two EAGER runs drift apart by 5 % in a few steps through reordered fp32 atomics — so
        # the graph trainer is re-seated on the eager one's state before every step instead of comparing trajectories.)
        with torch.no_grad():
            for a, b in zip(e._flat_state(), g._flat_state()):
                b.copy_(a)
        g.optimizer.steps = e.optimizer.steps
        prev = _state(e)
        out = {}
        for name, s in (("eager", e), ("graph", g)):                 # e's ~1 500 eager launches sit between g's replays
            torch.cuda.manual_seed(200 + i)
            ld = s(batches[(i + 1) % 4])
            out[name] = {k: float(v) for k, v in ld.items()}
        assert g._graph is not None
        assert set(out["eager"]) == set(out["graph"]) and len(out["eager"]) >= 12
        dev = {k: abs(v - out["graph"][k]) / (abs(v) + 1e-3) for k, v in out["eager"].items()}
        # identical inputs; what differs is the order of fp32 atomic sums, which now and then flips a near-tied Hungarian pair or an
        # importance-sampled point in one head of this chaotic toy model (typical: 11 of 12 losses within 0.3 %, one at 1.2 %; about one
        # run in ten had a term beyond 20 %).  The bounds ASSERTED here are therefore those of the failure this test exists for — a stale
        # or wrong graph gave NaN gradient norms, losses off by factors and garbage weights, every time — and a run outside the
        # statistical bounds is reported as a warning that pytest counts instead of being retried.
        assert all(v == v and abs(v) < 1e4 for v in out["graph"].values()), (i, out["graph"])
        assert max(dev.values()) <= 1.0 and sorted(dev.values())[len(dev) // 2] <= 0.1, (i, dev)
        ne, ng = float(e.optimizer.grad_norm()), float(g.optimizer.grad_norm())
        assert ne > 0 and ng == ng and 0.5 * ne <= ng <= 2.0 * ne, (i, ne, ng)
        if max(dev.values()) > 0.2 or sorted(dev.values())[len(dev) // 2] > 3e-2 or abs(ne - ng) > 0.3 * ne:
            import warnings
            warnings.warn(f"replay vs eager step {i} outside the statistical bounds (a flipped near-tie): max loss dev {max(dev.values()):.3f}, "
                          f"gradient norms {ne:.4f} / {ng:.4f}")
        for p0, a, b in zip(prev, _state(e), _state(g)):
            upd = float((a.float() - p0.float()).abs().max())
            assert torch.isfinite(b.float()).all()
            # one clipped AdamW step moves a weight by ~lr whatever the gradient's size: the replayed step's largest move is bounded
            # by a small multiple of the eager step's (a stale graph wrote garbage here)
            assert float((b.float() - p0.float()).abs().max()) <= 4.0 * upd + 1e-12, (i, upd)
    assert g.optimizer.steps == e.optimizer.steps and g.iter == e.iter


def check_mismatching_batches_run_eagerly_between_replays():
    from partdistillation_amd.engine.synthetic import make_batch
    from partdistillation_amd.engine.trainer import TrainStep
    torch.manual_seed(0)
    step = TrainStep(_cfg())
    a = make_batch(2, 128, n_parts=3, seed=1, device=DEV)
    b = make_batch(2, 128, n_parts=4, seed=2, device=DEV)            # one more target mask per image: another signature
    assert step._signature(a) != step._signature(b)
    step(a)
    step.capture(a)
    hist = []
    for i in range(24):                                              # replay, 3 eager steps, replay, ...: the order that faulted
        ld = step(a if i % 4 == 0 else b)
        hist.append(float(sum(v.detach() for v in ld.values())))
    torch.cuda.synchronize()
    assert step._graph is not None and all(x == x and abs(x) < 1e4 for x in hist), hist
    assert sum(hist[-4:]) < sum(hist[:4]), hist                       # and it trains


def check_capture_refuses_packet_capture():
    from partdistillation_amd.engine.synthetic import make_batch
    from partdistillation_amd.engine.trainer import TrainStep
    torch.manual_seed(0)
    step = TrainStep(_cfg())
    # the variable is read when the module is imported (the HIP runtime reads it at initialisation: a later os.environ change
    # would pass a check and change nothing), so this child is started WITHOUT the safe setting
    assert os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") != "0"
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"              # too late: must still refuse
    with pytest.raises(RuntimeError, match="DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"):
        step.capture(make_batch(2, 128, n_parts=3, seed=1, device=DEV))


def _child(check, packet_capture="0"):
    # PD_CMDBUF=0: the whole-step graph and the command buffers are alternatives (TrainStep.capture refuses to run after a recording)
    env = dict(os.environ, DEBUG_CLR_GRAPH_PACKET_CAPTURE=packet_capture, PD_CMDBUF="0")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), check], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def test_replay_equals_eager_and_capture_keeps_the_trajectory():
    _child("check_replay_equals_eager")


def test_mismatching_batches_run_eagerly_between_replays():
    _child("check_mismatching_batches_run_eagerly_between_replays")


def test_capture_refuses_packet_capture():
    _child("check_capture_refuses_packet_capture", packet_capture="1")


def test_object_class_is_part_of_the_signature_only_where_the_step_reads_it():
    from partdistillation_amd.engine.synthetic import make_batch
    from partdistillation_amd.engine.trainer import TrainStep
    torch.manual_seed(0)
    step = TrainStep(_cfg())
    a = make_batch(2, 128, n_parts=3, seed=1, device=DEV)
    b = make_batch(2, 128, n_parts=3, seed=2, device=DEV)
    assert [x["gt_object_class"] for x in a] != [x["gt_object_class"] for x in b]
    assert step._signature(a) == step._signature(b)                  # the proposal model never looks at the object class
    assert not getattr(step.model, "host_reads_object_class", False)
    from partdistillation_amd.part_distillation_model import PartDistillationModel
    assert PartDistillationModel.host_reads_object_class             # its decoder selects class-head rows on the host


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    globals()[sys.argv[1]]()
    torch.cuda.synchronize()
    print("ok", sys.argv[1])
