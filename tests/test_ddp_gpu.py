"""GPU test of the data-parallel training step: 2 ranks sharing the one GPU over gloo (RCCL refuses two ranks on
one device; the hooks / side-stream / multi-tensor gather path is the same).  The all-reduced flat gradients must
equal the mean of the two single-process gradients computed on the same batches with the same random points."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg():
    from partdistillation_amd.config import setup_cfg
    return setup_cfg(os.path.join(ROOT, "partdistillation_amd", "configs", "proposal_learning", "r50_mask2former.yaml"),
                     ["MODEL.MASK_FORMER.NUM_OBJECT_QUERIES", "20", "MODEL.MASK_FORMER.DEC_LAYERS", "3",
                      "MODEL.SEM_SEG_HEAD.TRANSFORMER_ENC_LAYERS", "1", "MODEL.MASK_FORMER.TRAIN_NUM_POINTS", "256",
                      "SOLVER.AMP.ENABLED", "False", "SOLVER.BASE_LR", "0.0", "MODEL.AMD.DDP_BUCKET_MB", "8"])


def _grads(step, batch, seed):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import common as C
    step.model.criterion.rand = C.ReplayRand(seed)
    step(batch)
    return {n: g._view(g.grad, p, off).detach().float().cpu().clone()
            for g in step.optimizer.flat.groups for n, p, off in zip(g.names, g.params, g.offsets)}


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from partdistillation_amd.engine.synthetic import make_batch
    from partdistillation_amd.engine.trainer import TrainStep
    torch.manual_seed(123)
    step = TrainStep(_cfg())
    assert step.world == 2 and len(step.reducer.buckets) >= 2
    batch = make_batch(1, 128, n_parts=3, seed=40 + rank, device="cuda")
    g = _grads(step, batch, 900 + rank)
    torch.save(g, os.path.join(tmp, f"ddp{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_step_matches_single_process_mean(tmp_path):
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    d0, d1 = torch.load(tmp_path / "ddp0.pt"), torch.load(tmp_path / "ddp1.pt")
    for n in d0:
        torch.testing.assert_close(d0[n], d1[n], rtol=0, atol=0)
    sys.path.insert(0, ROOT)
    from partdistillation_amd.engine.synthetic import make_batch
    from partdistillation_amd.engine.trainer import TrainStep
    singles = []
    for rank in range(2):
        torch.manual_seed(123)
        step = TrainStep(_cfg())
        singles.append(_grads(step, make_batch(1, 128, n_parts=3, seed=40 + rank, device="cuda"), 900 + rank))
    checked = 0
    for n in d0:
        want = 0.5 * (singles[0][n] + singles[1][n])
        scale = want.abs().max().clamp_min(1e-9)
        # head / pixel decoder: fp32 atomics re-ordering only.  Backbone convolution weights additionally depend on
        # WHICH MIOpen weight-gradient algorithm each process happened to pick (measured with tools/debug_ddp.py: two
        # single-process runs of the same step differ by up to 5e-2 of the tensor scale on res4 convolutions, identically
        # with the fused cores switched off); a wrong reduction would be off by O(1) (sum instead of mean, missed bucket)
        tol = 1e-1 if n.startswith("backbone.") else 3e-3
        assert ((d0[n] - want).abs().max() / scale).item() < tol, n
        checked += 1
    assert checked > 100


@pytest.mark.timeout(900)
def test_rccl_two_gpu_bench_line():
    """the RCCL path itself (backend "nccl" over xGMI), through the driver's own launch line for bench.py --gpus 2.  Needs two
    visible GPUs: the single-GPU boxes of the development pool skip it, the driver's multi-GPU node runs it."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs for the nccl (RCCL) backend")
    import json
    import subprocess
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--size", "256"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=800,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, out.stdout[-3000:]
    rec = json.loads(lines[-1])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 4 and rec["scaling"] == "weak"
    assert rec["value"] > 0 and rec["config"]["final_total_loss"] == rec["config"]["final_total_loss"]


@pytest.mark.timeout(900)
def test_bench_two_ranks_sharing_the_gpu_over_gloo():
    """bench.py's N > 1 path (rank / world from the environment, per-rank batches, barrier + max-over-ranks timing, one JSON line
    from rank 0, weak scaling) through the driver's launch line, with both ranks on the one GPU of the development box over gloo
    (bench.py's PD_TEST_SHARE_GPU hook; RCCL refuses two ranks on one device)."""
    import json
    import subprocess
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--size", "256"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=800,
                         env=dict(os.environ, PD_TEST_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-3000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 4 and rec["scaling"] == "weak" and rec["steps"] == 3
    assert rec["value"] > 0 and rec["config"]["final_total_loss"] == rec["config"]["final_total_loss"]
    assert "cpu_baseline" not in rec and "parity" not in rec                     # N = 1 only


# ----------------------------------------------------------------------------- part distillation: row-sparse class head on the GPU
def _pd_cfg():
    from partdistillation_amd.config import setup_cfg
    return setup_cfg(os.path.join(ROOT, "partdistillation_amd", "configs", "part_distillation", "swinb_mask2former.yaml"),
                     ["MODEL.SWIN.EMBED_DIM", "32", "MODEL.SWIN.DEPTHS", "[2, 2, 2, 2]", "MODEL.SWIN.NUM_HEADS", "[2, 2, 4, 4]",
                      "MODEL.SWIN.WINDOW_SIZE", "4", "MODEL.SWIN.DROP_PATH_RATE", "0.0", "MODEL.MASK_FORMER.NUM_OBJECT_QUERIES", "20",
                      "MODEL.MASK_FORMER.DEC_LAYERS", "3", "MODEL.SEM_SEG_HEAD.TRANSFORMER_ENC_LAYERS", "1",
                      "MODEL.MASK_FORMER.TRAIN_NUM_POINTS_MATCH", "256", "MODEL.MASK_FORMER.TRAIN_NUM_POINTS_LOSS", "256",
                      "PART_DISTILLATION.NUM_OBJECT_CLASSES", "50", "SOLVER.AMP.ENABLED", "False", "SOLVER.BASE_LR", "0.0",
                      "MODEL.AMD.DDP_BUCKET_MB", "8"])


def _pd_batch(rank):
    from partdistillation_amd.engine.synthetic import make_batch
    return make_batch(2, 128, n_parts=3, seed=60 + rank, device="cuda", part_distillation=True, num_part_classes=8, num_object_classes=50)


def _pd_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from partdistillation_amd.engine.trainer import TrainStep
    torch.manual_seed(321)
    step = TrainStep(_pd_cfg())
    assert len(step.reducer.sparse_groups) == 1                                   # the float64 class head
    g = _grads(step, _pd_batch(rank), 950 + rank)
    torch.save({k: v for k, v in g.items() if "class_embed" in k or "query_feat" in k or "mask_embed.layers.0" in k}, os.path.join(tmp, f"pd{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_part_distillation_row_sparse_class_head(tmp_path):
    """PartDistillationModel, 2 ranks on the GPU over gloo: the fp64 class head's gradient goes through the row-sparse exchange
    (engine/ddp.py) on CUDA tensors and must equal the mean of the two single-process gradients — non-zero exactly in the rows
    of the four images' object classes (+ the no-object row)."""
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    mp.spawn(_pd_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    d0, d1 = torch.load(tmp_path / "pd0.pt"), torch.load(tmp_path / "pd1.pt")
    for n in d0:
        torch.testing.assert_close(d0[n], d1[n], rtol=0, atol=0)
    sys.path.insert(0, ROOT)
    from partdistillation_amd.engine.trainer import TrainStep
    singles, classes = [], set()
    for rank in range(2):
        torch.manual_seed(321)
        step = TrainStep(_pd_cfg())
        batch = _pd_batch(rank)
        classes |= {int(b["gt_object_class"]) for b in batch}
        singles.append(_grads(step, batch, 950 + rank))
    for n in d0:
        want = 0.5 * (singles[0][n] + singles[1][n])
        scale = want.abs().max().clamp_min(1e-12)
        assert ((d0[n] - want).abs().max() / scale).item() < 3e-3, n
    w = d0["sem_seg_head.predictor.class_embed.weight"]
    assert w.dtype == torch.float32 or w.dtype == torch.float64
    rows = sorted((w.abs().sum(1) > 0).nonzero().flatten().tolist())
    assert rows == sorted({c * 8 + k for c in classes for k in range(8)} | {400}), (rows, classes)


def _rccl_worker(rank, world, port, tmp, part):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PD_DDP_FORCE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    from partdistillation_amd.engine.trainer import TrainStep
    from partdistillation_amd.engine.synthetic import make_batch
    torch.manual_seed(321 if part else 123)
    step = TrainStep(_pd_cfg() if part else _cfg())
    assert step.world == 1 and step.reducer.active and len(step.reducer._hooks) > 100
    assert len(step.reducer.sparse_groups) == (1 if part else 0)
    batch = _pd_batch(0) if part else make_batch(1, 128, n_parts=3, seed=40, device="cuda")
    g = _grads(step, batch, 950 if part else 900)
    g2 = _grads(step, batch, 951 if part else 901)                      # a second step: the buckets re-arm, the pending counts reset
    torch.cuda.synchronize()
    torch.save((g, g2), os.path.join(tmp, "rccl.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("part", [False, True])
def test_collectives_through_rccl_with_one_rank(tmp_path, part):
    """Every collective of the data-parallel step through the RCCL backend ("nccl") on this one GPU: PD_DDP_FORCE=1 makes a one-rank job
    issue the bucket all-reduces (ReduceOp.AVG on the side stream, hooks firing during backward), the parameter broadcast, the criterion's
    early num_masks all-reduce and — for the part-distillation model — the static row-sparse exchange of the fp64 class head.  With one
    participant the results must equal the plain single-process step's; what the test buys is that the RCCL call sequence (ops, dtypes,
    streams, async handles) has run before a multi-GPU node sees it."""
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    mp.spawn(_rccl_worker, args=(1, port, str(tmp_path), part), nprocs=1, join=True)
    got, got2 = torch.load(tmp_path / "rccl.pt")
    sys.path.insert(0, ROOT)
    from partdistillation_amd.engine.synthetic import make_batch
    from partdistillation_amd.engine.trainer import TrainStep
    torch.manual_seed(321 if part else 123)
    step = TrainStep(_pd_cfg() if part else _cfg())
    assert not step.reducer.active
    batch = _pd_batch(0) if part else make_batch(1, 128, n_parts=3, seed=40, device="cuda")
    want = _grads(step, batch, 950 if part else 900)
    want2 = _grads(step, batch, 951 if part else 901)
    checked = 0
    for a, b in ((got, want), (got2, want2)):
        for n in b:
            scale = b[n].abs().max().clamp_min(1e-12)
            tol = 1e-1 if n.startswith("backbone.") else 3e-3            # (the tolerances of the two-rank test above, and why)
            assert ((a[n] - b[n]).abs().max() / scale).item() < tol, n
            checked += 1
    assert checked > 200
