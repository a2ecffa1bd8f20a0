"""Development tool: the comparison of tests/test_ddp_gpu.py with switches (FUSED_DEC / FUSED_ENC = 0|1) and a report of
the worst parameters; also measures run-to-run noise of the single-process gradients."""
import os, sys, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.multiprocessing as mp
import test_ddp_gpu as T


def patch():
    from partdistillation_amd.modeling.transformer_decoder.mask2former_transformer_decoder import MultiScaleMaskedTransformerDecoder as D
    from partdistillation_amd.modeling.pixel_decoder.msdeformattn import MSDeformAttnTransformerEncoder as E
    if os.environ.get("FUSED_DEC", "1") == "0":
        D._core_dtype = lambda self, x: None
    if os.environ.get("FUSED_ENC", "1") == "0":
        E.fused_core = False


def worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    patch()
    T._worker(rank, world, port, tmp)


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-9)).item()


if __name__ == "__main__":
    import tempfile
    tmp = tempfile.mkdtemp()
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    mp.spawn(worker, args=(2, port, tmp), nprocs=2, join=True)
    d0 = torch.load(os.path.join(tmp, "ddp0.pt"))
    patch()
    from partdistillation_amd.engine.synthetic import make_batch
    from partdistillation_amd.engine.trainer import TrainStep
    runs = []
    for rep in range(2):
        singles = []
        for rank in range(2):
            torch.manual_seed(123)
            step = TrainStep(T._cfg())
            singles.append(T._grads(step, make_batch(1, 128, n_parts=3, seed=40 + rank, device="cuda"), 900 + rank))
        runs.append(singles)
    noise = sorted(((rel(runs[0][r][n], runs[1][r][n]), n) for r in range(2) for n in d0), reverse=True)[:5]
    print("run-to-run noise of single-process grads:", noise)
    errs = sorted(((rel(d0[n], 0.5 * (runs[0][0][n] + runs[0][1][n])), n) for n in d0), reverse=True)
    print("ddp vs mean of singles, worst:", errs[:12])
    print("count > 2e-3:", sum(e > 2e-3 for e, _ in errs), "of", len(errs))
