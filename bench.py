"""Benchmark of the hot path: images/s of one full training step (forward + Hungarian
criterion + backward + gradient all-reduce + clipped AdamW) of the R50 Mask2Former
part-proposal model on synthetic 1024x1024 batches, bs=2 per GPU (BASELINE.json
configs[1]), on N GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement).  Extra objects:
  roofline      achieved algorithmic GB/s of the dominant hand-written kernel (MSDA backward), measured with
                HIP events around every launch inside the timed steps (same stream as the launches)
  cpu_baseline  the CPU oracle (oracle/step_ref.py, plain PyTorch fp32) timed on this host's cores on a bounded
                sample of the same workload (N=1 only)
  parity        max deviation of the 30 weighted loss terms of ONE benchmarked-precision step (bf16 autocast, the weights
                the timed steps left behind, one 1024 x 1024 image, replayed random points) from the CPU oracle on identical
                inputs, N=1 only; the oracle runs in the cpu_baseline child
  whole_step    algorithmic GFLOP/image x images/s against the composite matrix-core floor of the step
  categories    GPU time per step by kernel family (own HIP / library GEMM / MIOpen / ATen / copies), measured live with
                torch.profiler's device activity records over extra steps after the timed region
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# No shipped MIOpen find-db any more (round 5: the stem was the last library convolution of this workload).  Paths that still reach MIOpen
# (PD_OWN_STEM=0, PD_R50_FUSED=0 comparisons) keep their search results in a private per-rank directory.
if "MIOPEN_USER_DB_PATH" not in os.environ:
    import tempfile
    os.environ["MIOPEN_USER_DB_PATH"] = os.path.join(tempfile.gettempdir(), "pd_miopen_db_rank" + os.environ.get("LOCAL_RANK", "0"))
    os.makedirs(os.environ["MIOPEN_USER_DB_PATH"], exist_ok=True)
if any(x == "--graph" and sys.argv[i + 1:i + 2] not in ([], ["0"]) or (x.startswith("--graph=") and x != "--graph=0")
       for i, x in enumerate(sys.argv)):
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"          # must precede the first HIP call; see TrainStep.capture()
    os.environ["PD_CMDBUF"] = "0"                               # the whole-step graph and the command buffers are alternatives

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

METRIC = "images/sec training step, R50 Mask2Former 1024² bs=2/GPU, 1/2/4/8 MI355X"
MFMA_FP32_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (6.3 TB/s achievable)


def msda_alg_bytes(batch, size, heads=8, head_dim=32, levels=3, points=4, esz=4):
    """forward = value + loc + attn read, out written (137.6 MB at config 2); backward = the forward's four
    tensors + grad_out read, grad_value + grad_loc + grad_attn written (275 MB; the survey's figure counts `out` among the
    tensors the backward touches although the kernel never reads it - kept as the contract's number, so `frac` errs low)."""
    s = sum((size // st) ** 2 for st in (32, 16, 8))
    v = batch * s * heads * head_dim * esz
    lo = batch * s * heads * levels * points * 2 * esz
    at = batch * s * heads * levels * points * esz
    fwd = v + lo + at + v
    return fwd, fwd + v + (v + lo + at)


def cpu_baseline_subprocess(opts, size, timeout=300.0, parity_file=None):
    """run cpu_baseline() in a child process under a hard wall-clock limit (a mis-threaded CPU run must never hold the
    benchmark hostage: on a 256-core host, torch with 256 intra-op threads took 775 s for the 256x256 probe)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--size", str(size)]
    if parity_file:
        cmd += ["--parity-file", parity_file]
    cmd += list(opts)
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout, text=True).stdout
        for line in out.splitlines()[::-1]:
            if line.startswith("{"):
                return json.loads(line)
        return {"error": "no output from the CPU baseline child"}
    except subprocess.TimeoutExpired:
        return {"error": f"CPU baseline exceeded {timeout:.0f} s wall clock and was stopped", "kind": "port"}


def oracle_parity(parity_file):
    """the oracle's 30 weighted losses for the step the GPU just ran (weights, image, masks, random-point seed and the
    product's Hungarian assignments from `parity_file`): losses are evaluated with the product's assignment, the oracle's
    own optimum is used to report how far (in the oracle's fp32 cost) that assignment is from optimal."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import common as C
    from oracle import step_ref as R
    blob = torch.load(parity_file, weights_only=True)
    rows, cols, n = blob["rows"].long(), blob["cols"].long(), int(blob["n_targets"])
    H = rows.shape[0]
    override = [[(rows[H - 1 if h == 0 else h - 1, :n], cols[H - 1 if h == 0 else h - 1, :n])] for h in range(H)]
    costs = []
    with torch.no_grad():
        losses, oidx = R.proposal_model_losses(blob["sd"], [{"image": blob["image"], "instances": {"gt_masks": blob["masks"]}}],
                                               C.ReplayRand(int(blob["seed"])), return_indices=True, indices_override=override,
                                               costs=costs)
    differ, gap = 0, 0.0
    for h in range(H):
        cm = costs[h][0].double()
        (pr, pc), (orow, ocol) = override[h][0], oidx[h][0]
        best, got = cm[orow, ocol].sum().item(), cm[pr, pc].sum().item()
        differ += set(zip(pr.tolist(), pc.tolist())) != set(zip(orow.tolist(), ocol.tolist()))
        gap = max(gap, (got - best) / max(abs(best), 1e-12))
    return {"losses": {k: float(v) for k, v in losses.items()}, "assignments_differing": differ, "assignment_cost_gap": gap}


def cpu_baseline(cfg_opts, size, seconds_budget=90.0, threads=None, parity_file=None):
    """time the oracle's training step (fwd + criterion + bwd + clipped AdamW) on the host cores: the benchmarked batch of 2 images when
    that fits the time budget, else one image."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import common as C
    from oracle import step_ref as R
    import partdistillation_amd.modeling  # noqa: F401  (registers the classes)
    import partdistillation_amd.proposal_model  # noqa: F401
    from partdistillation_amd.compat import build_model
    from partdistillation_amd.config import setup_cfg
    from partdistillation_amd.engine.synthetic import make_batch
    cores = threads or min(os.cpu_count() or 1, 32)          # torch's intra-op pool stops scaling (and thrashes) far below 256
    torch.set_num_threads(cores)
    cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd", "configs", "proposal_learning", "r50_mask2former.yaml"),
                    ["MODEL.DEVICE", "cpu"] + cfg_opts)
    torch.manual_seed(0)
    model = build_model(cfg)                      # only to obtain reference-initialised weights (never run on CPU)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    params = [k for k, v in model.named_parameters()]
    for k in params:
        sd[k].requires_grad_(True)
    del model

    def one_step(s, nb=1):
        batch = make_batch(nb, s, seed=1234, device="cpu")
        obatch = [{"image": b["image"], "instances": {"gt_masks": b["instances"].gt_masks.tensor}} for b in batch]
        g = torch.Generator().manual_seed(0)
        rand = lambda shape: torch.rand(shape, generator=g)
        t0 = time.perf_counter()
        losses = R.proposal_model_losses(sd, obatch, rand)
        sum(losses.values()).backward()
        ps = [sd[k] for k in params]
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in ps]
        state = [(torch.zeros_like(p), torch.zeros_like(p)) for p in ps]
        with torch.no_grad():
            R.clipped_adamw_step([p.data for p in ps], grads, state, lrs=[1e-4] * len(ps), wds=[0.05] * len(ps), step=1)
        dt = time.perf_counter() - t0
        for p in ps:
            p.grad = None
        return dt

    probe = max(128, size // 4)
    one_step(probe)                              # warms the allocator / thread pool
    t_probe = one_step(probe)
    est = t_probe * (size / probe) ** 2
    parity = None
    if parity_file:
        try:
            parity = oracle_parity(parity_file)
        except Exception as e:                   # noqa: BLE001 - the baseline timing must survive a parity failure
            parity = {"error": repr(e)}
    if 2.2 * est <= seconds_budget:              # the GPU line's own batch: 2 images per step
        dt = one_step(size, 2)
        sample = f"2 images {size}x{size} (the benchmarked batch), one full step (fwd+criterion+bwd+clipped AdamW), fp32"
        return {"value": 2.0 / dt, "unit": "images/s", "cores": cores, "kind": "port", "sample": sample, "seconds": dt, "_parity": parity}
    if est <= seconds_budget:
        dt, sample = one_step(size), (f"1 image {size}x{size}, one full step (fwd+criterion+bwd+clipped AdamW), fp32 - deviation from SURVEY 8(d): "
                                      "the GPU line steps bs=2 per GPU, the CPU sample is ONE image (value is images/s, so the ratio is per image)")
        return {"value": 1.0 / dt, "unit": "images/s", "cores": cores, "kind": "port", "sample": sample,
                "seconds": dt, "_parity": parity}
    sample = (f"1 image {probe}x{probe} (1/{(size // probe) ** 2} of the pixels of the {size}x{size} workload; the full-size "
              f"step was estimated at {est:.0f} s > budget), one full step, fp32; value scaled by the pixel ratio")
    return {"value": 1.0 / est, "unit": "images/s", "cores": cores, "kind": "port", "sample": sample, "seconds": t_probe,
            "_parity": parity}


# kernel families for the live per-category table (torch.profiler device records).  First match wins.
_CATEGORIES = [
    ("own_msda", r"^msda_"),
    ("own_fp32_wgrad_mfma", r"^(gemm_wgrad_f32|gemm_wgrad_f16x2|wgrad_tr_reduce|wgrad_h2w_reduce)"),
    # every forward / input-gradient kernel of the fp32 pixel decoder (tiled, row-stream, producer / consumer) + the row-maxima passes that feed them
    ("own_fp32x3_gemm_conv", r"^(gemm_tn_f32|gemm_tn_f16x2|gemm_kpc_f16x2|gemm_rows_f16x2|row_amax_f32|add_rows_amax|cast_bf16_f32_amax|gemm_wgrad_f32x3|conv3x3_)"),
    ("own_igemm_bf16_conv_linear", r"^(igemm_bf16|igemm3x3_bf16|filter_transpose_grouped|wgrad_bf16|stem_|maxpool3s2)"),
    ("own_conv_bf16_filter_grads", r"^conv_(wgrad|igemm)"),
    ("own_attention_mfma", r"^(attn_|wattn_)"),
    ("own_decoder_fused", r"^(dec_fwd_|dec_bwd_|dec_pack_|decoder_head)"),
    ("own_skinny_bf16_gemm", r"^sgemm_"),
    ("own_criterion", r"^(pair_logits|loss_vectors|mask_point_losses|uncertain_points|matcher_|match_point_logits|point_sample|skinny_linear|lsa_)"),
    ("own_rowwise_norm_optim_misc", r"^(add_ln_|colsum_|mem_prep|msda_prep|gn_coeffs|affine_act|nc_|multi_gather|upsample|layernorm_rows|ln_rows|"
                                    r"sumsq|adamw|swin_ln|kmeans|scores_|mask_assign|resample_|resize_|normalize_|rle_|amax_|quantize_|bn_|sum3_|transpose_batched|"
                                    r"copy_d2d|copy_segments|relu_bwd|mx8_)"),
    ("library_gemm_fp32", r"^Cijk_.*_S_B"),
    ("library_gemm_bf16", r"^(Cijk_|.*kernel_batched_gemm|.*kernel_gemm)"),
    ("miopen_conv", r"(igemm_|grouped_conv|naive_conv|SubTensorOp|batched_transpose|gridwise|MIOpen|miopen|Im2Col|Col2Im)"),
    ("memset_copy", r"(fillBuffer|copyBuffer|Memcpy|Memset|memcpy|memset)"),
    ("torch_aten", r"(^at::|at::native|^at_cuda_detail|softmax_warp|c10::)"),
]


def profile_categories(step, batches, nsteps=2):
    """GPU-busy time per step by kernel family from torch.profiler's device activity records of `nsteps` extra eager
    steps (after the timed region).  -> (list of {category, ms_per_step, launches_per_step}, busy_ms_per_step, launches)"""
    import collections
    import re
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for i in range(nsteps):
            step(batches[i % len(batches)])
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0.0, 0])
    for e in prof.events():
        if getattr(e, "device_type", None) is None or "CUDA" not in str(e.device_type):
            continue
        us = float(getattr(e, "device_time_total", 0.0) or getattr(e, "cuda_time_total", 0.0) or 0.0)
        name = re.sub(r"^void ", "", e.name)
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        cat = next((c for c, pat in _CATEGORIES if re.search(pat, name)), "own_other")
        agg[cat][0] += us
        agg[cat][1] += 1
    busy = sum(v[0] for v in agg.values())
    out = [{"category": c, "ms_per_step": v[0] / 1e3 / nsteps, "launches_per_step": v[1] / nsteps, "share": v[0] / max(busy, 1e-9)}
           for c, v in sorted(agg.items(), key=lambda kv: -kv[1][0])]
    return out, busy / 1e3 / nsteps, sum(v[1] for v in agg.values()) / nsteps


def category_rooflines(cats, batch, size, freeze):
    """attach algorithmic work + roofline fraction to the families whose work is a closed form of the workload
    (encoder tokens M = batch * sum_l (size/stride_l)^2, 6 encoder layers)."""
    M = batch * sum((size // st) ** 2 for st in (32, 16, 8))
    hw4 = batch * (size // 4) ** 2
    enc_w = 0 if "encoder" in freeze else 1                 # frozen encoder: no weight gradients (and the backbone's none either)
    from partdistillation_amd.functions import gemm as gemm_fn
    np_w, np_f = products_per_fp32_product()
    wg = (np_w, 2500.0, f"16-bit matrix 2.5 PF, {np_w:.0f} 16-bit products per fp32 product (gemm_wgrad_f32x3_tr)") if gemm_fn.WGRAD_X3 else \
         (1.0, MFMA_FP32_PEAK_TFLOPS, "fp32 matrix 157.3 TF")
    work = {
        # fp32 weight gradients of the 6 encoder layers: value/out 256x256, offsets+weights 288x256, FFN 2 x 1024x256
        # + (same kernel family since round 2) the filter gradients of the fp32 FPN convolutions: 3 x 3 and two 1 x 1 at stride 4, the
        # three input projections (2048 / 1024 / 512 -> 256 at strides 32 / 16 / 8); the matched mask-logit gradient of the criterion
        "own_fp32_wgrad_mfma": (2.0 * (6 * M * (2 * 256 * 256 + 288 * 256 + 2 * 1024 * 256) * enc_w
                                         + (hw4 * (9 + 2) * 256 * 256 + sum(batch * (size // st) ** 2 * c * 256 for st, c in ((32, 2048), (16, 1024), (8, 512)))) * enc_w
                                         + hw4 * 40 * 256), wg[1], wg[2], wg[0]),
        # encoder FFN + 256-wide projections forward + input gradient, 3x3 FPN conv forward + input gradient: np_f 16-bit MFMA products
        # per fp32 product (3 in the fp16 two-plane form, 6 in the bf16 three-plane form)
        "own_fp32x3_gemm_conv": ((6 * 2 * 2.0 * M * (2 * 256 * 1024 + 2 * 256 * 256 + 288 * 256) + 2 * 2.0 * hw4 * 9 * 256 * 256), 2500.0,
                                 f"16-bit matrix 2.5 PF, {np_f:.0f} 16-bit products per fp32 product", np_f),
    }
    # R50 bottleneck body forward + input gradient on pd_igemm_bf16 (52 convolutions per direction: 166 GFLOP per image and direction at
    # 1024 x 1024, tools/bench_igemm.py) + the decoder's key / value input gradients over the memory tokens (9 layers x 2 x [B HW_l, 256] x [256, 256])
    r50_dirs = 1 if "backbone" in freeze else 2
    work["own_igemm_bf16_conv_linear"] = (166.0e9 * batch * (size / 1024.0) ** 2 * r50_dirs
                                          + 3 * 2 * 2.0 * 256 * 256 * sum(batch * (size // st) ** 2 for st in (32, 16, 8)), 2500.0,
                                          "bf16 matrix 2.5 PF (v_mfma_f32_32x32x16_bf16)", 1.0)
    for c in cats:
        w = work.get(c["category"])
        if w and w[0] > 0 and c["ms_per_step"] > 0:
            tf = w[0] / (c["ms_per_step"] * 1e-3) / 1e12
            c.update({"alg_gflop_per_step": w[0] / 1e9, "achieved_TFLOPs": tf, "peak_TFLOPs": w[1], "frac": tf / w[1],
                      "issued_products_per_fp32_product": w[3], "frac_on_issued_products": w[3] * tf / w[1],
                      "peak_note": w[2] + "; alg_gflop = 2 M N K per product (SURVEY 8d)"})
    return cats


def products_per_fp32_product():
    """-> (weight-gradient kernels, forward / input-gradient kernels): 16-bit MFMA products the fp32 pixel-decoder kernels issue per
    fp32 product — 3 in the fp16 two-plane form (csrc/gemm_f16x2.hip), 6 in the bf16 three-plane form (csrc/gemm_x3.hip)"""
    from partdistillation_amd.functions import encoder_core as ec
    return (3.0 if (ec.H2 and ec.H2_WGRAD) else 6.0), (3.0 if ec.H2 else 6.0)


def step_pmc_traffic(label, batch, size):
    """(HBM bytes per launch, source) of the kernel a timing label names, from the newest committed whole-step counter summary
    (profiles/rNN_step_pmc_traffic_by_kernel.csv, written by tools/pmc_step.sh at config 2: bs 2, 1024 x 1024); (None, None) elsewhere"""
    import csv
    import glob
    if (batch, size) != (2, 1024):
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_step_pmc_traffic_by_kernel.csv")))
    if not files:
        return None, None
    import re
    conv = "conv 3x3" in label
    base = re.split(r" / | \(\+", label)[0].replace(", conv 3x3", "").strip()
    if base.endswith(">"):
        base = base[:-1]
    tot, n = 0.0, 0.0
    for r in csv.DictReader(open(files[-1])):
        k = r["kernel"]
        if not k.startswith(base) or (k[len(base):len(base) + 1] not in (",", ">", "<", "")):
            continue
        if k.startswith("gemm_tn_f16x2<") and (", true" in k) != conv:
            continue
        tot += float(r["hbm_MB_per_launch"]) * float(r["launches_per_step"])
        n += float(r["launches_per_step"])
    if n == 0:
        return None, None
    return tot / n * 1024 * 1024, os.path.relpath(files[-1], ROOT) + " (rocprofv3 --pmc over bench.py, same launch mix)"


def roofline_of(dom, kernels):
    """roofline object of the kernel with the largest share of the step among the timed hand-written kernels:
    achieved = algorithmic bytes (or flops) per launch / average launch duration (HIP events on the launch stream)."""
    if dom is None:
        return None
    if "bound" in dom:                                             # a GEMM entry with both floors: report the one that binds
        mf = dom["bound"] == "mfma"
        return {"bound": dom["bound"], "kernel": dom["kernel"], "achieved": dom["achieved_TFLOPs"] if mf else dom["achieved_GBs"],
                "peak": dom["peak_TFLOPs"] if mf else HBM_PEAK_GBS, "unit": "TFLOP/s" if mf else "GB/s",
                "frac": dom["mfma_frac"] if mf else dom["hbm_frac"], "traffic": dom.get("traffic"),
                "traffic_source": dom.get("traffic_source"),
                "other_floor_frac": dom["hbm_frac"] if mf else dom["mfma_frac"],
                "alg_flops_per_launch": dom["alg_flops"], "issued_flops_per_launch": dom["issued_flops"],
                "mfma_frac_on_issued_products": dom["mfma_frac_issued"],
                "alg_bytes_per_launch": dom["alg_bytes"], "avg_launch_ms": dom["avg_ms"],
                "launches_timed": dom["launches"], "peak_source": dom["peak_source"], "other_kernels": kernels}
    if "achieved_TFLOPs" in dom:
        peak = dom.get("peak_TFLOPs", MFMA_FP32_PEAK_TFLOPS)
        return {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["achieved_TFLOPs"], "peak": peak,
                "unit": "TFLOP/s", "frac": dom["achieved_TFLOPs"] / peak, "traffic": None,
                "alg_flops_per_launch": dom["alg_flops"], "avg_launch_ms": dom["avg_ms"], "launches_timed": dom["launches"],
                "peak_source": dom.get("peak_source", "MI355X_MICROARCH.md: fp32 matrix (v_mfma_f32_32x32x2_f32) 157.3 TFLOP/s"),
                "other_kernels": kernels}
    return {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": dom["achieved_GBs"] / HBM_PEAK_GBS, "traffic": dom.get("traffic"),
            "traffic_source": dom.get("traffic_source"),
            "alg_bytes_per_launch": dom["alg_bytes"], "avg_launch_ms": dom["avg_ms"], "launches_timed": dom["launches"],
            "other_kernels": kernels}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=2, help="images per GPU")
    ap.add_argument("--freeze", default="", help='comma list for MODEL.MASK_FORMER.FREEZE_KEYS, e.g. "backbone,encoder"')
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) run only the CPU oracle timing and print it")
    ap.add_argument("--parity-file", default="", help="(internal) state the GPU parity step left for the oracle child")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle comparison of one benchmarked-precision step")
    ap.add_argument("--no-categories", action="store_true", help="skip the torch.profiler per-family GPU time table")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--miopen-find", type=int, default=1, help="1: let MIOpen search conv algorithms during warm-up")
    ap.add_argument("--skip-kernel-timing", action="store_true", help="skip the eager per-launch timing steps (profiling runs)")
    ap.add_argument("--graph", type=int, default=0,
                    help="1: capture the whole step in a hipGraph (single GPU) and replay it, with ROCm's graph packet capture "
                         "switched off (packet-captured graphs go stale after eager launches on ROCm 7.2).  Off by default: "
                         "without packet capture a replay costs MORE host CPU than eager issue (DESIGN.md §5 'hipGraph')")
    ap.add_argument("opts", nargs="*", help="extra KEY VALUE config overrides")
    a = ap.parse_args()

    if a.cpu_baseline_only:
        print(json.dumps(cpu_baseline(list(a.opts), a.size, threads=a.cpu_threads or None, parity_file=a.parity_file or None)), flush=True)
        return
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with python -m torch.distributed.run --nproc-per-node N ...")
    if os.environ.get("PD_TEST_SHARE_GPU"):            # test hook: N ranks on ONE GPU over gloo (tests/test_ddp_gpu.py)
        local = 0
    torch.cuda.set_device(local)
    force = world == 1 and os.environ.get("PD_DDP_FORCE", "0") == "1"   # one rank, every collective of the step through RCCL anyway (development)
    if force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("PD_TEST_SHARE_GPU"):
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    torch.backends.cudnn.benchmark = bool(a.miopen_find)

    from partdistillation_amd import lib
    lib.load()                                                     # fail loudly if the HIP extension is missing
    from partdistillation_amd.config import setup_cfg
    from partdistillation_amd.engine.synthetic import make_batch
    from partdistillation_amd.engine.trainer import TrainStep
    from partdistillation_amd.modeling.pixel_decoder.ops.functions import ms_deform_attn_func as msda_fn

    freeze = [k for k in a.freeze.split(",") if k]
    cfg_opts = ["INPUT.IMAGE_SIZE", str(a.size)] + list(a.opts)
    if freeze:
        cfg_opts += ["MODEL.MASK_FORMER.FREEZE_KEYS", str(freeze)]
    cfg = setup_cfg(os.path.join(ROOT, "partdistillation_amd", "configs", "proposal_learning", "r50_mask2former.yaml"), cfg_opts)
    torch.manual_seed(0)                                           # identical init on every rank (+ broadcast in TrainStep)
    step = TrainStep(cfg)
    batches = [make_batch(a.batch, a.size, seed=1234 + rank + 1000 * i, device="cuda") for i in range(4)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = bool(a.graph) and world == 1
    for i in range(a.warmup):
        step(batches[i % len(batches)])
    dbg = bool(os.environ.get("PD_DEBUG_GRAPH"))
    if use_graph:
        step.capture(batches[0])
        if dbg:
            torch.cuda.synchronize(); print("captured", file=sys.stderr, flush=True)
        step(batches[1])                                           # one replay before the clock starts
        if dbg:
            torch.cuda.synchronize(); print("first replay ok", file=sys.stderr, flush=True)
    barrier()
    # one event per step on the launch stream: the distribution (median / min / max) of the per-step times of the timed region,
    # next to the mean the contract asks for (no synchronisation inside the region: the events are read after the closing barrier)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t0 = time.perf_counter()
    c0 = time.process_time()
    marks[0].record()
    for i in range(a.steps):
        losses = step(batches[i % len(batches)])
        marks[i + 1].record()
        if dbg:
            torch.cuda.synchronize(); print("step", i, "ok", file=sys.stderr, flush=True)
    issue = time.perf_counter() - t0                               # host WALL time to issue the steps: includes the time the host is
    host_cpu = time.process_time() - c0                            # blocked on a full launch queue; CPU time of the process = its real work
    barrier()
    elapsed = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    # per-launch timing of the hand-written MSDA kernels with HIP events on the launch stream.  Events cannot be
    # recorded between the nodes of a replayed graph, so these launches are timed in extra EAGER steps of the same
    # workload right after the timed region (when --graph 0 they are timed inside the timed region itself).
    fwd_ms, bwd_ms, wgrad, x3fwd = [], [], [], []
    if not a.skip_kernel_timing:
        step.release_graph()
        from partdistillation_amd.functions import gemm as gemm_fn
        # one untimed step in this mode first: the eager (not replayed) encoder path launches kernel variants the recorded region never does, and
        # the first launch of a kernel in a process carries its code-object load (one 5 ms launch in 18 moved an average from 110 to 406 us on a
        # fresh box)
        msda_fn.enable_timing(True)
        gemm_fn.enable_timing(True)
        step(batches[0])
        torch.cuda.synchronize()
        msda_fn.enable_timing(True)                                # (clears the lists)
        gemm_fn.enable_timing(True)
        for i in range(3):
            step(batches[i % len(batches)])
        torch.cuda.synchronize()
        fwd_ms, bwd_ms = msda_fn.timing_ms()
        wgrad = gemm_fn.timing()
        x3fwd = gemm_fn.timing("fwd")
        msda_fn.enable_timing(False)
        gemm_fn.enable_timing(False)
    cats = None
    if rank == 0 and world == 1 and not a.no_categories:
        try:
            cats = profile_categories(step, batches)
        except Exception as e:                                     # noqa: BLE001 - a profiler problem must not cost the bench line
            cats = ({"error": repr(e)}, None, None)
    # one step at the benchmarked precision whose 30 losses the CPU oracle recomputes on identical inputs (N=1 only)
    parity_file, parity_gpu = None, None
    if rank == 0 and world == 1 and not a.no_parity and not a.no_cpu_baseline and not a.opts and not freeze:
        import tempfile
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import common as C
        one = make_batch(1, a.size, seed=4321, device="cuda")
        sd = {k: v.detach().float().cpu().clone() for k, v in step.state_dict()["model"].items()}
        step.model.criterion.rand = C.ReplayRand(2718)
        real_step, step.optimizer.step = step.optimizer.step, (lambda: None)       # forward + backward only: weights stay as dumped
        pl = step(one)
        step.optimizer.step, step.model.criterion.rand = real_step, None
        rows, cols = pl.indices
        parity_gpu = {k: float(v) for k, v in pl.items()}
        fd, parity_file = tempfile.mkstemp(suffix=".pt", prefix="pd_parity_")
        os.close(fd)
        torch.save({"sd": sd, "image": one[0]["image"].cpu(), "masks": one[0]["instances"].gt_masks.tensor.cpu(), "seed": torch.tensor(2718),
                    "rows": rows.cpu(), "cols": cols.cpu(), "n_targets": torch.tensor(one[0]["instances"].gt_masks.tensor.shape[0])}, parity_file)
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    total_loss = float(sum(v.detach() for v in losses.values()))
    if total_loss != total_loss or abs(total_loss) == float("inf"):
        raise SystemExit("bench.py: the loss is not finite after the timed steps - the measurement is invalid")

    if rank == 0:
        images = a.batch * world * a.steps
        fb, bb = msda_alg_bytes(a.batch, a.size)
        avg = lambda xs: sum(xs) / max(len(xs), 1)
        kernels = []
        if fwd_ms:
            kernels.append({"kernel": "msda_fwd_d32" if os.environ.get("PD_MSDA_FWD_Q4", "1") == "0" else "msda_fwd_q4", "launches": len(fwd_ms), "avg_ms": avg(fwd_ms),
                            "alg_bytes": fb, "achieved_GBs": fb / avg(fwd_ms) / 1e6})
        if bwd_ms:
            import ctypes
            gate = (ctypes.c_uint * 3)()
            lib.load().pd_msda_backward_last_gate(gate)      # host memory only: what the backward launches measured / which variant ran
            kernels.append({"kernel": "msda_bwd_owner4_d32", "launches": len(bwd_ms), "avg_ms": avg(bwd_ms),
                            "alg_bytes": bb, "achieved_GBs": bb / avg(bwd_ms) / 1e6,
                            "variant_of_last_launch": {2: "halo 9, 16 channels per workgroup", 3: "halo 5, 32 channels per workgroup"}.get(int(gate[2]), "ungated"),
                            "halo5_window_miss_fraction": (gate[0] / gate[1]) if gate[1] else None,
                            "note": "offsets after the timed steps from reference initialisation; tools/msda_sweep.sh covers N(0, sigma) and a trained-model stand-in"})
        def gemm_entry(name, sel, nprod, note):
            """roofline entry of one timed GEMM kernel + shape: both floors — nprod 16-bit products per fp32 product at the 2.5 PF matrix peak,
            and operands + result once each at 8 TB/s — and the one that binds (the larger floor)"""
            t_ms = sum(t for t, _ in sel)
            fl, by = sum(f[0] for _, f in sel), sum(f[1] for _, f in sel)
            n = len(sel)
            # algorithmic FLOPs = 2 M N K.  The kernels ISSUE nprod 16-bit products per fp32 product; that is reported next
            # to it (issued_*), never as the algorithmic figure.
            mfma_floor_ms, hbm_floor_ms = fl / 2500.0e12 * 1e3, by / (HBM_PEAK_GBS * 1e9) * 1e3
            e = {"kernel": name, "launches": n, "avg_ms": t_ms / n, "alg_flops": fl / n, "issued_flops": nprod * fl / n, "alg_bytes": by / n,
                 "achieved_TFLOPs": fl / t_ms / 1e9, "issued_TFLOPs": nprod * fl / t_ms / 1e9, "peak_TFLOPs": 2500.0, "achieved_GBs": by / t_ms / 1e6,
                 "mfma_frac": mfma_floor_ms / t_ms, "mfma_frac_issued": nprod * mfma_floor_ms / t_ms, "hbm_frac": hbm_floor_ms / t_ms,
                 "bound": "mfma" if mfma_floor_ms >= hbm_floor_ms else "hbm",
                 "peak_source": "MI355X_MICROARCH.md: dense 16-bit matrix (v_mfma_f32_32x32x16_f16 / _bf16) 2.5 PFLOP/s and HBM3E 8 TB/s; "
                                f"alg_flops = 2 M N K (the kernel issues {nprod:.0f} 16-bit products per fp32 product: issued_flops), "
                                "alg_bytes = operands + result once each; " + note}
            return e
        if wgrad:                               # fp32 weight-gradient GEMMs: the encoder's grouped launch and the single ones (convolutions, criterion)
            for kind in sorted({f[2] for _, f in wgrad}):
                sel = [(t, f) for t, f in wgrad if f[2] == kind]
                nprod = 3.0 if kind.startswith("h2") else 6.0 if kind.startswith("x3") else 1.0
                name = {"h2 grouped": "gemm_wgrad_f16x2_wide_grouped / gemm_wgrad_f32x3_tr_grouped<f16x2> (+ their reduces)", "h2": "gemm_wgrad_f32x3_tr<f16x2> (+ wgrad_tr_reduce)",
                        "x3 grouped": "gemm_wgrad_f32x3_tr_grouped (+ wgrad_tr_reduce_grouped)", "x3": "gemm_wgrad_f32x3_tr (+ wgrad_tr_reduce)"}.get(kind, kind)
                kernels.append(gemm_entry(name, sel, nprod, "avg_ms includes the partial-tile reduce launch"))
        for lab in sorted({f[1] for _, f in x3fwd}):   # fp32 forward / input-gradient products on the 16-bit matrix cores, per kernel and shape
            sel = [(t, f) for t, f in x3fwd if f[1] == lab]
            kernels.append(gemm_entry(lab, [(t, (f[0], f[2])) for t, f in sel], 3.0 if "f16x2" in lab else 6.0, ""))
        # HBM bytes per launch: the committed counter passes over bench.py ITSELF (tools/pmc_step.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate passes, FETCH x 2 per MI355X_MICROARCH.md, per-kernel averages over the timed steps) - the same launch mix as the
        # timed kernels above, so traffic / alg_bytes_per_launch is a traffic ratio
        for kk in kernels:
            kk["traffic"], kk["traffic_source"] = step_pmc_traffic(kk["kernel"], a.batch, a.size)
        dom = max(kernels, key=lambda k: k["avg_ms"] * k["launches"]) if kernels else None
        out = {
            "metric": METRIC, "value": images / elapsed, "unit": "images/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "ms_per_step_stats": {"median": per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2]),
                                  "min": per_step[0], "max": per_step[-1], "p10": per_step[int(0.1 * (len(per_step) - 1))],
                                  "p90": per_step[int(round(0.9 * (len(per_step) - 1)))],
                                  "source": "HIP events between the steps of the timed region on the launch stream (rank 0); the headline value is "
                                            "the contract's mean over the region; SURVEY 8(d) headline setting: --steps 50 --warmup 10"},
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"R50 Mask2Former part-proposal training step (ProposalModel), {a.size}x{a.size} synthetic, "
                                   f"bs={a.batch}/GPU, Q=100, 10 prediction heads, bf16 autocast; matcher fp32; pixel decoder fp32 storage with its GEMMs / convolutions "
                                   "on the fp16 matrix cores as 2 x fp16 planes per operand (22 significand bits, rows scaled by powers of two: normwise-fp32, "
                                   "error <= max(2^-22 |x|, 2^-39 row max) per element - not elementwise IEEE fp32)",
                       "global_batch": a.batch * world, "parallelism": f"dp{world}",
                       "finetune": "frozen:" + ",".join(freeze) if freeze else "full", "hipgraph": use_graph,
                       "final_total_loss": total_loss, "host_issue_ms_per_step": issue / a.steps * 1e3,
                       "host_cpu_ms_per_step": host_cpu / a.steps * 1e3,
                       "host_note": "host_issue = wall time of the issuing loop (includes waiting on a full launch queue when the GPU is the "
                                    "limiter: it then follows the GPU step, not the host's work); host_cpu = CPU time of the process over the same loop (all threads)"},
            "roofline": roofline_of(dom, kernels),
        }
        # whole-step matrix-core fraction (1 565 GFLOP / image full fine-tune, ~1 030 frozen; of the full
        # step ~870 GF / image are fp32 pixel-decoder work pinned by the reference's precision contract, the rest bf16)
        gf_img = 1030.0 if freeze else 1565.0
        fp32_gf = (870.0 if not freeze else 870.0 * 2 / 3) * a.batch
        bf16_gf = gf_img * a.batch - fp32_gf
        _, np_f = products_per_fp32_product()
        floor_native = fp32_gf / MFMA_FP32_PEAK_TFLOPS + bf16_gf / 2500.0           # fp32 share on the native fp32 matrix instruction
        floor_issued = np_f * fp32_gf / 2500.0 + bf16_gf / 2500.0                   # fp32 share as the kernels issue it: np_f 16-bit products each
        floor_alg = gf_img * a.batch / 2500.0                                        # every algorithmic FLOP at the 16-bit matrix peak
        out["whole_step"] = {"alg_gflop_per_image": gf_img, "achieved_TFLOPs": gf_img * out["value"] / world / 1e3,
                             "frac_of_bf16_peak_on_algorithmic_flops": floor_alg / out["ms_per_step"],
                             "floor_ms_as_issued": floor_issued, "mfma_fraction_as_issued": floor_issued / out["ms_per_step"],
                             "floor_ms_native_fp32": floor_native, "mfma_fraction_native_fp32": floor_native / out["ms_per_step"],
                             "note": "SURVEY 8(d) algorithmic FLOPs (2 M N K per product).  frac_of_bf16_peak_on_algorithmic_flops = all of them at 2.5 PF / "
                                     f"measured step; as_issued prices the fp32 pixel-decoder share at {np_f:.0f} 16-bit products per fp32 product (what the "
                                     "two-plane kernels execute) at 2.5 PF; native_fp32 prices that share at the 157.3 TF fp32 matrix instruction - a floor the "
                                     "two-plane kernels already beat, kept for continuity with rounds 1-3 (mfma_fraction there)"}
        if cats is not None:
            table, busy, launches = cats
            if busy is None:
                out["categories"] = table
            else:
                out["categories"] = {"gpu_busy_ms_per_step": busy, "launches_per_step": launches, "source": "torch.profiler device records, 2 eager steps after the timed region",
                                     "families": category_rooflines(table, a.batch, a.size, freeze)}
        if world == 1 and not a.no_cpu_baseline:
            cb = cpu_baseline_subprocess(list(a.opts), a.size, parity_file=parity_file)
            po = cb.pop("_parity", None) if isinstance(cb, dict) else None
            out["cpu_baseline"] = cb
            if parity_gpu is not None and po and "losses" in po:
                rel = {k: abs(parity_gpu[k] - v) / max(abs(v), 1e-12) for k, v in po["losses"].items()}
                ab = {k: abs(parity_gpu[k] - v) for k, v in po["losses"].items()}
                worst = max(rel, key=rel.get)
                out["parity"] = {"what": "30 weighted losses of one bf16-autocast step (1 image, weights after the timed steps, replayed points) vs the fp32 CPU oracle on identical inputs",
                                 "n_losses": len(rel), "max_rel_loss_dev": rel[worst], "max_abs_loss_dev": max(ab.values()), "worst_term": worst,
                                 "total_gpu": sum(parity_gpu.values()), "total_cpu": sum(po["losses"].values()),
                                 "assignments_differing_from_oracle_optimum": po["assignments_differing"],
                                 "assignment_cost_gap_rel": po["assignment_cost_gap"], "tolerance_rel": 2e-2, "tolerance_abs": 2e-3,   # |gpu - cpu| <= rel |cpu| + abs per term
                                 "within_tolerance": all(ab[k] <= 2e-2 * abs(po["losses"][k]) + 2e-3 for k in ab)}
            elif parity_gpu is not None:
                out["parity"] = {"error": (po or {}).get("error", "the oracle child returned no losses")}
            if parity_file:
                try:
                    os.remove(parity_file)
                except OSError:
                    pass
        if world == 1 and not a.no_cpu_baseline and not freeze and not a.opts and (a.batch, a.size) == (2, 1024):
            # the shipped scripts' setting (reference sh_files/proposal_learning/train_multi.sh:8: FREEZE_KEYS backbone + encoder) next to the
            # full fine-tune the line reports (SURVEY 8d asks for both): the same bench in a child process, timing only
            import gc
            import subprocess
            try:
                # the child gets the GPU to itself: this process's step (arenas, recorded regions, parameters) is released first
                step = batches = losses = None
                from partdistillation_amd import cmdbuf as _cb
                _cb.drop_all()
                gc.collect()
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--freeze", "backbone,encoder", "--steps", str(a.steps), "--warmup", str(a.warmup),
                                     "--no-cpu-baseline", "--no-categories", "--no-parity", "--skip-kernel-timing"], stdout=subprocess.PIPE,
                                    stderr=subprocess.PIPE, timeout=600, text=True)
                r = pr.stdout
                fz = next((json.loads(l) for l in r.splitlines()[::-1] if l.startswith('{"metric"')), None)
                out["frozen_backbone_encoder"] = ({"value": fz["value"], "unit": fz["unit"], "ms_per_step": fz["ms_per_step"], "steps": fz["steps"],
                                                   "finetune": fz["config"]["finetune"]} if fz else
                                                  {"error": "no line from the child", "returncode": pr.returncode, "stderr_tail": pr.stderr[-400:]})
            except Exception as e:                      # noqa: BLE001 - an extra key must never take the line down
                out["frozen_backbone_encoder"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1 or force:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
