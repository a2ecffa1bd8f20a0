"""GPU parity tests of the HIP multi-scale deformable attention operator, called
through the C-ABI (ctypes -> libpd_hip.so), against the oracle and the goldens.
Re-implements the three checks of the reference's ops/test.py (:38-91)."""
import pytest
import torch

import common as C
from oracle import msda as omsda

pytestmark = pytest.mark.gpu


def _msda():
    import partdistillation_amd.MultiScaleDeformableAttention as MSDA
    from partdistillation_amd.modeling.pixel_decoder.ops.functions import MSDeformAttnFunction
    return MSDA, MSDeformAttnFunction


def _mk(N, M, D, shapes, Lq, P, dt, seed=0, spread=0.45):
    shapes = torch.as_tensor(shapes, dtype=torch.long)
    lvl = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    L = shapes.shape[0]
    value = C.seeded((N, S, M, D), seed + 1, dtype=dt)
    loc = C.seeded((N, Lq, M, L, P, 2), seed + 2, spread, dtype=dt) + 0.5
    attn = C.seeded((N, Lq, M, L, P), seed + 3, dtype=dt).flatten(-2).softmax(-1).view(N, Lq, M, L, P)
    gout = C.seeded((N, Lq, M * D), seed + 4, dtype=dt)
    return value, shapes, lvl, loc, attn, gout


def _cuda(*ts):
    return [t.cuda() for t in ts]


def test_reference_fixture_forward_double_and_float(golden):
    """ops/test.py:38-67: fp64 allclose default tol; fp32 rtol 1e-2 atol 1e-3."""
    MSDA, Fn = _msda()
    g = golden("msda")["test_py"]
    v, sh, lv, lo, at = _cuda(g["value"], g["shapes"], g["lvl"], g["loc"], g["attn"])
    out64 = Fn.apply(v.double(), sh, lv, lo.double(), at.double(), 2).cpu()
    assert torch.allclose(out64, g["out_f64"])
    out32 = Fn.apply(v, sh, lv, lo, at, 2).cpu()
    assert torch.allclose(out32, g["out_f32"], rtol=1e-2, atol=1e-3)
    assert torch.allclose(out32, g["out_f32"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("D", [30, 32, 64, 71, 1025, 2048, 3096])
def test_backward_double_all_channel_counts(D):
    """ops/test.py:69-91 exercises D in {30,32,64,71,1025,2048,3096} with gradcheck;
    here the analytic C oracle is the checker (fp64, atomics reorder only)."""
    MSDA, Fn = _msda()
    value, shapes, lvl, loc, attn, gout = _mk(1, 2, D, [(6, 4), (3, 2)], 2, 2, torch.float64, seed=D)
    v, sh, lv, lo, at, go = _cuda(value, shapes, lvl, loc, attn, gout)
    v.requires_grad_(), lo.requires_grad_(), at.requires_grad_()
    out = Fn.apply(v, sh, lv, lo, at, 2)
    torch.testing.assert_close(out.cpu(), omsda.msda_forward(value, shapes, lvl, loc, attn), rtol=1e-10, atol=1e-12)
    out.backward(go)
    gv, gl, ga = omsda.msda_backward(value, shapes, lvl, loc, attn, gout)
    torch.testing.assert_close(v.grad.cpu(), gv, rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(lo.grad.cpu(), gl, rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(at.grad.cpu(), ga, rtol=1e-9, atol=1e-11)


def test_gradcheck_numerical_small():
    """the reference's own check (ops/test.py:69-84), D=4 and D=32 to keep it quick."""
    MSDA, Fn = _msda()
    for D in (4, 32):
        value, shapes, lvl, loc, attn, _ = _mk(1, 2, D, [(6, 4), (3, 2)], 2, 2, torch.float64, seed=7)
        v, sh, lv, lo, at = _cuda(value * 0.01, shapes, lvl, loc, attn)
        v.requires_grad_(), lo.requires_grad_(), at.requires_grad_()
        assert torch.autograd.gradcheck(Fn.apply, (v, sh, lv, lo, at, 2))


@pytest.mark.parametrize("case", ["m2f", "ragged"])
def test_goldens_forward_backward(golden, case):
    MSDA, Fn = _msda()
    g = golden("msda")[case]
    c = {"m2f": dict(N=1, M=8, D=32, P=4, dt=torch.float32), "ragged": dict(N=2, M=3, D=5, P=3, dt=torch.float64)}[case]
    shapes, lvl, Lq = g["shapes"], g["lvl"], int(g["Lq"])
    S, Lv, dt = int(shapes.prod(1).sum()), shapes.shape[0], c["dt"]
    value = C.seeded((c["N"], S, c["M"], c["D"]), 11, dtype=dt)
    loc = C.seeded((c["N"], Lq, c["M"], Lv, c["P"], 2), 12, 0.45, dtype=dt) + 0.5
    attn = C.seeded((c["N"], Lq, c["M"], Lv, c["P"]), 13, dtype=dt).flatten(-2).softmax(-1).view(c["N"], Lq, c["M"], Lv, c["P"])
    gout = C.seeded((c["N"], Lq, c["M"] * c["D"]), 14, dtype=dt)
    v, sh, lv, lo, at, go = _cuda(value, shapes, lvl, loc, attn, gout)
    v.requires_grad_(), lo.requires_grad_(), at.requires_grad_()
    out = Fn.apply(v, sh, lv, lo, at, 128)
    tol = dict(rtol=1e-4, atol=1e-5) if dt == torch.float32 else dict(rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(out.cpu(), g["out"], **tol)
    out.backward(go)
    torch.testing.assert_close(v.grad.cpu(), g["grad_value"], **tol)
    torch.testing.assert_close(lo.grad.cpu(), g["grad_loc"], **tol)
    torch.testing.assert_close(at.grad.cpu(), g["grad_attn"], **tol)


@pytest.mark.parametrize("N,Lq", [(1, 1), (2, 37), (3, 100), (2, 0)])
def test_fast_path_vs_oracle_ragged_sizes(N, Lq):
    """fp32 / D=32 / L=3 / P=4 fast path incl. partial last block, batch straddling, Lq=0,
    points far outside the maps."""
    MSDA, Fn = _msda()
    value, shapes, lvl, loc, attn, gout = _mk(N, 8, 32, [(5, 3), (9, 11), (16, 20)], Lq, 4, torch.float32, seed=N * 100 + Lq, spread=0.7)
    v, sh, lv, lo, at, go = _cuda(value, shapes, lvl, loc, attn, gout)
    out = MSDA.ms_deform_attn_forward(v, sh, lv, lo, at, 128)
    assert out.shape == (N, Lq, 256)
    torch.testing.assert_close(out.cpu(), omsda.msda_forward(value, shapes, lvl, loc, attn), rtol=1e-4, atol=1e-5)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, sh, lv, lo, at, go, 128)
    ov, ol, oa = omsda.msda_backward(value, shapes, lvl, loc, attn, gout)
    torch.testing.assert_close(gv.cpu(), ov, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gl.cpu(), ol, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(ga.cpu(), oa, rtol=1e-4, atol=1e-4)


def test_nan_and_out_of_range_locations_contribute_nothing():
    MSDA, Fn = _msda()
    value, shapes, lvl, loc, attn, gout = _mk(1, 8, 32, [(4, 4), (8, 8), (16, 16)], 16, 4, torch.float32, seed=5)
    loc[0, 3] = float("nan")
    loc[0, 4] = 7.0
    loc[0, 5] = -3.0
    v, sh, lv, lo, at, go = _cuda(value, shapes, lvl, loc, attn, gout)
    out = MSDA.ms_deform_attn_forward(v, sh, lv, lo, at, 128)
    assert torch.isfinite(out).all()
    assert out[0, 3:6].abs().sum() == 0
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, sh, lv, lo, at, go, 128)
    assert torch.isfinite(gv).all() and torch.isfinite(gl).all() and torch.isfinite(ga).all()
    assert gl[0, 3:6].abs().sum() == 0 and ga[0, 3:6].abs().sum() == 0


def test_error_behaviour_matches_reference():
    """ms_deform_attn_cuda.cu:34-58 / ms_deform_attn.h:45."""
    MSDA, Fn = _msda()
    value, shapes, lvl, loc, attn, gout = _mk(3, 2, 4, [(4, 4)], 5, 2, torch.float32)
    v, sh, lv, lo, at, go = _cuda(value, shapes, lvl, loc, attn, gout)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_forward(value, shapes, lvl, loc, attn, 2)
    with pytest.raises(RuntimeError, match="contiguous"):
        MSDA.ms_deform_attn_forward(v.transpose(2, 3).contiguous().transpose(2, 3), sh, lv, lo, at, 2)
    with pytest.raises(RuntimeError, match="must divide im2col_step"):
        MSDA.ms_deform_attn_forward(v, sh, lv, lo, at, 2)            # 3 % 2 != 0
    with pytest.raises(RuntimeError, match="must divide im2col_step"):
        MSDA.ms_deform_attn_backward(v, sh, lv, lo, at, go, 2)
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_forward(v.half(), sh, lv, lo.half(), at.half(), 3)
    MSDA.ms_deform_attn_forward(v, sh, lv, lo, at, 3)
    MSDA.ms_deform_attn_forward(v, sh, lv, lo, at, 128)               # min(batch, step)


def test_full_size_linearity_and_torch_crosscheck():
    """BASELINE config-2 geometry (N=2, 1024^2: 32^2+64^2+128^2 tokens, M=8, D=32,
    L=3, P=4): size-independent properties — linearity in value and in the
    attention weights — plus a cross-check against the grid_sample restatement
    evaluated on the GPU."""
    MSDA, Fn = _msda()
    shapes = [(32, 32), (64, 64), (128, 128)]
    S = sum(h * w for h, w in shapes)
    g = torch.Generator(device="cuda").manual_seed(1)
    sh = torch.as_tensor(shapes, dtype=torch.long, device="cuda")
    lv = torch.cat((sh.new_zeros((1,)), sh.prod(1).cumsum(0)[:-1]))
    v1 = torch.randn(2, S, 8, 32, device="cuda", generator=g)
    v2 = torch.randn(2, S, 8, 32, device="cuda", generator=g)
    loc = torch.rand(2, S, 8, 3, 4, 2, device="cuda", generator=g) * 1.2 - 0.1
    a1 = torch.rand(2, S, 8, 3, 4, device="cuda", generator=g)
    a2 = torch.rand(2, S, 8, 3, 4, device="cuda", generator=g)
    f = lambda v, a: MSDA.ms_deform_attn_forward(v, sh, lv, loc, a, 128)
    o11, o21, o12 = f(v1, a1), f(v2, a1), f(v1, a2)
    torch.testing.assert_close(f(2.0 * v1 - 0.5 * v2, a1), 2.0 * o11 - 0.5 * o21, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(f(v1, a1 + 3.0 * a2), o11 + 3.0 * o12, rtol=1e-4, atol=1e-4)
    ref = omsda.msda_torch(v1, sh.cpu(), loc, a1)
    torch.testing.assert_close(o11, ref, rtol=1e-4, atol=1e-4)
    # backward: adjoint identity <out(v), g> == <v, grad_value(g)>  (out is linear in value)
    go = torch.randn(2, S, 256, device="cuda", generator=g)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v1, sh, lv, loc, a1, go, 128)
    lhs = (o11.double() * go.double()).sum()
    rhs = (v1.double() * gv.double()).sum()
    torch.testing.assert_close(lhs, rhs, rtol=1e-5, atol=1e-3)
    # grad_attn is the same contraction with attn replaced by its gradient: <ga, a1> == <out, go>
    torch.testing.assert_close((ga.double() * a1.double()).sum(), lhs, rtol=1e-5, atol=1e-3)
    # autograd of the torch restatement on the same inputs
    v1r, locr, a1r = v1.clone().requires_grad_(), loc.clone().requires_grad_(), a1.clone().requires_grad_()
    omsda.msda_torch(v1r, sh.cpu(), locr, a1r).backward(go)
    torch.testing.assert_close(gv, v1r.grad, rtol=1e-3, atol=1e-3)
    # grad_loc is discontinuous where a sample sits exactly on a pixel boundary (floor() flips between
    # y*H-0.5 and grid_sample's ((2y-1)+1)*H/2-0.5 roundings): allow a 1e-5 fraction of such points
    bad = ~torch.isclose(gl, locr.grad, rtol=1e-3, atol=2e-2)
    assert bad.float().mean().item() < 1e-5
    torch.testing.assert_close(ga, a1r.grad, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("dist", ["0.5", "2", "4", "trained"])
@pytest.mark.parametrize("variant", [0, 1])
def test_backward_config2_geometry_offset_distributions_vs_c_oracle(dist, variant):
    """the LDS-window backward kernels (msda_bwd_owner4_d32, the default, and the per-level msda_bwd_tiled_d32) at
    BASELINE config-2 geometry (1024^2: 32^2 + 64^2 + 128^2 tokens, M = 8, D = 32, L = 3, P = 4; one image so that the scalar C
    oracle finishes in seconds) under the offsets a model actually produces — the initialisation grid of ms_deform_attn.py:70-84
    plus N(0, sigma) cells with sigma in {0.5, 2, 4}, and a heavy-tailed stand-in for a trained model (tools/bench_msda.py): at
    sigma = 4 about 40 % of the bilinear corners fall outside every cached window and take the global-atomic path.  All three
    gradients against the C oracle (ms_deform_im2col_cuda.cuh:92-164 restated, serial fp32 accumulation): the windows accumulate in
    fixed point with a quantum <= 2^-23 of the tile's max |grad_out| per add, so grad_value carries re-association-level noise —
    stated tolerance 2e-5 of the tensor's maximum (measured ~3e-6), grad_loc / grad_attn rtol 1e-4 + 1e-5 of the maximum."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_msda
    from partdistillation_amd import lib
    MSDA, Fn = _msda()
    value, sh, lv, loc, attn, gout = bench_msda.make(1, 1024, px=(0.0 if dist == "trained" else float(dist)),
                                                     offsets=("trained" if dist == "trained" else None), seed=int(float(dist) * 10) if dist != "trained" else 77)
    L = lib.load()
    L.pd_debug_set(b"msda_bwd_variant", variant)
    try:
        gv, gl, ga = MSDA.ms_deform_attn_backward(value, sh, lv, loc, attn, gout, 128)
        torch.cuda.synchronize()
    finally:
        L.pd_debug_set(b"msda_bwd_variant", 0)
    ov, ol, oa = omsda.msda_backward(value.cpu(), sh.cpu(), lv.cpu(), loc.cpu(), attn.cpu(), gout.cpu())
    for name, got, want, rel in (("grad_value", gv, ov, 0.0), ("grad_loc", gl, ol, 1e-4), ("grad_attn", ga, oa, 1e-4)):
        scale = want.abs().max().item()
        err = (got.cpu() - want).abs()
        bound = (2e-5 if name == "grad_value" else 1e-5) * scale + rel * want.abs()
        frac = (err > bound).float().mean().item()
        # grad_loc is a one-sided derivative for samples within rounding distance of a cell border (floor() side): allow 1e-5 of them
        assert frac <= (1e-5 if name == "grad_loc" else 0.0), (name, dist, variant, frac, err.max().item(), scale)


def test_torch_library_operator_matches_the_function_and_passes_opcheck():
    """torch.ops.pd.ms_deform_attn_forward / _backward (registered in partdistillation_amd/MultiScaleDeformableAttention.py) run the same
    kernels as MSDeformAttnFunction, differentiate through register_autograd, and pass torch.library.opcheck (schema, fake
    tensor, autograd registration, AOT dispatch) at Mask2Former head dimensions."""
    MSDA, Fn = _msda()
    value, shapes, lvl, loc, attn, gout = _mk(2, 8, 32, [(8, 8), (4, 4), (2, 2)], 84, 4, torch.float32, seed=5)
    v, sh, lv, lo, at, go = _cuda(value, shapes, lvl, loc, attn, gout)
    v1, lo1, at1 = v.clone().requires_grad_(), lo.clone().requires_grad_(), at.clone().requires_grad_()
    v2, lo2, at2 = v.clone().requires_grad_(), lo.clone().requires_grad_(), at.clone().requires_grad_()
    o1 = Fn.apply(v1, sh, lv, lo1, at1, 128)
    o2 = torch.ops.pd.ms_deform_attn_forward(v2, sh, lv, lo2, at2, 128)
    assert torch.equal(o1, o2)
    o1.backward(go), o2.backward(go)
    torch.testing.assert_close(v1.grad, v2.grad, rtol=1e-5, atol=1e-6)          # atomics: order only
    torch.testing.assert_close(lo1.grad, lo2.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(at1.grad, at2.grad, rtol=1e-5, atol=1e-6)
    torch.library.opcheck(torch.ops.pd.ms_deform_attn_forward.default, (v2.detach().requires_grad_(), sh, lv, lo2.detach().requires_grad_(),
                                                                       at2.detach().requires_grad_(), 128),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration", "test_aot_dispatch_static"))


# ----------------------------------------------------------------------------- the module path's fused form (pd_msda_fused_*)
def _fused_case(shapes, B, seed, off_scale):
    """-> raw projection output oa [B S, 288] (offsets | logits), reference points [B S, 3, 2], value, grad_out and — formed on the CPU
    exactly as ms_deform_attn.py:108-117 does — the sampling locations and attention probabilities they stand for"""
    M, D, L, P = 8, 32, 3, 4
    shapes = torch.as_tensor(shapes, dtype=torch.long)
    lvl = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    value = C.seeded((B, S, M, D), seed + 1)
    offs = C.seeded((B * S, M, L, P, 2), seed + 2, off_scale)                       # pixels
    logits = C.seeded((B * S, M, L * P), seed + 3, 2.0)
    oa = torch.cat([offs.reshape(B * S, -1), logits.reshape(B * S, -1)], 1).contiguous()
    # the encoder's reference points: pixel centres of every level's grid, normalised (msdeformattn.py:88-103 with valid ratios of 1)
    pts = []
    for h, w in shapes.tolist():
        ys, xs = torch.meshgrid(torch.linspace(0.5, h - 0.5, h) / h, torch.linspace(0.5, w - 0.5, w) / w, indexing="ij")
        pts.append(torch.stack((xs.reshape(-1), ys.reshape(-1)), -1))
    ref = torch.cat(pts)[None, :, None, :].expand(B, S, L, 2).reshape(B * S, L, 2).contiguous()
    gout = C.seeded((B, S, M * D), seed + 4)
    return value, shapes, lvl, oa, ref, gout, (B, S, M, D, L, P)


def _loc_attn(oa, ref, shapes, dims):
    B, S, M, D, L, P = dims
    n = M * L * P
    norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).to(oa.dtype)               # (W, H)
    loc = ref[:, None, :, None, :] + oa[:, :2 * n].view(B * S, M, L, P, 2) / norm[None, None, :, None, :]
    attn = oa[:, 2 * n:].view(B * S, M, L * P).softmax(-1).view(B * S, M, L, P)
    return loc.view(B, S, M, L, P, 2), attn.view(B, S, M, L, P)


@pytest.mark.parametrize("variant", [0, 2, 3])
@pytest.mark.parametrize("shapes,off_scale", [([(32, 32), (16, 16), (8, 8)], 2.0), ([(24, 20), (12, 10), (6, 5)], 3.0), ([(64, 64), (32, 32), (16, 16)], 9.0)])
def test_fused_forward_backward_vs_oracle(shapes, off_scale, variant):
    """pd_msda_fused_forward / _backward (softmax over a head's 12 logits and reference point + offset / (W, H) formed inside the
    kernels, the gradient of the RAW projection output written by the backward kernel) against the C oracle run on the locations /
    probabilities torch forms from the same numbers, chained through their definition in float64 (ms_deform_attn.py:108-117):
    power-of-two and other level sizes, offsets of a few pixels (window hits) up to ~30 (the global-atomic path), both LDS-window
    variants forced and the measured choice.  Forward rtol 1e-5; grad_value as the unfused kernel's test (2e-5 of its maximum);
    d_oa rtol 1e-4 + 2e-5 of its maximum; the rows' maxima exact for what was written."""
    from partdistillation_amd import lib
    from partdistillation_amd.functions import encoder_core as EC
    value, sh, lvl, oa, ref, gout, dims = _fused_case(shapes, 2, 11 + int(off_scale), off_scale)
    B, S, M, D, L, P = dims
    assert EC.msda_fused_supported(B, S, M, D, L, S, P)
    loc, attn = _loc_attn(oa, ref, sh, dims)
    want = omsda.msda_forward(value, sh, lvl, loc, attn)
    v, shd, lvd, oad, refd, god = _cuda(value, sh, lvl, oa, ref, gout)
    am = torch.zeros(B * S, device="cuda")
    out, stats = EC.msda_fused_forward(v, shd, lvd, oad, refd, am)
    torch.testing.assert_close(out.cpu().view(B, S, M * D), want, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(am.cpu(), out.abs().amax(1).cpu(), rtol=0, atol=0)
    lg = oa[:, 2 * M * L * P:].reshape(B * S * M, L * P)
    torch.testing.assert_close(stats[:, 0].cpu(), lg.amax(1), rtol=0, atol=0)
    torch.testing.assert_close(stats[:, 1].cpu(), 1.0 / (lg - lg.amax(1, keepdim=True)).exp().sum(1), rtol=1e-5, atol=0)
    Lh = lib.load()
    Lh.pd_debug_set(b"msda_bwd_variant", variant)
    try:
        gv, d_oa, d_am = EC.msda_fused_backward(v, shd, lvd, oad, refd, stats, out, god)
        torch.cuda.synchronize()
    finally:
        Lh.pd_debug_set(b"msda_bwd_variant", 0)
    ov, ol, og = omsda.msda_backward(value, sh, lvl, loc, attn, gout)
    # chain rule in float64: d offset = grad_loc / (W, H); d logit = a (g - sum a g)
    norm = torch.stack([sh[:, 1], sh[:, 0]], -1).double()
    d_off = ol.double().view(B * S, M, L, P, 2) / norm[None, None, :, None, :]
    a64, g64 = attn.double().view(B * S, M, L * P), og.double().view(B * S, M, L * P)
    d_lg = a64 * (g64 - (a64 * g64).sum(-1, keepdim=True))
    want_doa = torch.cat([d_off.reshape(B * S, -1), d_lg.reshape(B * S, -1)], 1).float()
    scale = ov.abs().max().item()
    assert ((gv.cpu() - ov).abs() <= 2e-5 * scale).all(), (gv.cpu() - ov).abs().max().item() / scale
    err = (d_oa.cpu() - want_doa).abs()
    bound = 2e-5 * want_doa.abs().max() + 1e-4 * want_doa.abs()
    # a sample within rounding distance of a cell border takes the other one-sided derivative (as in the unfused test): 1e-5 of them
    assert (err > bound).float().mean().item() <= 1e-5, (err.max().item(), want_doa.abs().max().item())
    torch.testing.assert_close(d_am.cpu(), d_oa.abs().amax(1).cpu(), rtol=0, atol=0)


def test_fused_path_equals_prep_plus_operator_in_the_encoder_layer():
    """the encoder core with PD_MSDA_FUSED on and off (pd_msda_prep_fwd + pd_msda_forward_amax / pd_msda_backward + pd_msda_prep_bwd_amax):
    same outputs and gradients to reordered-atomics noise — and the fused run really took the fused kernels."""
    from partdistillation_amd.functions import encoder_core as EC
    torch.manual_seed(0)
    cfg = {**C.TINY, "conv_dim": 256, "mask_dim": 256, "enc_ffn": 512, "batch": 2}       # 8 heads x 32 channels: the geometry the fused kernels serve
    pd = build_pixel_decoder_small(cfg)
    feats = {k: v.cuda().requires_grad_() for k, v in C.make_features(cfg, 231).items()}
    res = {}
    calls = {"n": 0}
    f0 = EC.msda_fused_forward
    EC.msda_fused_forward = lambda *a: (calls.__setitem__("n", calls["n"] + 1), f0(*a))[1]
    try:
        for fused in (True, False):
            EC.FUSED_MSDA = fused
            for p in pd.parameters():
                p.grad = None
            mf, enc, ms = pd.forward_features(feats)
            loss = (mf * C.seeded(mf.shape, 5).cuda()).sum() + sum((m * C.seeded(m.shape, 6 + i).cuda()).sum() for i, m in enumerate(ms))
            gf = torch.autograd.grad(loss, list(feats.values()) + [p for p in pd.parameters() if p.requires_grad], allow_unused=True)
            res[fused] = ([mf.detach()] + [m.detach() for m in ms], [g.detach() if g is not None else None for g in gf])
    finally:
        EC.FUSED_MSDA, EC.msda_fused_forward = True, f0
    assert calls["n"] >= cfg["enc_layers"]
    for a, b in zip(res[True][0], res[False][0]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
    for a, b in zip(res[True][1], res[False][1]):
        assert (a is None) == (b is None)
        if a is not None:
            assert ((a - b).abs().max() <= 2e-4 * b.abs().max().clamp_min(1e-6)).item(), ((a - b).abs().max().item(), b.abs().max().item())


def build_pixel_decoder_small(cfg):
    from partdistillation_amd.compat import ShapeSpec
    from partdistillation_amd.modeling.pixel_decoder.msdeformattn import MSDeformAttnPixelDecoder
    specs = {f"res{i + 2}": ShapeSpec(channels=c, stride=s) for i, (c, s) in enumerate(zip(cfg["channels"], (4, 8, 16, 32)))}
    return MSDeformAttnPixelDecoder(specs, transformer_dropout=0.0, transformer_nheads=cfg["nheads"], transformer_dim_feedforward=cfg["enc_ffn"],
                                    transformer_enc_layers=cfg["enc_layers"], conv_dim=cfg["conv_dim"], mask_dim=cfg["mask_dim"], norm="GN",
                                    transformer_in_features=["res3", "res4", "res5"], common_stride=4).cuda()
