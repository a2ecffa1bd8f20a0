"""GPU tests of the fp32 MFMA GEMMs (pd_gemm.h) against torch fp64 references (tolerance = fp32 round-off of a
K-long dot product), including ragged tails and the autograd wrapper."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(a, b):
    return (a.double() @ b.double().t())


@pytest.mark.parametrize("M,N,K", [(43008, 1024, 256), (43008, 256, 1024), (43008, 288, 256), (1000, 100, 36), (130, 260, 8),
                                   (1, 4, 4), (257, 129, 132)])
def test_gemm_tn_matches_fp64(M, N, K):
    from partdistillation_amd.functions import gemm
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g)
    b = torch.randn(N, K, device="cuda", generator=g)            # asymmetric operands: catches transposes
    bias = torch.randn(N, device="cuda", generator=g)
    want = _ref(a, b) + bias.double()
    tol = dict(rtol=1e-5, atol=2e-6 * K ** 0.5 * 4)
    torch.testing.assert_close(gemm.gemm_tn(a, b, bias).double(), want, **tol)
    torch.testing.assert_close(gemm.gemm_tn(a, b, None, relu=True).double(), _ref(a, b).clamp_min(0), **tol)


@pytest.mark.parametrize("M,N,K", [(43008, 1024, 256), (43008, 256, 1024), (43008, 288, 256), (1000, 100, 36), (7, 4, 8), (513, 132, 260)])
def test_gemm_wgrad_matches_fp64(M, N, K):
    from partdistillation_amd.functions import gemm
    g = torch.Generator(device="cuda").manual_seed(M * 3 + N + K)
    dy = torch.randn(M, N, device="cuda", generator=g)
    x = torch.randn(M, K, device="cuda", generator=g)
    want = dy.double().t() @ x.double()
    torch.testing.assert_close(gemm.gemm_wgrad(dy, x).double(), want, rtol=1e-5, atol=2e-6 * M ** 0.5 * 4)
    dw, db = gemm.gemm_wgrad(dy, x, with_bias=True)
    torch.testing.assert_close(dw.double(), want, rtol=1e-5, atol=2e-6 * M ** 0.5 * 4)
    torch.testing.assert_close(db.double(), dy.double().sum(0), rtol=1e-5, atol=2e-6 * M ** 0.5 * 4)


def test_linear_f32_autograd_matches_torch():
    from partdistillation_amd.functions import gemm as G
    from partdistillation_amd.functions.gemm import linear_f32
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(2, 300, 64, device="cuda", generator=g, requires_grad=True)
    w = torch.randn(96, 64, device="cuda", generator=g, requires_grad=True)
    b = torch.randn(96, device="cuda", generator=g, requires_grad=True)
    go = torch.randn(2, 300, 96, device="cuda", generator=g)
    for relu, all_mfma in ((False, False), (True, False), (False, True), (True, True)):
        G.ALL_MFMA = all_mfma
        y = linear_f32(x, w, b, relu)
        ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
        ref = ref.relu() if relu else ref
        torch.testing.assert_close(y.double(), ref, rtol=1e-5, atol=1e-4)
        gx, gw, gb = torch.autograd.grad(y, (x, w, b), go)
        rx, rw, rb = torch.autograd.grad(ref, (x, w, b), go.double())
        torch.testing.assert_close(gx.double(), rx.double(), rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(gw.double(), rw.double(), rtol=1e-5, atol=1e-3)
        torch.testing.assert_close(gb.double(), rb.double(), rtol=1e-5, atol=1e-3)
    G.ALL_MFMA = False


@pytest.mark.parametrize("M,N,K,relu", [(43008, 1024, 256, True), (43008, 256, 1024, False), (1344, 288, 256, False), (300, 70, 36, False),
                                        (129, 257, 20, True)])
def test_gemm_tn_x3_is_fp32_accurate(M, N, K, relu):
    """pd_gemm_tn_f32x3 (exact 3-way bf16 split of the fp32 operands, 6 partial products on the bf16 matrix cores) against
    fp64: its error must stay at the level of the library's own fp32 GEMM (<= 1.5x its maximum error + 1 ulp of the scale),
    on operands with a wide dynamic range (the split is exact for any finite fp32 value)."""
    from partdistillation_amd.functions import gemm
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g) * torch.exp(torch.randn(M, 1, device="cuda", generator=g) * 2)
    w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    b = torch.randn(N, device="cuda", generator=g)
    ref = torch.addmm(b.double(), a.double(), w.double().t())
    lib = torch.addmm(b, a, w.t())
    if relu:
        ref, lib = ref.relu(), lib.relu()
    got = gemm.gemm_tn_x3(a, w, b, relu)
    assert got.dtype == torch.float32 and got.shape == (M, N)
    row_scale = (a.double().abs() @ w.double().abs().t() + b.double().abs())               # per-element error scale sum |a||w|
    e_x3 = ((got.double() - ref).abs() / row_scale).max().item()
    e_lib = ((lib.double() - ref).abs() / row_scale).max().item()
    assert e_x3 <= 1.5 * e_lib + 2 ** -23, (e_x3, e_lib)
    assert e_x3 < 2e-6


@pytest.mark.parametrize("M,N,K,bias", [(43008, 1024, 256, True), (43008, 256, 1024, False), (5000, 288, 256, True), (333, 70, 36, True),
                                        (17, 130, 257, False), (64, 128, 128, True)])
def test_gemm_wgrad_x3_is_fp32_accurate(M, N, K, bias):
    """pd_gemm_wgrad_acc_f32x3 against fp64, next to the exact-fp32 MFMA kernel it replaces: dW += dY^T X over the rows
    (M up to 43008: errors of both are dominated by the fp32 accumulation order), dB exact fp32 column sums."""
    from partdistillation_amd.functions import gemm
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    dy = torch.randn(M, N, device="cuda", generator=g) * torch.exp(torch.randn(M, 1, device="cuda", generator=g))
    x = torch.randn(M, K, device="cuda", generator=g)
    ref = dy.double().t() @ x.double()
    outs = {}
    for x3 in ((True, False) if N % 4 == 0 and K % 4 == 0 else (True,)):           # the exact kernel needs N, K multiples of 4
        dw = torch.full((N, K), 0.5, device="cuda")
        db = torch.full((N,), -1.0, device="cuda") if bias else None
        gemm.gemm_wgrad_acc(dy, x, dw, db, x3=x3)
        outs[x3] = (dw, db)
    scale = (dy.double().abs().t() @ x.double().abs())
    e_x3 = ((outs[True][0].double() - 0.5 - ref).abs() / scale).max().item()
    if False in outs:
        e_f32 = ((outs[False][0].double() - 0.5 - ref).abs() / scale).max().item()
        assert e_x3 <= 2.0 * e_f32 + 2 ** -22, (e_x3, e_f32)
    assert e_x3 < 3e-6
    if bias:
        torch.testing.assert_close(outs[True][1].double() + 1.0, dy.double().sum(0), rtol=1e-4, atol=1e-3 * dy.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(1, 128, 128), (31, 132, 260), (4100, 40, 256), (2048, 516, 68)])
def test_gemm_wgrad_x3_transpose_read_edges(M, N, K):
    """the transpose-read kernel on ragged tiles (N, K no multiples of 128), a single row, row-strided operands (column slices of
    wider matrices) — through the workspace ABI and through the atomics one (pd_gemm_wgrad_acc_f32x3, no workspace)."""
    from partdistillation_amd import lib
    from partdistillation_amd.functions import gemm
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N + K)
    big_y = torch.randn(M, N + 24, device="cuda", generator=g)
    big_x = torch.randn(M, K + 12, device="cuda", generator=g)
    dy, x = big_y[:, 8:8 + N], big_x[:, 4:4 + K]                      # 16-byte aligned column windows, ld > width
    assert dy.stride(0) == N + 24 and x.stride(0) == K + 12
    ref = dy.double().t() @ x.double()
    scale = dy.double().abs().t() @ x.double().abs() + 1e-30
    dw, db = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
    gemm.gemm_wgrad_acc(dy, x, dw, db, x3=True)
    assert ((dw.double() - ref).abs() / scale).max().item() < 3e-6
    torch.testing.assert_close(db.double(), dy.double().sum(0), rtol=1e-4, atol=1e-3)
    dw2, db2 = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
    lib.check(lib.load().pd_gemm_wgrad_acc_f32x3(dy.data_ptr(), x.data_ptr(), dw2.data_ptr(), db2.data_ptr(), M, N, K, dy.stride(0),
                                                 x.stride(0), K, lib.current_stream()))
    assert ((dw2.double() - ref).abs() / scale).max().item() < 3e-6
    torch.testing.assert_close(db2, db, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("B,H,W,Ci,Co,bias", [(2, 64, 64, 256, 256, False), (1, 37, 29, 32, 48, True), (3, 8, 200, 16, 272, True),
                                              (3, 19, 23, 128, 80, True), (2, 5, 7, 256, 256, True)])
@pytest.mark.parametrize("h2", [True, False])
def test_conv3x3_x3_matches_fp64_convolution(B, H, W, Ci, Co, bias, h2, monkeypatch):
    """pd_conv3x3_nhwc_f32x3 (implicit GEMM on the 3-way bf16 split) forward, input gradient (the same kernel on dY with the
    flipped, transposed filter) and the weight / bias gradient (pd_conv3x3_wgrad_nhwc_f32x3 where Ci % 128 == 0: the transpose-read
    split kernel with the im2col gather in its staging — images narrower than its 16-row stage, pixel counts that are no multiple
    of anything; the library's otherwise) against an fp64 convolution; errors at the level of the library's fp32 convolution."""
    import torch.nn.functional as F
    from partdistillation_amd.functions import conv_x3
    monkeypatch.setattr(conv_x3, "H2", h2)            # the fp16 two-plane form (pd_conv3x3_nhwc_f16x2 / _wgrad_) and the 3-plane bf16 one
    g = torch.Generator(device="cuda").manual_seed(H * W + Ci)
    x = torch.randn(B, Ci, H, W, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    if h2:                                             # pixels of very different magnitude: the per-pixel scales matter
        x = x * torch.logspace(-2, 2, H * W, device="cuda")[torch.randperm(H * W, device="cuda", generator=g)].view(1, 1, H, W)
    x = x.contiguous(memory_format=torch.channels_last).requires_grad_()
    w = (torch.randn(Co, Ci, 3, 3, device="cuda", generator=g) * (9 * Ci) ** -0.5).contiguous(memory_format=torch.channels_last).requires_grad_()
    b = torch.randn(Co, device="cuda", generator=g).requires_grad_() if bias else None
    go = torch.randn(B, Co, H, W, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(Ci, Co, 3, padding=1, bias=bias).cuda()
    assert conv_x3.supported(x, conv)
    y = conv_x3.conv3x3(x, w, b)
    gx, gw = torch.autograd.grad(y, (x, w), go, retain_graph=bias)
    xr, wr = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
    yr = F.conv2d(xr, wr, b.detach().double() if bias else None, padding=1)
    rx, rw = torch.autograd.grad(yr, (xr, wr), go.double())
    yl = F.conv2d(x.detach(), w.detach(), b.detach() if bias else None, padding=1)

    def err(a, r):
        return ((a.double() - r).abs().max() / r.abs().max()).item()
    assert y.is_contiguous(memory_format=torch.channels_last) and y.shape == (B, Co, H, W)
    # the two-plane form carries 22 significand bits per operand (fp32: 24): its bound is one binade wider
    assert err(y, yr) <= 2.0 * err(yl, yr) + 2 ** (-21 if h2 else -22), (err(y, yr), err(yl, yr))
    assert err(gx, rx) < 3e-6 and err(gw, rw) < 2e-5
    if bias:
        (gb,) = torch.autograd.grad(y, (b,), go)
        torch.testing.assert_close(gb.double(), go.double().sum((0, 2, 3)), rtol=1e-4, atol=1e-3)



@pytest.mark.parametrize("M,N,K,K2", [(4096, 1024, 256, 256), (3000, 256, 64, 128), (1024, 512, 400, 36)])
def test_gemm_tn_x3_relu_bits_and_relumask_epilogues(M, N, K, K2):
    """pd_gemm_tn_f32x3_relu_bits (forward: relu(A B^T + b) + sign bits) and pd_gemm_tn_f32x3_relumask (backward: (G W) masked by
    those bits, column sums accumulated) against fp64 and the two-step path they replace."""
    from partdistillation_amd.functions import gemm
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    b = torch.randn(N, device="cuda", generator=g)
    h, bits = gemm.gemm_tn_x3_relu_bits(a, w, b)
    href = torch.addmm(b.double(), a.double(), w.double().t()).relu()
    assert ((h.double() - href).abs().max().item() / href.abs().max().item()) < 2e-6
    g2 = torch.randn(M, K2, device="cuda", generator=g)
    w2 = torch.randn(N, K2, device="cuda", generator=g) * K2 ** -0.5
    acc = torch.full((N,), 0.25, device="cuda")
    got = gemm.gemm_tn_x3_relumask(g2, w2, bits, acc)
    ref = (g2.double() @ w2.double().t()) * (h > 0)
    scale = ref.abs().max().item()
    assert ((got.double() - ref).abs().max().item() / scale) < 2e-6
    assert torch.equal(got != 0, (h > 0) & (ref != 0))
    torch.testing.assert_close(acc.double() - 0.25, ref.sum(0), rtol=1e-4, atol=1e-3 * scale)


@pytest.mark.parametrize("M,N,K", [(40000, 1024, 256), (33000, 256, 1024), (65536, 512, 64)])
def test_gemm_tn_x3_with_presplit_weight_planes(M, N, K):
    """pd_split3_bf16 (exact: hi + mid + lo == W, also for the transposed planes) and pd_gemm_tn_f32x3_pre: bit-identical to
    pd_gemm_tn_f32x3's wide kernel on the same operands (the same six products in the same order), fp32-accurate against fp64."""
    from partdistillation_amd import lib
    from partdistillation_amd.functions import gemm
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    b = torch.randn(N, device="cuda", generator=g)
    pl = gemm.split3(w)
    assert torch.equal(pl[0].float() + pl[1].float() + pl[2].float(), w)                    # exact 3-way split
    plt = gemm.split3(w, transpose=True)
    assert plt.shape == (3, K, N) and torch.equal(plt.float().sum(0), w.t())
    got = gemm.gemm_tn_x3_pre(a, pl, b)
    lib.load().pd_debug_set(b"x3_narrow", 2)                                                # force the 256 x 256 kernel for the comparison
    same = gemm.gemm_tn_x3(a, w, b)
    lib.load().pd_debug_set(b"x3_narrow", 0)
    assert torch.equal(got, same)
    ref = torch.addmm(b.double(), a.double(), w.double().t())
    assert ((got.double() - ref).abs().max().item() / ref.abs().max().item()) < 2e-6
    h, bits = gemm.gemm_tn_x3_pre(a, pl, b, mode=1, want_bits=True)
    assert torch.equal(h, same.relu())
    acc = torch.zeros(N, device="cuda")
    masked = gemm.gemm_tn_x3_pre(a, pl, None, mode=2, bits=bits, colsum=acc)
    want = (a.double() @ w.double().t()) * (h > 0)
    assert ((masked.double() - want).abs().max().item() / want.abs().max().item()) < 2e-6
    torch.testing.assert_close(acc.double(), want.sum(0), rtol=1e-4, atol=1e-3 * want.abs().max().item())


def test_grouped_weight_gradients_match_fp64_and_the_single_launches():
    """pd_gemm_wgrad_f32x3_grouped (functions/gemm.WgradQueue): one encoder layer's five weight gradients at BASELINE config-2 size
    (43 008 tokens: 256 x 1024, 1024 x 256, 256 x 256, 288 x 256 with its bias gradient, 256 x 256) plus a ragged problem, as ONE
    launch accumulated INTO pre-filled buffers, against fp64 and against the one-launch-per-gradient path."""
    from partdistillation_amd.functions import gemm
    g = torch.Generator(device="cuda").manual_seed(11)
    T = 43008
    probs = [(T, 256, 1024, False), (T, 1024, 256, False), (T, 256, 256, True), (T, 288, 256, True), (T, 256, 256, False), (1000, 100, 36, True)]
    q = gemm.WgradQueue()
    keep = []
    for M, N, K, bias in probs:
        dy = torch.randn(M, N, device="cuda", generator=g)
        x = torch.randn(M, K, device="cuda", generator=g)
        dw0 = torch.randn(N, K, device="cuda", generator=g)                # accumulate semantics: dW += ...
        db0 = torch.randn(N, device="cuda", generator=g) if bias else None
        dw, db = dw0.clone(), (db0.clone() if bias else None)
        q.add(dy, x, dw, db)
        keep.append((dy, x, dw0, db0, dw, db))
    q.flush()
    for (M, N, K, bias), (dy, x, dw0, db0, dw, db) in zip(probs, keep):
        want = dw0.double() + dy.double().t() @ x.double()
        tol = dict(rtol=1e-5, atol=2e-6 * M ** 0.5 * 4)
        torch.testing.assert_close(dw.double(), want, **tol)
        single = dw0.clone()
        sb = db0.clone() if bias else None
        gemm.gemm_wgrad_acc(dy, x, single, sb, x3=True)
        torch.testing.assert_close(dw, single, rtol=1e-5, atol=1e-6 * M ** 0.5 * 4)
        if bias:
            torch.testing.assert_close(db.double(), db0.double() + dy.double().sum(0), **tol)


@pytest.mark.parametrize("M,N,K", [(1500, 256, 256), (2048, 1024, 256), (1100, 256, 1024), (333, 72, 40), (4096, 288, 256)])
@pytest.mark.parametrize("mag", [1.0, 1e-7, 3e4])
def test_gemm_tn_h2_is_fp32_accurate_over_the_exponent_range(M, N, K, mag):
    """pd_gemm_tn_f16x2 (two fp16 planes, three products, rows scaled from their maxima): normwise error vs fp64 at the level of the
    library's fp32 GEMM for operands from 1e-7 to 3e4 (gradients to activations), rows of very different magnitude, every tile shape;
    the row maxima it emits are exact."""
    from partdistillation_amd import lib
    from partdistillation_amd.functions import gemm
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda") * mag * torch.logspace(-3, 3, M, device="cuda")[torch.randperm(M, device="cuda"), None]
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda") * mag
    ref = torch.addmm(b.double(), a.double(), w.double().t())
    rown = ref.abs().amax(1, keepdim=True).clamp_min(1e-300)                         # every ROW is judged against its own scale
    e_lib = ((torch.addmm(b, a, w.t()).double() - ref).abs() / rown).max().item()
    aa, wa = gemm.row_amax(a), gemm.row_amax(w)
    assert torch.equal(aa, a.abs().amax(1))
    # 0: the product's choice (interleaved interior step); 70: the guarded step by shape; 4 / 14, 3 / 13: forced tile shapes and register stages
    for tile in (0, 70, 4, 14) + ((3, 13) if N % 256 == 0 and M >= 1024 else ()):
        lib.load().pd_debug_set(b"f16x2_tile", tile)
        try:
            cm = torch.zeros(M, device="cuda")
            got = gemm.gemm_tn_h2(a, w, b, a_amax=aa, b_amax=wa, c_amax=cm)
        finally:
            lib.load().pd_debug_set(b"f16x2_tile", 0)
        e = ((got.double() - ref).abs() / rown).max().item()
        assert e <= 2.0 * e_lib + 2 ** -21 and e < 3e-6, (tile, e, e_lib)
        assert torch.equal(cm, got.abs().amax(1))
    r = gemm.gemm_tn_h2(a, w, b, mode=1, a_amax=aa, b_amax=wa)
    assert ((r.double() - ref.clamp_min(0)).abs() / rown).max().item() < 3e-6


@pytest.mark.parametrize("M,N,K", [(8300, 256, 256), (9001, 512, 256), (43520, 1024, 256)])
@pytest.mark.parametrize("scaled", [True, False])
def test_gemm_tn_h2_row_stream_matches_the_tiled_kernel(M, N, K, scaled):
    """gemm_rows_f16x2_k256 (the experimental persistent row-stream form of the K = 256 shapes, pd_debug_set("f16x2_tile", 61)):
    fp32-accurate against fp64, exact row maxima, ragged last tile, several column panels."""
    from partdistillation_amd import lib
    from partdistillation_amd.functions import gemm
    torch.manual_seed(M + N)
    a = torch.randn(M, K, device="cuda") * (torch.logspace(-3, 3, M, device="cuda")[torch.randperm(M, device="cuda"), None] if scaled else 1.0)
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda")
    ref = torch.addmm(b.double(), a.double(), w.double().t())
    rown = ref.abs().amax(1, keepdim=True)
    aa, wa = (gemm.row_amax(a), gemm.row_amax(w)) if scaled else (None, None)
    lib.load().pd_debug_set(b"f16x2_tile", 80)                      # (the tiled kernel: the default takes the row stream for K = 256 since round 5)
    try:
        tiled = gemm.gemm_tn_h2(a, w, b, a_amax=aa, b_amax=wa)
    finally:
        lib.load().pd_debug_set(b"f16x2_tile", 0)
    lib.load().pd_debug_set(b"f16x2_tile", 61)
    try:
        cm = torch.zeros(M, device="cuda")
        got = gemm.gemm_tn_h2(a, w, b, a_amax=aa, b_amax=wa, c_amax=cm)
        again = gemm.gemm_tn_h2(a, w, b, a_amax=aa, b_amax=wa)
    finally:
        lib.load().pd_debug_set(b"f16x2_tile", 0)
    assert torch.equal(got, again)
    assert ((got.double() - ref).abs() / rown).max().item() < 3e-6
    assert ((got.double() - tiled.double()).abs() / rown).max().item() < 1e-6
    assert torch.equal(cm, got.abs().amax(1))


@pytest.mark.parametrize("M,N,K", [(43008, 256, 1024), (8300, 512, 512), (9001, 256, 576), (8192, 256, 2048)])
@pytest.mark.parametrize("scaled", [True, False])
@pytest.mark.parametrize("variant", [0])        # the product's choice for these shapes: producer / consumer wavefronts (gemm_kpc_f16x2); the opt-in forms: tools/probes/test_optin_kernels.py
def test_gemm_tn_h2_producer_consumer_kernel_matches_fp64_and_the_tiled_kernel(M, N, K, scaled, variant):
    """gemm_kpc_f16x2 (deep K, 256-column panels: K outside, a 192-row block's accumulators resident, producer / consumer wavefronts):
    fp32-accurate against fp64, within 1e-6 of the tiled kernel, ragged last row block, two column panels,
    K not a power of two, with / without bias and row scales."""
    from partdistillation_amd import lib
    from partdistillation_amd.functions import gemm
    L = lib.load()
    torch.manual_seed(M + K)
    a = torch.randn(M, K, device="cuda") * (torch.logspace(-3, 3, M, device="cuda")[torch.randperm(M, device="cuda"), None] if scaled else 1.0)
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda") if M % 2 else None
    ref = a.double() @ w.double().t() + (b.double() if b is not None else 0.0)
    rown = ref.abs().amax(1, keepdim=True)
    aa, wa = (gemm.row_amax(a), gemm.row_amax(w)) if scaled else (None, None)
    L.pd_debug_set(b"f16x2_tile", 80)
    try:
        tiled = gemm.gemm_tn_h2(a, w, b, a_amax=aa, b_amax=wa)
    finally:
        L.pd_debug_set(b"f16x2_tile", 0)
    L.pd_debug_set(b"f16x2_tile", variant)
    try:
        assert L.pd_gemm_tn_f16x2_which(M, N, K, 0, 0, int(scaled)) == 5
        got = gemm.gemm_tn_h2(a, w, b, a_amax=aa, b_amax=wa)
        again = gemm.gemm_tn_h2(a, w, b, a_amax=aa, b_amax=wa)
    finally:
        L.pd_debug_set(b"f16x2_tile", 0)
    assert torch.equal(got, again)
    assert ((got.double() - ref).abs() / rown).max().item() < 3e-6
    assert ((got.double() - tiled.double()).abs() / rown).max().item() < 1e-6


def test_gemm_tn_h2_relu_bits_and_mask_epilogues():
    from partdistillation_amd.functions import gemm
    torch.manual_seed(5)
    M, N, K = 2304, 512, 256
    x = torch.randn(M, K, device="cuda"); w1 = torch.randn(N, K, device="cuda") * K ** -0.5; b1 = torch.randn(N, device="cuda")
    h, bits = gemm.gemm_tn_h2(x, w1, b1, mode=1, want_bits=True, b_amax=gemm.row_amax(w1))
    href = torch.addmm(b1.double(), x.double(), w1.double().t()).clamp_min(0)
    assert ((h.double() - href).abs().max().item() / href.abs().max().item()) < 2e-6
    dy = torch.randn(M, K, device="cuda") * 1e-5; w2t = torch.randn(N, K, device="cuda") * K ** -0.5
    acc = torch.full((N,), 0.25, device="cuda")
    cm = torch.zeros(M, device="cuda")
    got = gemm.gemm_tn_h2(dy, w2t, None, mode=2, bits=bits, colsum=acc, a_amax=gemm.row_amax(dy), b_amax=gemm.row_amax(w2t), c_amax=cm)
    ref = (dy.double() @ w2t.double().t()) * (h > 0)
    scale = ref.abs().max().item()
    assert ((got.double() - ref).abs().max().item() / scale) < 2e-6
    assert torch.equal(got != 0, (h > 0) & (ref != 0))
    assert torch.equal(cm, got.abs().amax(1))
    torch.testing.assert_close(acc.double() - 0.25, ref.sum(0), rtol=1e-4, atol=1e-2 * scale)   # 0.25 + 2 304 fp32 atomics of ~1e-5 terms


@pytest.mark.parametrize("M", [8300, 43008])
@pytest.mark.parametrize("tile", [0, 62])
def test_gemm_tn_h2_relu_epilogues_on_the_row_stream(M, tile):
    """N = 1024 <- K = 256 with the ReLU epilogues (the encoder FFN: linear1 + ReLU + sign bits forward, the masked input gradient + bias
    column sums backward) on the row-stream kernel (round 5: its own bit order) and, with pd_debug_set("f16x2_tile", 62), on the tiled
    kernel: fp32-accurate against fp64, the mask of the backward launch = the forward's ReLU pattern, exact row maxima, ragged last tile."""
    from partdistillation_amd import lib
    from partdistillation_amd.functions import gemm
    torch.manual_seed(M)
    N, K = 1024, 256
    x = torch.randn(M, K, device="cuda") * (1 + 5 * torch.rand(M, 1, device="cuda")); w1 = torch.randn(N, K, device="cuda") * K ** -0.5; b1 = torch.randn(N, device="cuda")
    lib.load().pd_debug_set(b"f16x2_tile", tile)
    try:
        hm = torch.zeros(M, device="cuda")
        h, bits = gemm.gemm_tn_h2(x, w1, b1, mode=1, want_bits=True, a_amax=gemm.row_amax(x), b_amax=gemm.row_amax(w1), c_amax=hm)
        href = torch.addmm(b1.double(), x.double(), w1.double().t()).clamp_min(0)
        assert ((h.double() - href).abs().max().item() / href.abs().max().item()) < 3e-6
        assert torch.equal(hm, h.abs().amax(1))
        dy = torch.randn(M, K, device="cuda") * 1e-4; w2t = torch.randn(N, K, device="cuda") * K ** -0.5
        acc = torch.zeros(N, device="cuda")
        cm = torch.zeros(M, device="cuda")
        got = gemm.gemm_tn_h2(dy, w2t, None, mode=2, bits=bits, colsum=acc, a_amax=gemm.row_amax(dy), b_amax=gemm.row_amax(w2t), c_amax=cm)
    finally:
        lib.load().pd_debug_set(b"f16x2_tile", 0)
    ref = (dy.double() @ w2t.double().t()) * (h > 0)
    scale = ref.abs().max().item()
    assert ((got.double() - ref).abs().max().item() / scale) < 3e-6
    assert torch.equal(got != 0, (h > 0) & (ref != 0))
    assert torch.equal(cm, got.abs().amax(1))
    torch.testing.assert_close(acc.double(), ref.sum(0), rtol=1e-4, atol=2e-5 * scale * M ** 0.5)


@pytest.mark.parametrize("M,N,K", [(43008, 1024, 256), (5000, 288, 256), (2100, 256, 1024), (700, 72, 40)])
@pytest.mark.parametrize("mag", [1.0, 1e-6])
def test_gemm_wgrad_h2_matches_fp64(M, N, K, mag):
    """pd_gemm_wgrad_acc_f16x2_ws / pd_gemm_wgrad_f16x2_grouped: dW += dY^T X with slab-wise power-of-two scales from the row maxima —
    normwise error vs fp64 at the exact-fp32 kernel's level, for gradient-sized dY with rows of very different magnitude."""
    from partdistillation_amd.functions import gemm
    torch.manual_seed(M + N)
    dy = torch.randn(M, N, device="cuda") * mag * torch.logspace(-2, 2, M, device="cuda")[torch.randperm(M, device="cuda"), None]
    x = torch.randn(M, K, device="cuda") * (1 + 5 * torch.rand(M, 1, device="cuda"))
    ref = dy.double().t() @ x.double()
    scale = ref.abs().max().item()
    ya, xa = gemm.row_amax(dy), gemm.row_amax(x)
    dw32 = torch.zeros(N, K, device="cuda"); gemm.gemm_wgrad_acc(dy, x, dw32, x3=False)
    e32 = ((dw32.double() - ref).abs().max().item() / scale)
    dw = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
    gemm.gemm_wgrad_acc(dy, x, dw, db, h2=True, y_amax=ya, x_amax=xa)
    e = ((dw.double() - ref).abs().max().item() / scale)
    assert e <= 2.0 * e32 + 2 ** -21 and e < 3e-6, (e, e32)
    torch.testing.assert_close(db.double(), dy.double().sum(0), rtol=1e-4, atol=1e-3 * dy.abs().max().item())
    q = gemm.WgradQueue(h2=True)
    dw2 = torch.zeros(N, K, device="cuda"); dw3 = torch.zeros(K, K, device="cuda"); db2 = torch.zeros(N, device="cuda")
    q.add(dy, x, dw2, db2, ya, xa)
    q.add(x, x, dw3, None, xa, xa)
    q.flush()
    assert ((dw2.double() - ref).abs().max().item() / scale) < 3e-6
    r3 = x.double().t() @ x.double()
    assert ((dw3.double() - r3).abs().max().item() / r3.abs().max().item()) < 3e-6
    torch.testing.assert_close(db2, db, rtol=1e-4, atol=1e-3 * dy.abs().max().item())


@pytest.mark.parametrize("B,Ci,H,W,Co", [(2, 256, 64, 48, 256), (1, 2048, 9, 7, 256), (2, 512, 33, 20, 256)])
def test_conv1x1_node_with_a_bf16_map_equals_the_float_cast_form(B, Ci, H, W, Co):
    """functions/conv_x3.Conv1x1OwnWgrad fed a bf16 channels-last backbone map (reference msdeformattn.py:324, 338: `features[f].float()` then
    the 1 x 1 convolution): pd_cast_bf16_f32_amax makes the fp32 copy and the row maxima in one pass, pd_gemm_tn_f16x2_bf16out returns the
    input gradient as bf16.  Against the same node on `x.float()` with autograd's casts: output bit-identical (same fp32 operands, same kernel), filter / bias
    gradients equal to round-off, the input gradient equal to the bf16 rounding of the fp32 one."""
    from partdistillation_amd.functions import conv_x3, gemm
    g = torch.Generator(device="cuda").manual_seed(B + Ci + H)
    xb = (torch.randn(B, Ci, H, W, device="cuda", generator=g) * 3).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, Ci, 1, 1, device="cuda", generator=g) * 0.05
    b = torch.randn(Co, device="cuda", generator=g)
    go = torch.randn(B, Co, H, W, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    # the fused cast: exact copy + exact maxima
    rows = xb.permute(0, 2, 3, 1).reshape(-1, Ci)
    x32, am = gemm.cast_rows_amax(rows)
    assert torch.equal(x32, rows.float()) and torch.equal(am, rows.float().abs().amax(1))
    outs = []
    for as_bf16 in (True, False):
        x = xb.clone().requires_grad_(True)
        wp, bp = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = conv_x3.conv1x1(x if as_bf16 else x.float(), wp, bp)
        y.backward(go)
        assert x.grad.dtype == torch.bfloat16
        outs.append((y.detach(), x.grad, wp.grad, bp.grad))
    (y1, dx1, dw1, db1), (y0, dx0, dw0, db0) = outs
    assert torch.equal(y1, y0)
    for a, c in ((dw1, dw0), (db1, db0)):                         # the filter gradient's partial tiles are combined in arrival order: round-off only
        assert float((a - c).abs().max()) <= 2e-6 * float(c.abs().max())
    assert torch.equal(dx1, dx0)                                  # the epilogue's round-to-nearest-even is torch's cast
