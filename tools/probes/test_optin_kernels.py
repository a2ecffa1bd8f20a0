"""Parity tests of the kernels that lost their same-box A/B and live outside the product library (tools/probes/gemm_f16x2_optin.h,
the CONV / pre-split-plane forms of gemm_kpc_f16x2): build the probe library and point PD_LIB_PATH at it,

    make -C partdistillation_amd/csrc probes
    PD_LIB_PATH=$PWD/partdistillation_amd/libpd_hip_probes.so python -m pytest tools/probes/test_optin_kernels.py -q

(moved out of tests/test_gemm_gpu.py in round 6; DESIGN.md keeps their measurements)."""
import os

import pytest
import torch

if not torch.cuda.is_available() or "probes" not in os.environ.get("PD_LIB_PATH", ""):
    pytest.skip("needs a GPU and PD_LIB_PATH = the probe library", allow_module_level=True)


def test_conv3x3_on_producer_consumer_wavefronts_equals_the_tiled_kernel():
    """gemm_kpc_f16x2<.., CONV> (opt-in, pd_debug_set("f16x2_tile", 92)): the 3 x 3 convolution's implicit GEMM with the tap shift added to the
    rows' buffer offsets and out-of-image taps read as zeros — against the tiled kernel on the same operands (same split, same products, another
    summation order) and against fp64; an image size whose last row block is ragged."""
    import torch.nn.functional as F
    from partdistillation_amd import lib
    from partdistillation_amd.functions import conv_x3
    L = lib.load()
    torch.manual_seed(3)
    x = torch.randn(2, 256, 70, 61, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(256, 256, 3, 3, device="cuda") * 0.02
    b = torch.randn(256, device="cuda")
    wk = w.permute(0, 2, 3, 1).contiguous()
    am = conv_x3._pixel_amax(x)
    tiled = conv_x3._raw(x, wk, b, 256, am)
    L.pd_debug_set(b"f16x2_tile", 92)
    try:
        got = conv_x3._raw(x, wk, b, 256, am)
    finally:
        L.pd_debug_set(b"f16x2_tile", 0)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    scale = ref.abs().amax(1, keepdim=True)
    assert ((got.double() - ref).abs() / scale).max().item() < 5e-6
    assert ((got.double() - tiled.double()).abs() / scale).max().item() < 5e-6        # (another fp32 summation order over K = 2 304: 2e-6 measured)




@pytest.mark.parametrize("M,N,K", [(43008, 256, 1024), (8300, 512, 512), (9001, 256, 576), (8192, 256, 2048)])
@pytest.mark.parametrize("scaled", [True, False])
@pytest.mark.parametrize("variant", [91, 92, 93])       # 91: barriers per chunk (gemm_kres_f16x2); 92 / 93: producer / consumer wavefronts (gemm_kpc_f16x2) with fp32 weights / pre-split weight planes
def test_gemm_tn_h2_resident_accumulators_match_fp64_and_the_tiled_kernel(M, N, K, scaled, variant):
    """gemm_kres_f16x2 (deep K, 256-column panels: K outside, a 192-row block's accumulators resident; experimental,
    pd_debug_set("f16x2_tile", 91)): fp32-accurate against fp64, within 1e-6 of the tiled kernel, ragged last row block, two column panels,
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
        assert variant != 91 or L.pd_gemm_tn_f16x2_which(M, N, K, 0, 0, int(scaled)) == 4
        got = gemm.gemm_tn_h2(a, w, b, a_amax=aa, b_amax=wa)
        again = gemm.gemm_tn_h2(a, w, b, a_amax=aa, b_amax=wa)
    finally:
        L.pd_debug_set(b"f16x2_tile", 0)
    assert torch.equal(got, again)
    assert ((got.double() - ref).abs() / rown).max().item() < 3e-6
    assert ((got.double() - tiled.double()).abs() / rown).max().item() < 1e-6




@pytest.mark.parametrize("M,N,K", [(8300, 256, 256), (9001, 288, 256), (13000, 1024, 256)])
@pytest.mark.parametrize("scaled", [True, False])
def test_gemm_tn_h2_register_operand_kernel_matches_the_tiled_kernel(M, N, K, scaled):
    """gemm_ra_f16x2_k256 (round 5, experimental: A rows global -> registers -> matrix cores, the weight block resident in LDS;
    pd_debug_set("f16x2_tile", 90)): fp32-accurate against fp64, exact row maxima, ragged last row block, 128- and 96-column blocks."""
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
    lib.load().pd_debug_set(b"f16x2_tile", 90)
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


