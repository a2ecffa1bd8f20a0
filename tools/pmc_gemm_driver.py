"""The fp32-on-bf16 GEMM kernels at the encoder's shapes (BASELINE config 2: 43 008 tokens), a few launches each, for the rocprofv3
PMC passes of tools/pmc_gemm.sh (HBM bytes per launch next to the algorithmic bytes)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from partdistillation_amd import lib; lib.load()
from partdistillation_amd.functions import gemm

T = 43008
torch.manual_seed(0)
x256, x1024, x288 = (torch.randn(T, k, device="cuda") for k in (256, 1024, 288))
w = {(n, k): torch.randn(n, k, device="cuda") * k ** -0.5 for n, k in [(1024, 256), (256, 1024), (256, 256), (288, 256), (256, 288)]}
b1024, b256 = torch.randn(1024, device="cuda"), torch.randn(256, device="cuda")
col = torch.zeros(1024, device="cuda")
for it in range(int(os.environ.get("ITERS", "4"))):
    h, bits = gemm.gemm_tn_x3_relu_bits(x256, w[(1024, 256)], b1024)             # linear1 + ReLU (+ sign bits)
    gemm.gemm_tn_x3(x1024, w[(256, 1024)], b256)                                   # linear2
    gemm.gemm_tn_x3_relumask(x256, w[(1024, 256)], bits, col)                      # d(hidden) with the ReLU mask
    gemm.gemm_tn_x3(x256, w[(256, 256)], b256)                                     # a 256-wide projection
    gemm.gemm_tn_x3(x256, w[(288, 256)], None)                                     # offsets + weights projection
    for dy, x in ((x256, x1024), (x1024, x256), (x256, x256), (x288, x256)):      # the layer's weight gradients
        dw = torch.zeros(dy.shape[1], x.shape[1], device="cuda")
        gemm.gemm_wgrad_acc(dy, x, dw)
torch.cuda.synchronize()
