"""The fp32 GEMM kernels of the pixel decoder (fp16 two-plane form) at the encoder's shapes (BASELINE config 2: 43 008 tokens) and the
3 x 3 FPN convolution, a few launches each, for the rocprofv3 PMC passes of tools/pmc_gemm.sh (HBM bytes per launch next to the
algorithmic bytes).  Writes the sequence of bench.py kernel labels it issued to $PMC_LABELS (one per library call, in order)."""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from partdistillation_amd import lib; lib.load()
from partdistillation_amd.functions import gemm, conv_x3

T = 43008
torch.manual_seed(0)
x = {k: torch.randn(T, k, device="cuda") for k in (256, 1024, 288)}
am = {k: gemm.row_amax(v) for k, v in x.items()}
w = {(n, k): torch.randn(n, k, device="cuda") * k ** -0.5 for n, k in [(1024, 256), (256, 1024), (256, 256), (288, 256), (256, 288)]}
wam = {k: gemm.row_amax(v) for k, v in w.items()}
b1024, b256 = torch.randn(1024, device="cuda"), torch.randn(256, device="cuda")
col = torch.zeros(1024, device="cuda")
img = torch.randn(2, 256, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last)
wk = (torch.randn(256, 256, 3, 3, device="cuda") * 0.02).contiguous(memory_format=torch.channels_last)
gemm.enable_timing(True)
labels = []
for it in range(int(os.environ.get("ITERS", "3"))):
    h, bits = gemm.gemm_tn_h2(x[256], w[(1024, 256)], b1024, mode=1, want_bits=True, a_amax=am[256], b_amax=wam[(1024, 256)])   # linear1 + ReLU
    gemm.gemm_tn_h2(x[1024], w[(256, 1024)], b256, a_amax=am[1024], b_amax=wam[(256, 1024)])                                   # linear2
    gemm.gemm_tn_h2(x[256], w[(1024, 256)], None, mode=2, bits=bits, colsum=col, a_amax=am[256], b_amax=wam[(1024, 256)])      # d(hidden), masked
    gemm.gemm_tn_h2(x[256], w[(256, 256)], b256, a_amax=am[256], b_amax=wam[(256, 256)])                                       # a 256-wide projection
    gemm.gemm_tn_h2(x[256], w[(288, 256)], None, a_amax=am[256], b_amax=wam[(288, 256)])                                       # offsets + weights
    conv_x3._raw(img, wk.permute(0, 2, 3, 1).contiguous(), None, 256, conv_x3._pixel_amax(img))
    q = gemm.WgradQueue(h2=True)                                                  # the layer's five weight gradients, grouped
    for dy, xx in ((256, 1024), (1024, 256), (256, 256), (288, 256), (256, 256)):
        q.add(x[dy], x[xx], torch.zeros(dy, xx, device="cuda"), None, am[dy], am[xx])
    q.flush()
torch.cuda.synchronize()
fw = [f[1] for _, f in gemm.timing("fwd")]
json.dump({"fwd": fw, "wgrad_grouped_every": 6}, open(os.environ.get("PMC_LABELS", "/tmp/pmc_labels.json"), "w"))
