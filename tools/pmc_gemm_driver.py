"""The fp32 GEMM kernels of the pixel decoder (fp16 two-plane form) in the mix ONE encoder layer issues them at BASELINE config 2 (43 008
tokens; forward + backward: 10 forward / input-gradient GEMMs, the five weight gradients as the two grouped launches) and the 3 x 3 FPN
convolution, for the rocprofv3 PMC passes of tools/pmc_gemm.sh (HBM bytes per launch next to the algorithmic bytes).  Writes the
sequence of bench.py kernel labels it issued to $PMC_LABELS (one per library call, in order)."""
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
g = lambda a, n, k, bias=None, **kw: gemm.gemm_tn_h2(x[a], w[(n, k)], bias, a_amax=am[a], b_amax=wam[(n, k)], **kw)
for it in range(int(os.environ.get("ITERS", "3"))):
    g(256, 256, 256, b256); g(256, 288, 256); g(256, 256, 256, b256)              # forward: value, offsets + weights, output projection
    h, bits = g(256, 1024, 256, b1024, mode=1, want_bits=True)                     # linear1 + ReLU (+ sign bits)
    g(1024, 256, 1024, b256)                                                       # linear2
    g(256, 1024, 256, mode=2, bits=bits, colsum=col)                               # backward: d(hidden), masked
    g(1024, 256, 1024)                                                             # d(FFN input)
    g(256, 256, 256); g(288, 256, 288); g(256, 256, 256)                           # d(attention output), d(query), d(value input)
    conv_x3._raw(img, wk.permute(0, 2, 3, 1).contiguous(), None, 256, conv_x3._pixel_amax(img))
    q = gemm.WgradQueue(h2=True)                                                   # the five weight gradients of SIX layers, as the step queues
    for layer in range(6):                                                         # them: two grouped launches (wide / narrow tiles) + their reduces
        for dy, xx in ((256, 1024), (1024, 256), (256, 256), (288, 256), (256, 256)):
            q.add(x[dy], x[xx], torch.zeros(dy, xx, device="cuda"), None, am[dy], am[xx])
    q.flush()
torch.cuda.synchronize()
json.dump({"fwd": [f[1] for _, f in gemm.timing("fwd")]}, open(os.environ.get("PMC_LABELS", "/tmp/pmc_labels.json"), "w"))
