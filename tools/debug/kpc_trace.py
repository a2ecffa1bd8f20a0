"""gemm_kpc_f16x2 with wall-clock stamps (pd_debug_set("f16x2_tile", 232)): what the first consumer and the first producer wavefront of workgroup 0
spend a chunk on.  Cycles of the shader clock counter (s_memtime), averaged over chunks 4 .. 27.  GPU box: python tools/debug/kpc_trace.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from partdistillation_amd import lib; L = lib.load()
from partdistillation_amd.functions import gemm
M, N, K = 43008, 256, 1024
if len(sys.argv) > 1 and sys.argv[1] == "conv":                  # the 3 x 3 convolution form (2 x 256^2 x 256): first 32 of its 72 chunks
    from partdistillation_amd.functions import conv_x3
    x = torch.randn(2, 256, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last)
    wk = (torch.randn(256, 256, 3, 3, device="cuda") * 0.02).permute(0, 2, 3, 1).contiguous()
    am = conv_x3._pixel_amax(x)
    L.pd_debug_set(b"f16x2_tile", 234)
    for _ in range(3):
        conv_x3._raw(x, wk, None, 256, am)
else:
    a, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * K ** -0.5
    aa, wa = gemm.row_amax(a), gemm.row_amax(w)
    L.pd_debug_set(b"f16x2_tile", int(sys.argv[1]) if len(sys.argv) > 1 else 232)
    for _ in range(3):
        gemm.gemm_tn_h2(a, w, None, a_amax=aa, b_amax=wa)
torch.cuda.synchronize()
L.pd_debug_set(b"f16x2_tile", 0)
buf = (ctypes.c_ulonglong * (2 * 64 * 8))()
L.pd_debug_read_kpc_trace.argtypes = [ctypes.c_void_p]
assert L.pd_debug_read_kpc_trace(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(2, 64, 8).astype(np.int64)
c, p = t[0, :32], t[1, :32]
sl = slice(4, 28)
print("whole K loop (first stamp of chunk 0 -> last of chunk 31): consumer", int(c[31, 2] - c[0, 0]), "producer", int(p[31, 5] - p[0, 0]), "cycles")
print("consumer per chunk: wait for `full` %.0f | fragments of step 0 in %.0f | step 0 products issued + fragments of step 1 in %.0f | step 1 products issued %.0f | to next chunk %.0f" % (
    (c[sl, 1] - c[sl, 0]).mean(), (c[sl, 3] - c[sl, 1]).mean(), (c[sl, 4] - c[sl, 3]).mean(), (c[sl, 2] - c[sl, 4]).mean(), (c[5:29, 0] - c[sl, 2]).mean()))
print("producer per chunk: wait for `empty` %.0f | loads of half 0 landed %.0f | split + stores of half 0 %.0f | loads of half 1 landed %.0f | split + stores of half 1 %.0f | to next chunk %.0f" % (
    (p[sl, 1] - p[sl, 0]).mean(), (p[sl, 2] - p[sl, 1]).mean(), (p[sl, 3] - p[sl, 2]).mean(), (p[sl, 4] - p[sl, 3]).mean(), (p[sl, 5] - p[sl, 4]).mean(), (p[5:29, 0] - p[sl, 5]).mean()))
print("chunk period: consumer %.0f, producer %.0f cycles" % (np.diff(c[4:29, 0]).mean(), np.diff(p[4:29, 0]).mean()))
