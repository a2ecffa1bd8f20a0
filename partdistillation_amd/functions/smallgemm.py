"""ctypes faces of the skinny-activation bf16 GEMMs (include/pd_smallgemm.h).  Raw functions, no autograd: building
blocks of the hand-written decoder backward (functions/decoder_core.py).  GPU only; there is no fallback."""
import torch

from .. import lib as _lib


def _stream():
    return _lib.current_stream()


def _chk2d(*ts):
    for t in ts:
        assert t.dtype == torch.bfloat16 and t.dim() == 2 and t.stride(1) == 1, (t.dtype, t.shape, t.stride())


def supported(K_tn=None, N_nn=None):
    return (K_tn is None or K_tn % 64 == 0) and (N_nn is None or N_nn % 64 == 0)


def linear(x, w, b=None, relu=False, out=None):
    """x [M,K] @ w[N,K].T (+ b) (ReLU) -> [M,N]   (bf16; K % 64 == 0)"""
    _chk2d(x, w)
    M, K = x.shape
    N = w.shape[0]
    y = out if out is not None else torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.load().pd_sgemm_tn_bf16(x.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(),
                                            M, N, K, x.stride(0), w.stride(0), y.stride(0), int(relu), _stream()))
    return y


def dgrad(dy, w, relu_ref=None, out=None, accumulate=False):
    """dy [M,N] @ w [N,K] -> [M,K], optionally accumulated into `out` and masked by relu_ref > 0   (N % 64 == 0)"""
    _chk2d(dy, w)
    M, N = dy.shape
    K = w.shape[1]
    dx = out if out is not None else torch.empty((M, K), dtype=torch.bfloat16, device=dy.device)
    assert relu_ref is None or (relu_ref.shape == dx.shape and relu_ref.stride() == dx.stride())
    _lib.check(_lib.load().pd_sgemm_nn_bf16(dy.data_ptr(), w.data_ptr(), relu_ref.data_ptr() if relu_ref is not None else None,
                                            dx.data_ptr(), M, N, K, dy.stride(0), w.stride(0), dx.stride(0), int(accumulate), _stream()))
    return dx


def wgrad(dy, x, out=None, bias_out=None):
    """dy [M,N].T @ x [M,K] -> [N,K] (bf16);  bias_out (fp32 [N], optional) = dy.sum(0)"""
    _chk2d(dy, x)
    M, N = dy.shape
    K = x.shape[1]
    dw = out if out is not None else torch.empty((N, K), dtype=torch.bfloat16, device=dy.device)
    assert dw.stride(1) == 1 and (bias_out is None or (bias_out.dtype == torch.float32 and bias_out.numel() == N))
    _lib.check(_lib.load().pd_sgemm_wgrad_bf16(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), bias_out.data_ptr() if bias_out is not None else None,
                                               M, N, K, dy.stride(0), x.stride(0), dw.stride(0), _stream()))
    return dw


_WGRAD_WS = {}


def wgrad_split(dy, x, want_bias=True):
    """dy [M,N].T @ x [M,K] for many rows (pd_sgemm_wgrad_split_bf16) -> (dW bf16 [N,K], dB fp32 [N] or None)"""
    _chk2d(dy, x)
    M, N = dy.shape
    K = x.shape[1]
    L = _lib.load()
    need = int(L.pd_sgemm_wgrad_split_workspace(M, N, K))
    key = str(dy.device)
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _WGRAD_WS[key] = torch.empty(need, dtype=torch.float32, device=dy.device)
    dw = torch.empty((N, K), dtype=torch.bfloat16, device=dy.device)
    db = torch.empty(N, dtype=torch.float32, device=dy.device) if want_bias else None
    _lib.check(L.pd_sgemm_wgrad_split_bf16(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr() if want_bias else None, ws.data_ptr(),
                                           M, N, K, dy.stride(0), x.stride(0), dw.stride(0), _stream()))
    return dw, db

