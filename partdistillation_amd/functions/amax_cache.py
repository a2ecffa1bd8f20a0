"""Row maxima travel with their tensors: a kernel that writes an fp32 map / token matrix and has its rows in registers anyway
(GroupNorm apply, upsample + add, LayerNorm, GEMM epilogues) notes the absolute row maxima here, and the fp16 two-plane GEMMs /
convolutions that read the tensor later (functions/gemm.gemm_tn_h2, functions/conv_x3) take them instead of a separate
pd_row_amax_f32 pass.  An entry is only handed out for the SAME tensor object at the SAME version (weak reference + version
counter): a freed and re-used address or an in-place update misses and the consumer computes the maxima itself.

CONTRACT for kernels that write through raw data_ptr() (they do not bump `_version`): a kernel that rewrites a tensor IN PLACE
after its maxima were put() here must call invalidate(t) (or put() the new maxima).  Stale, too-small maxima would scale rows
past the fp16 range inside the two-plane GEMMs (Inf / NaN, silently).  PD_AMAX_CHECK=1 makes get() recompute the maxima with
torch and raise on a stale entry (debugging aid; one reduction + a host sync per lookup)."""
import os
import weakref

_CHECK = os.environ.get("PD_AMAX_CHECK", "0") != "0"

_CACHE = {}
_MAX = 64


def put(t, amax):
    if len(_CACHE) >= _MAX:
        for k in [k for k, (r, _, _) in _CACHE.items() if r() is None]:
            del _CACHE[k]
        if len(_CACHE) >= _MAX:
            _CACHE.clear()
    _CACHE[t.data_ptr()] = (weakref.ref(t), t._version, amax)


def invalidate(t):
    """forget the maxima of `t` (call after rewriting it in place through a raw pointer)"""
    _CACHE.pop(t.data_ptr(), None)


def get(t):
    e = _CACHE.get(t.data_ptr())
    if e is not None and e[0]() is t and e[1] == t._version and e[2].numel() * t.shape[1 if t.dim() == 4 else -1] == t.numel():
        if _CHECK:
            rows = t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]) if t.dim() == 4 else t.reshape(-1, t.shape[-1])
            true = rows.abs().amax(1).reshape(-1)
            if not bool((e[2].reshape(-1) >= true * (1 - 1e-6)).all()):
                raise RuntimeError("amax_cache: stale row maxima (the tensor was rewritten in place without invalidate())")
        return e[2]
    return None
