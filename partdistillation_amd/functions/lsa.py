"""Device-side Hungarian assignment (pd_lsa_batched, include/pd_criterion.h)."""
import torch

from .. import lib as _lib


def solve_batched(cost, ncols):
    """cost float32 [NB, R, CMAX] (cuda), ncols int32 [NB] -> (rows, cols) int64 [NB, CMAX], -1 padded,
    pairs ordered by ascending cost.  No host synchronisation."""
    if not cost.is_cuda:
        raise RuntimeError("pd_lsa_batched runs on the GPU only (no CPU fallback in partdistillation_amd)")
    cost = cost.contiguous().float()
    ncols = ncols.to(device=cost.device, dtype=torch.int32).contiguous()
    nb, r, cmax = cost.shape
    rows = torch.empty((nb, cmax), dtype=torch.int64, device=cost.device)
    cols = torch.empty((nb, cmax), dtype=torch.int64, device=cost.device)
    if nb * cmax == 0:
        return rows, cols
    with torch.cuda.device(cost.device):
        rc = _lib.load().pd_lsa_batched(cost.data_ptr(), ncols.data_ptr(), rows.data_ptr(), cols.data_ptr(), nb, r, cmax,
                                        _lib.current_stream())
    _lib.check(rc)
    return rows, cols


def solve_ragged(costs):
    """list of [Q, n_b] cost matrices -> list of (rows[n_b'], cols[n_b']) like the reference matcher returns
    (n_b' = min(Q, n_b)); the slicing uses host-known sizes only, so it does not synchronise."""
    if len(costs) == 0:
        return []
    q = costs[0].shape[0]
    cmax = max(max(c.shape[1] for c in costs), 1)
    batch = costs[0].new_zeros((len(costs), q, cmax), dtype=torch.float32)
    for b, c in enumerate(costs):
        batch[b, :, : c.shape[1]] = c
    ncols = torch.tensor([c.shape[1] for c in costs], dtype=torch.int32)
    rows, cols = solve_batched(batch, ncols)
    out = []
    for b, c in enumerate(costs):
        k = min(q, c.shape[1])
        out.append((rows[b, :k], cols[b, :k]))
    return out
