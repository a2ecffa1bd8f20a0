"""K-means (Lloyd) on the device with the semantics of sklearn.cluster.KMeans(algorithm="lloyd") — the clustering the
reference runs on the CPU for every image (proposal_generation_model.py:202-211: `KMeans(n_clusters=K, random_state=0)`
on the object's feature vectors; scikit-learn is the pinned third-party implementation, 1.7.x here).

Followed from sklearn/cluster/_kmeans.py: the data is centred, tol is scaled by the mean per-feature variance, an
iteration assigns every point to argmin_k |c_k|^2 - 2 x.c_k (first minimum) and moves the centres to the cluster
means; it stops when the labels repeat (strict convergence) or when the squared centre shift <= tol, in which case the
assignment is recomputed once for the final centres.  Differences: the k-means++ seeding uses the device generator
(numpy's RandomState stream is not reproduced — parity tests inject the initial centres), and empty clusters keep their
previous centre instead of sklearn's relocation heuristic.

`kmeans_lloyd_batched` is the product path: all images of a batch advance together through pd_kmeans_assign /
pd_kmeans_update (include/pd_kmeans.h: one wavefront per point, convergence decided on the device, the host looks at the
`done` flags every few iterations).  `kmeans_lloyd` is the same algorithm in library calls with a read-back per
iteration (more than 4 clusters or more than 2048 channels)."""
import math

import torch

from .. import lib as _lib


def kmeans_plusplus(X, K, generator=None):
    """sklearn _kmeans_plusplus on centred data X [N,C] (fp32, device) -> initial centres [K,C]."""
    N = X.shape[0]
    trials = 2 + int(math.log(K))
    first = torch.randint(N, (1,), device=X.device, generator=generator)
    centers = [X[first[0]]]
    xsq = (X * X).sum(1)
    closest = (xsq - 2.0 * (X @ centers[0]) + (centers[0] * centers[0]).sum()).clamp_min_(0)
    for _ in range(1, K):
        pot = closest.sum()
        r = torch.rand(trials, device=X.device, generator=generator) * pot
        cand = torch.searchsorted(closest.cumsum(0), r).clamp_(max=N - 1)
        Xc = X[cand]                                                                       # [T,C]
        d = (xsq[None, :] - 2.0 * (Xc @ X.t()) + (Xc * Xc).sum(1)[:, None]).clamp_min_(0)  # [T,N]
        d = torch.minimum(d, closest[None, :])
        best = d.sum(1).argmin()
        closest = d[best]
        centers.append(Xc[best])
    return torch.stack(centers)


def kmeans_plusplus_batched(Xs, K, generator=None):
    """the same seeding for B images at once (Xs: list of centred [N_b, C] fp32 tensors) -> [B, K, C]: every step of kmeans_plusplus as one
    batched operator over a zero-padded [B, N_max, C] stack — the per-image loop issued ~40 small launches per image, 2.5 ms of the 17 ms a
    batch of four 1024^2 images takes (config 4).  The device generator's stream is consumed in another order than by the loop; parity
    tests inject the initial centres."""
    B, dev = len(Xs), Xs[0].device
    n = torch.tensor([x.shape[0] for x in Xs], dtype=torch.long).to(dev, non_blocking=True) if any(x.shape[0] != Xs[0].shape[0] for x in Xs) else None
    Nmax = max(x.shape[0] for x in Xs)
    if n is None:
        X = torch.stack(Xs)                                                                # [B, N, C]
        valid = None
        nb = torch.full((B,), Nmax, dtype=torch.long, device=dev)
    else:
        X = torch.zeros((B, Nmax, Xs[0].shape[1]), dtype=torch.float32, device=dev)
        for b, x in enumerate(Xs):
            X[b, :x.shape[0]] = x
        nb = n
        valid = torch.arange(Nmax, device=dev)[None, :] < nb[:, None]
    trials = 2 + int(math.log(K))
    ar = torch.arange(B, device=dev)
    first = (torch.rand(B, device=dev, generator=generator) * nb).long().clamp_(max=Nmax - 1)
    first = torch.minimum(first, nb - 1)
    c0 = X[ar, first]                                                                      # [B, C]
    centers = [c0]
    xsq = (X * X).sum(2)                                                                   # [B, N]
    closest = (xsq - 2.0 * torch.bmm(X, c0[:, :, None])[:, :, 0] + (c0 * c0).sum(1, keepdim=True)).clamp_min_(0)
    if valid is not None:
        closest = closest * valid
    for _ in range(1, K):
        pot = closest.sum(1)                                                               # [B]
        r = torch.rand((B, trials), device=dev, generator=generator) * pot[:, None]
        cand = torch.searchsorted(closest.cumsum(1), r)
        cand = torch.minimum(cand, (nb - 1)[:, None])                                      # [B, T]
        Xc = X[ar[:, None], cand]                                                          # [B, T, C]
        d = (xsq[:, None, :] - 2.0 * torch.bmm(Xc, X.transpose(1, 2)) + (Xc * Xc).sum(2)[:, :, None]).clamp_min_(0)   # [B, T, N]
        d = torch.minimum(d, closest[:, None, :])
        if valid is not None:
            d = d * valid[:, None, :]
        best = d.sum(2).argmin(1)                                                          # [B]
        closest = d[ar, best]
        centers.append(Xc[ar, best])
    return torch.stack(centers, 1)                                                         # [B, K, C]


def kmeans_lloyd(X, K, init=None, max_iter=300, tol=1e-4, generator=None):
    """X [N,C] fp32 on the device -> (centres [K,C], labels [N] int64, iterations).  `init` [K,C] (in the coordinates of
    X) replaces the k-means++ seeding."""
    X = X.float()
    mean = X.mean(0)
    Xc = X - mean
    scaled_tol = Xc.var(0, unbiased=False).mean() * tol
    centers = (init.float() - mean) if init is not None else kmeans_plusplus(Xc, K, generator)
    labels_old = torch.full((X.shape[0],), -1, dtype=torch.long, device=X.device)
    strict, it = False, 0
    ones = torch.ones((X.shape[0],), dtype=torch.float32, device=X.device)
    for it in range(1, max_iter + 1):
        labels = ((centers * centers).sum(1)[None, :] - 2.0 * (Xc @ centers.t())).argmin(1)
        onehot = torch.zeros((K, X.shape[0]), dtype=torch.float32, device=X.device).scatter_(0, labels[None, :], ones[None, :])
        counts = onehot.sum(1)
        new = torch.where(counts[:, None] > 0, (onehot @ Xc) / counts.clamp_min(1)[:, None], centers)
        same = torch.equal(labels, labels_old)                                 # host read-back (one per iteration)
        shift = ((new - centers) ** 2).sum()
        centers = new
        if same:
            strict = True
            break
        if bool(shift <= scaled_tol):
            break
        labels_old = labels
    if not strict:
        labels = ((centers * centers).sum(1)[None, :] - 2.0 * (Xc @ centers.t())).argmin(1)
    return centers + mean, labels, it


SEED_LOOP = bool(int(__import__("os").environ.get("PD_KMEANS_SEED_LOOP", "0")))   # tools only: k-means++ seeding image by image (the first version)
SLAB = int(__import__("os").environ.get("PD_KMEANS_SLAB", "32"))
ATOMIC = bool(int(__import__("os").environ.get("PD_KMEANS_ATOMIC", "0")))     # points per workgroup of pd_kmeans_assign (<= 64)
TRACE = bool(int(__import__("os").environ.get("PD_KMEANS_TRACE", "0")))
BOUNDED = bool(int(__import__("os").environ.get("PD_KMEANS_BOUNDED", "1")))   # E-step with distance bounds (pd_kmeans_assign_bounded: exact, same results);
                                                                               # 0: every point against every centre in every iteration


def kmeans_lloyd_batched(datas, K, inits=None, max_iter=300, tol=1e-4, generator=None, check_every=8):
    """datas: list of [N_b, C] fp32 CUDA tensors (N_b > K) -> (centres [B,K,C], iterations list).  HIP kernels; K <= 8
    (instantiated for 4 = pixel grouping and 8 = part ranking), C % 4 == 0, C <= 2048 (else falls back to kmeans_lloyd per
    image)."""
    B, C = len(datas), datas[0].shape[1]
    dev = datas[0].device
    if not dev.type == "cuda":
        raise RuntimeError("pd_kmeans_* run on the GPU only (no CPU fallback in partdistillation_amd)")
    if K > 8 or C % 4 or C > 2048:
        out = [kmeans_lloyd(d, K, init=None if inits is None else inits[b], max_iter=max_iter, tol=tol, generator=generator)
               for b, d in enumerate(datas)]
        return torch.stack([o[0] for o in out]), [o[2] for o in out]
    means = [d.float().mean(0) for d in datas]
    Xs = [d.float() - m for d, m in zip(datas, means)]
    tols = torch.stack([x.var(0, unbiased=False).mean() * tol for x in Xs]).float().contiguous()
    need = [b for b in range(B) if inits is None or inits[b] is None]                       # images without injected initial centres: seeded together
    if SEED_LOOP:
        seeded = {b: kmeans_plusplus(Xs[b], K, generator) for b in need}
    else:
        seeded = dict(zip(need, kmeans_plusplus_batched([Xs[b] for b in need], K, generator))) if need else {}
    centers = torch.stack([seeded[b] if b in seeded else (inits[b].float() - means[b]) for b in range(B)]).contiguous()   # [B,K,C]
    X = torch.cat(Xs).contiguous()
    table, ranges, off = [], [], 0
    for b, x in enumerate(Xs):
        ranges += [len(table), -(-x.shape[0] // SLAB)]                                       # (first block, number of blocks) of image b
        for s0 in range(0, x.shape[0], SLAB):
            table.append((b, off + s0, min(SLAB, x.shape[0] - s0)))
        off += x.shape[0]
    from .fused import upload_small
    blocks = upload_small([v for row in table for v in row], torch.int32, dev).view(-1, 3)   # pinned + async: no stall behind the backbone
    block_range = upload_small(ranges, torch.int32, dev)
    labels = torch.full((X.shape[0],), -1, dtype=torch.int32, device=dev)
    sums = torch.zeros((B, K, C), dtype=torch.float32, device=dev)
    counts = torch.zeros((B, K), dtype=torch.float32, device=dev)
    psums = torch.empty((len(table), K, C), dtype=torch.float32, device=dev)                 # per-slab partial sums (no atomics)
    pcounts = torch.empty((len(table), K), dtype=torch.float32, device=dev)
    flags = torch.zeros((4, B), dtype=torch.int32, device=dev)                               # changed, done, n_iter, ticket
    scratch = torch.empty(int(_lib.load().pd_kmeans_reduce_update_scratch_floats(B, K, C)), dtype=torch.float32, device=dev)
    cnorm = (centers * centers).sum(-1).contiguous()
    bounds = torch.empty((3, X.shape[0]), dtype=torch.float32, device=dev)                   # ub, lb, |x|^2 (read only once a point has a label)
    cshift = torch.zeros((B, 2, 8), dtype=torch.float32, device=dev)                         # centre moves of the last update (bounded E-step)
    lib, st = _lib.load(), _lib.current_stream()
    p = dict(X=X.data_ptr(), blocks=blocks.data_ptr(), centers=centers.data_ptr(), cnorm=cnorm.data_ptr(), done=flags[1].data_ptr(),
             labels=labels.data_ptr(), psums=psums.data_ptr(), pcounts=pcounts.data_ptr(), changed=flags[0].data_ptr(),
             range=block_range.data_ptr(), sums=sums.data_ptr(), counts=counts.data_ptr(), tols=tols.data_ptr(), n_iter=flags[2].data_ptr(),
             scratch=scratch.data_ptr(), ticket=flags[3].data_ptr())
    # convergence is decided on the device; the host only has to STOP issuing.  It looks at the `done` flags through non-blocking copies to
    # pinned memory and keeps issuing while a copy is in flight: a blocking read every `check_every` iterations left the GPU idle while the
    # host caught up (8 ms of wall time for 4.7 ms of kernels at config 4).  Iterations issued after every image is done are ~4 us no-ops.
    it = 0
    pending = []                                                                             # (event, pinned snapshot of the done flags)
    finished = False
    while it < max_iter and not finished:
        for _ in range(min(check_every, max_iter - it)):
            if ATOMIC:                                                                       # tools only: the first version's accumulation
                _lib.check(lib.pd_kmeans_assign(p["X"], p["blocks"], len(table), p["centers"], p["cnorm"], p["done"], p["labels"],
                                                p["sums"], p["counts"], p["changed"], C, K, st))
            elif BOUNDED:
                _lib.check(lib.pd_kmeans_assign_bounded(p["X"], p["blocks"], len(table), p["centers"], p["cnorm"], p["done"], p["labels"],
                                                        p["psums"], p["pcounts"], p["changed"], bounds[0].data_ptr(), bounds[1].data_ptr(),
                                                        bounds[2].data_ptr(), cshift.data_ptr(), C, K, st))
                _lib.check(lib.pd_kmeans_reduce_update_shift(p["psums"], p["pcounts"], p["range"], p["centers"], p["cnorm"], p["changed"], p["tols"],
                                                             p["done"], p["n_iter"], p["scratch"], p["ticket"], cshift.data_ptr(), B, K, C, st))
            else:
                _lib.check(lib.pd_kmeans_assign_partial(p["X"], p["blocks"], len(table), p["centers"], p["cnorm"], p["done"], p["labels"],
                                                        p["psums"], p["pcounts"], p["changed"], C, K, st))
                _lib.check(lib.pd_kmeans_reduce_update(p["psums"], p["pcounts"], p["range"], p["centers"], p["cnorm"], p["changed"], p["tols"],
                                                       p["done"], p["n_iter"], p["scratch"], p["ticket"], B, K, C, st))
            if ATOMIC:
                _lib.check(lib.pd_kmeans_update(p["centers"], p["cnorm"], p["sums"], p["counts"], p["changed"], p["tols"], p["done"],
                                                p["n_iter"], B, K, C, st))
            it += 1
            if TRACE and BOUNDED:                                                            # tools only: which share of the points the NEXT E-step will read
                img = torch.repeat_interleave(torch.arange(B, device=dev), torch.tensor([x.shape[0] for x in Xs], device=dev))
                a = labels.long().clamp_min(0)
                u = (bounds[0] + cshift[img, 0, a]) * 1.000001
                l = (bounds[1] - cshift[img, 1, a]) * 0.999999
                skip = (l > u) & (l * l - u * u > 2e-3 * (bounds[2] + cnorm.max(1).values[img])) & ~flags[1].bool()[img]
                live = ~flags[1].bool()[img]
                print(f"iteration {it}: done {flags[1].tolist()}, points to read {int((live & ~skip).sum())} of {int(live.sum())} live, "
                      f"centre moves {[round(float(v), 4) for v in cshift[:, 0, :K].max(1).values]}")
        snap = torch.empty(B, dtype=torch.int32, pin_memory=True)
        snap.copy_(flags[1], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        pending.append((ev, snap))
        if len(pending) > 2:                                                                 # at most two chunks ahead of what the host has seen
            pending[0][0].synchronize()
        while pending and pending[0][0].query():
            if bool(pending.pop(0)[1].all()):
                finished = True
                break
    return centers + torch.stack(means)[:, None, :], flags[2].tolist()
