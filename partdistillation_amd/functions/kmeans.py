"""K-means (Lloyd) on the device with the semantics of sklearn.cluster.KMeans(algorithm="lloyd") — the clustering the
reference runs on the CPU for every image (proposal_generation_model.py:202-211: `KMeans(n_clusters=K, random_state=0)`
on the object's feature vectors; scikit-learn is the pinned third-party implementation, 1.7.x here).

Followed from sklearn/cluster/_kmeans.py: the data is centred, tol is scaled by the mean per-feature variance, an
iteration assigns every point to argmin_k |c_k|^2 - 2 x.c_k (first minimum) and moves the centres to the cluster
means; it stops when the labels repeat (strict convergence) or when the squared centre shift <= tol, in which case the
assignment is recomputed once for the final centres.  Differences: the k-means++ seeding uses the device generator
(numpy's RandomState stream is not reproduced — parity tests inject the initial centres), and empty clusters keep their
previous centre instead of sklearn's relocation heuristic.  The two GEMM-shaped steps (point x centre scores, one-hot x
points) are library GEMMs; the whole loop stays on the device with one scalar read-back per iteration."""
import math

import torch


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
