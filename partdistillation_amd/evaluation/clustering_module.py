"""``ClusteringModule`` (reference part_distillation/evaluation/clustering_module.py:17-80): collects the per-proposal
features PartRankingModel returns in mode "cluster", and turns them into K-means centroids per object class.

The reference moves every feature to the CPU, gathers them on rank 0 and runs sklearn KMeans(n_clusters, random_state=0)
per class there.  Here the features stay on the device and each class is clustered by functions/kmeans.py (sklearn's Lloyd
semantics: centred data, variance-scaled tolerance, first-minimum assignment; k-means++ seeding from a seeded device
generator — numpy's RandomState stream is not reproduced, `init` injects centres for parity tests).  With several ranks
the (feature, label) lists are exchanged with all_gather_object, rank 0 clusters (classes with too few proposals get
torch.randn centroids from ITS generator, reference :66-67) and its dictionary is broadcast — every rank ends up with
rank 0's classifier, as with the reference's `all_gather(cluster_centroids_dict)[0]` (:72-73)."""
import copy

import torch

from ..functions import kmeans as _kmeans


class ClusteringModule:
    def __init__(self, distributed=True, num_clusters=8, seed=0):
        self._distributed, self.num_clusters, self.seed = distributed, num_clusters, seed
        self.init = None                         # test hook: callable(class id, features) -> [num_clusters, C] initial centres
        self.reset()

    def reset(self):
        self._proposal_features, self._class_labels_list = [], []

    def process(self, inputs, outputs):
        for out in outputs:
            self._proposal_features.append(out["proposal_features"])
            self._class_labels_list.append(out["gt_label"])

    def evaluate(self):
        feats, labels = self._proposal_features, self._class_labels_list
        dist = torch.distributed
        multi = self._distributed and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        dev = feats[0].device if feats else torch.device("cpu")
        if multi:
            gathered = [None] * dist.get_world_size()
            dist.all_gather_object(gathered, ([f.cpu() for f in feats], [l.cpu() for l in labels]))
            feats = [f.to(dev) for g in gathered for f in g[0]]
            labels = [l.to(dev) for g in gathered for l in g[1]]
        out = {}
        if not multi or dist.get_rank() == 0:                     # the main process clusters (reference :58-67)
            feats = torch.cat(feats, dim=0).float()
            labels = torch.cat([l.to(feats.device) for l in labels], dim=0)
            todo = []
            for cid in labels.unique().tolist():
                x = feats[labels == cid]
                if x.shape[0] > self.num_clusters:
                    todo.append((int(cid), x))
                else:                                 # too few proposals of this class (reference :66-67)
                    out[int(cid)] = torch.randn(self.num_clusters, x.shape[1], device=x.device)
            if todo and todo[0][1].is_cuda and self.num_clusters <= 8 and todo[0][1].shape[1] % 4 == 0:
                # all object classes advance together in the batched HIP Lloyd kernels (functions/kmeans.py): the convergence test
                # runs on the device, the host reads the `done` flags every 8 iterations — not once per class and iteration
                gen = torch.Generator(device=todo[0][1].device).manual_seed(self.seed)
                inits = [self.init(cid, x) for cid, x in todo] if self.init is not None else None
                cents, _ = _kmeans.kmeans_lloyd_batched([x for _, x in todo], self.num_clusters, inits=inits, generator=gen)
                for (cid, _), c in zip(todo, cents):
                    out[cid] = c.float()
            else:
                for cid, x in todo:
                    out[cid] = self._get_cluster_centroids(x, cid)
        if multi:                                                 # ... and everyone takes ITS result (reference :69-71)
            box = [{k: v.cpu() for k, v in out.items()}]
            dist.broadcast_object_list(box, src=0)
            out = {k: v.to(dev) for k, v in box[0].items()}
        return copy.deepcopy(out)

    def _get_cluster_centroids(self, x, cid):
        gen = torch.Generator(device=x.device).manual_seed(self.seed)
        init = self.init(cid, x) if self.init is not None else None
        centres, _, _ = _kmeans.kmeans_lloyd(x, self.num_clusters, init=init, generator=gen)
        return centres.float()
