"""Set criterion (reference modeling/criterion.py:94-286): Hungarian matching,
weighted cross-entropy over the queries, point-sampled BCE + dice on the matched
pairs, repeated for every auxiliary (deep-supervision) output.

Same constructor, ``forward(outputs, targets) -> dict`` contract, loss names
and arithmetic; differences are in execution only:
  * ``num_masks`` is all-reduced once, never ``.item()``-ed (the reference
    syncs the host at :254) — it stays a device scalar;
  * matched indices stay on the device (see matcher.py).
"""
import torch
import torch.nn.functional as F
from torch import nn

from ..functions.fused import upload_small

from ..utils.misc import (collectives_active, get_uncertain_point_coords_with_randomness, get_world_size, is_dist_avail_and_initialized,
                          nested_tensor_from_tensor_list, point_sample)


def dice_loss(inputs: torch.Tensor, targets: torch.Tensor, num_masks):
    inputs = inputs.sigmoid().flatten(1)
    numerator = 2 * (inputs * targets).sum(-1)
    denominator = inputs.sum(-1) + targets.sum(-1)
    loss = 1 - (numerator + 1) / (denominator + 1)
    return loss.sum() / num_masks


def sigmoid_ce_loss(inputs: torch.Tensor, targets: torch.Tensor, num_masks):
    loss = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    return loss.mean(1).sum() / num_masks


def calculate_uncertainty(logits):
    assert logits.shape[1] == 1
    return -(torch.abs(logits))


class SetCriterion(nn.Module):
    def __init__(self, num_classes, matcher, weight_dict, eos_coef, losses, num_points, oversample_ratio,
                 importance_sample_ratio):
        super().__init__()
        self.num_classes, self.matcher, self.weight_dict, self.eos_coef, self.losses = \
            num_classes, matcher, weight_dict, eos_coef, losses
        empty_weight = torch.ones(self.num_classes + 1)
        empty_weight[-1] = self.eos_coef
        self.register_buffer("empty_weight", empty_weight)
        self.num_points, self.oversample_ratio, self.importance_sample_ratio = \
            num_points, oversample_ratio, importance_sample_ratio
        self.rand = None            # replay hook for parity tests (see matcher.rand)
        self._nm_cache = {}
        self.batched = True         # all heads in one pass (criterion_batched.py); False = the reference's per-head loop

    # ------------------------------------------------------------------ losses
    def loss_labels(self, outputs, targets, indices, num_masks):
        assert "pred_logits" in outputs
        src_logits = outputs["pred_logits"].float()
        idx = self._get_src_permutation_idx(indices)
        if idx is None:
            return {"loss_ce": src_logits.sum() * 0.0}
        target_classes_o = torch.cat([t["labels"][J] for t, (_, J) in zip(targets, indices)])
        target_classes = torch.full(src_logits.shape[:2], self.num_classes, dtype=torch.int64, device=src_logits.device)
        target_classes[idx] = target_classes_o
        return {"loss_ce": F.cross_entropy(src_logits.transpose(1, 2), target_classes, self.empty_weight)}

    def loss_masks(self, outputs, targets, indices, num_masks):
        assert "pred_masks" in outputs
        src_idx = self._get_src_permutation_idx(indices)
        tgt_idx = self._get_tgt_permutation_idx(indices)
        if src_idx is None or tgt_idx is None:
            zero = outputs["pred_masks"].sum() * 0.0
            return {"loss_mask": zero, "loss_dice": zero}
        src_masks = outputs["pred_masks"][src_idx]
        if "_padded_masks" in targets[0]:                       # padded once per step by the meta-arch
            target_masks = targets[0]["_padded_masks"]
        else:
            target_masks, _ = nested_tensor_from_tensor_list([t["masks"] for t in targets]).decompose()
        target_masks = target_masks[tgt_idx].to(src_masks)
        src_masks, target_masks = src_masks[:, None], target_masks[:, None]
        with torch.no_grad():
            point_coords = get_uncertain_point_coords_with_randomness(
                src_masks, calculate_uncertainty, self.num_points, self.oversample_ratio, self.importance_sample_ratio,
                rand=self.rand)
            point_labels = point_sample(target_masks, point_coords, align_corners=False).squeeze(1)
        point_logits = point_sample(src_masks, point_coords, align_corners=False).squeeze(1)
        with torch.autocast(device_type=point_logits.device.type, enabled=False):
            point_logits, point_labels = point_logits.float(), point_labels.float()
            return {"loss_mask": sigmoid_ce_loss(point_logits, point_labels, num_masks),
                    "loss_dice": dice_loss(point_logits, point_labels, num_masks)}

    @staticmethod
    def _perm(indices, which):
        parts = [(torch.full_like(p[which], i), p[which]) for i, p in enumerate(indices)]
        if len(parts) == 0:
            return None
        return torch.cat([a for a, _ in parts]), torch.cat([b for _, b in parts])

    def _get_src_permutation_idx(self, indices):
        return self._perm(indices, 0)

    def _get_tgt_permutation_idx(self, indices):
        return self._perm(indices, 1)

    def get_loss(self, loss, outputs, targets, indices, num_masks):
        loss_map = {"labels": self.loss_labels, "masks": self.loss_masks}
        assert loss in loss_map, f"do you really want to compute {loss} loss?"
        return loss_map[loss](outputs, targets, indices, num_masks)

    def prefetch_num_masks(self, targets, device):
        """multi-rank runs: start the scalar all-reduce of reference :252-254 NOW (the meta-architecture calls this as soon as
        the targets exist, before the backbone runs): it depends only on the targets, so by the time the criterion needs the
        value the collective finished long ago instead of sitting between the decoder and the losses."""
        if not collectives_active():
            return
        count = float(sum(len(t["labels"]) for t in targets))
        n = upload_small([count], torch.float, device)                # asynchronous: the host keeps running ahead
        work = torch.distributed.all_reduce(n, async_op=True)
        self._nm_pending = (count, n, work)

    def num_masks(self, targets, device):
        """average number of target masks per rank, clamped to >= 1 (reference :248-254), as a device scalar."""
        count = float(sum(len(t["labels"]) for t in targets))
        if not collectives_active():
            key = (count, str(device))                       # cached constant: no per-step host->device copy
            if key not in self._nm_cache:
                self._nm_cache[key] = torch.tensor(max(count, 1.0), dtype=torch.float, device=device)
            return self._nm_cache[key]
        pending, self._nm_pending = getattr(self, "_nm_pending", None), None
        if pending is not None and pending[0] == count:
            _, n, work = pending
            work.wait()                                      # the compute stream waits for the (long finished) collective
        else:
            n = upload_small([count], torch.float, device)
            torch.distributed.all_reduce(n)
        return torch.clamp(n / get_world_size(), min=1)[0]

    def _can_batch(self, outputs, targets):
        return (self.batched and "aux_outputs" in outputs and outputs["pred_logits"].is_cuda and len(targets) > 0
                and self.losses == ["labels", "masks"] and all(t["labels"].shape[0] > 0 for t in targets)
                and (outputs["pred_masks"] is not None or outputs.get("mask_embeds") is not None))

    def forward(self, outputs, targets):
        if self._can_batch(outputs, targets):
            from .criterion_batched import batched_set_criterion
            if "_padded_masks" in targets[0]:
                padded = targets[0]["_padded_masks"]
            else:
                padded, _ = nested_tensor_from_tensor_list([t["masks"] for t in targets]).decompose()
            return batched_set_criterion(self, outputs, targets, padded)
        if outputs.get("pred_masks") is None and outputs.get("mask_embeds") is not None:
            from .transformer_decoder.mask2former_transformer_decoder import materialize_masks
            outputs = materialize_masks(dict(outputs))          # the per-head loop needs the dense masks
        outputs_without_aux = {k: v for k, v in outputs.items() if k != "aux_outputs"}
        if self.rand is not None:
            self.matcher.rand = self.rand
        indices = self.matcher(outputs_without_aux, targets)
        num_masks = self.num_masks(targets, outputs["pred_logits"].device)
        losses = {}
        for loss in self.losses:
            losses.update(self.get_loss(loss, outputs, targets, indices, num_masks))
        if "aux_outputs" in outputs:
            for i, aux_outputs in enumerate(outputs["aux_outputs"]):
                indices = self.matcher(aux_outputs, targets)
                for loss in self.losses:
                    l_dict = self.get_loss(loss, aux_outputs, targets, indices, num_masks)
                    losses.update({k + f"_{i}": v for k, v in l_dict.items()})
        return losses

    def __repr__(self):
        body = [f"matcher: {self.matcher.__repr__(_repr_indent=8)}", f"losses: {self.losses}",
                f"weight_dict: {self.weight_dict}", f"num_classes: {self.num_classes}", f"eos_coef: {self.eos_coef}",
                f"num_points: {self.num_points}", f"oversample_ratio: {self.oversample_ratio}",
                f"importance_sample_ratio: {self.importance_sample_ratio}"]
        return "\n".join(["Criterion " + self.__class__.__name__] + ["    " + line for line in body])
