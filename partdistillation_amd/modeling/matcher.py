"""Hungarian matcher (reference modeling/matcher.py:74-201): per image, cost =
w_mask * point-BCE + w_class * (-p[target class]) + w_dice * point-dice over
``num_points`` random points shared by all masks of the image; one-to-one
assignment; pairs ordered by cost.

The assignment itself is solved by the HIP kernel ``pd_lsa_batched`` (one
workgroup per problem, float64 shortest-augmenting-path — the algorithm of
scipy.optimize.linear_sum_assignment the reference calls at :161) so no cost
matrix travels to the host."""
import torch
import torch.nn.functional as F
from torch import nn

from ..functions import lsa as lsa_op
from ..utils.misc import point_sample


def batch_dice_loss(inputs: torch.Tensor, targets: torch.Tensor):
    inputs = inputs.sigmoid().flatten(1)
    numerator = 2 * torch.einsum("nc,mc->nm", inputs, targets)
    denominator = inputs.sum(-1)[:, None] + targets.sum(-1)[None, :]
    return 1 - (numerator + 1) / (denominator + 1)


def batch_sigmoid_ce_loss(inputs: torch.Tensor, targets: torch.Tensor):
    """pos/neg BCE contracted with the targets; softplus(-x) = softplus(x) - x folds the two einsums of the
    reference (:57-62) into one contraction plus a row sum."""
    hw = inputs.shape[1]
    neg = F.softplus(inputs)                                  # BCE(x, 0)
    loss = neg.sum(-1)[:, None] - torch.einsum("nc,mc->nm", inputs, targets)
    return loss / hw


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_mask: float = 1, cost_dice: float = 1, num_points: int = 0):
        super().__init__()
        self.cost_class, self.cost_mask, self.cost_dice, self.num_points = cost_class, cost_mask, cost_dice, num_points
        assert cost_class != 0 or cost_mask != 0 or cost_dice != 0, "all costs cant be 0"
        self.rand = None                                      # replay hook for parity tests: rand(shape) -> [0,1)

    def _rand(self, shape, device):
        return (self.rand(shape).to(device) if self.rand is not None else torch.rand(shape, device=device))

    @torch.no_grad()
    def cost_matrix(self, logits, pred_masks, labels, tgt_masks, point_coords):
        """[Q, n] cost of one image (reference :108-158)."""
        prob = logits.sigmoid() if logits.shape[-1] == 1 else logits.float().softmax(-1)
        cost_class = -prob[:, labels]
        tgt = point_sample(tgt_masks[:, None].to(pred_masks), point_coords.repeat(tgt_masks.shape[0], 1, 1),
                           align_corners=False).squeeze(1)
        out = point_sample(pred_masks[:, None], point_coords.repeat(pred_masks.shape[0], 1, 1),
                           align_corners=False).squeeze(1)
        with torch.autocast(device_type=out.device.type, enabled=False):
            out, tgt = out.float(), tgt.float()
            cost_mask = batch_sigmoid_ce_loss(out, tgt)
            cost_dice = batch_dice_loss(out, tgt)
        return self.cost_mask * cost_mask + self.cost_class * cost_class + self.cost_dice * cost_dice

    @torch.no_grad()
    def memory_efficient_forward(self, outputs, targets):
        bs, num_queries = outputs["pred_logits"].shape[:2]
        costs = []
        for b in range(bs):
            coords = self._rand((1, self.num_points, 2), outputs["pred_masks"].device)
            C = self.cost_matrix(outputs["pred_logits"][b], outputs["pred_masks"][b], targets[b]["labels"],
                                 targets[b]["masks"], coords)
            costs.append(C.reshape(num_queries, -1))
        return lsa_op.solve_ragged(costs)

    @torch.no_grad()
    def forward(self, outputs, targets):
        """-> list (len B) of (index_i, index_j) int64 tensors (on the cost device), sorted by cost."""
        return self.memory_efficient_forward(outputs, targets)

    def __repr__(self, _repr_indent=4):
        body = [f"cost_class: {self.cost_class}", f"cost_mask: {self.cost_mask}", f"cost_dice: {self.cost_dice}"]
        return "\n".join(["Matcher " + self.__class__.__name__] + [" " * _repr_indent + line for line in body])
