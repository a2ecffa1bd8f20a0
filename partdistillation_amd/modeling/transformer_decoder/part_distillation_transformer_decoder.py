"""Part-distillation decoder (reference
transformer_decoder/part_distillation_transformer_decoder.py:22-254): the class
head is ``Linear(hidden, N_obj*K + 1)`` in float64 and only the K columns of
each image's object class (+ the no-object column) are kept.

The reference multiplies by the whole [N_obj*K+1, hidden] matrix and slices
afterwards (:237-238, 215-230).  Here the K+1 needed weight rows are gathered
first and only they are multiplied — the same float64 products (only the
BLAS summation order may differ, ~1e-16 relative), and rows outside the slice
receive an exactly-zero gradient, as the reference's ``outputs.sum()*0`` trick
(:228) yields."""
import torch
from torch import nn

from ...functions.fused import upload_small

from ...compat import TRANSFORMER_DECODER_REGISTRY, configurable
from .mask2former_transformer_decoder import MultiScaleMaskedTransformerDecoder


@TRANSFORMER_DECODER_REGISTRY.register()
class PartDistillationTransformerDecoder(MultiScaleMaskedTransformerDecoder):
    @configurable
    def __init__(self, in_channels, mask_classification, *args, num_object_classes: int, num_part_classes: int, **kwargs):
        super().__init__(in_channels, mask_classification, *args, **kwargs)
        self.class_embed = nn.Linear(self.hidden_dim, num_part_classes * num_object_classes + 1).double()
        self.num_part_classes = num_part_classes
        # only the rows _prepare_extra selects receive a non-zero gradient: the data-parallel reducer exchanges just those
        # (engine/ddp.py: row-sparse groups)
        self.class_embed.weight._pd_row_sparse = self.class_embed.bias._pd_row_sparse = True

    @classmethod
    def from_config(cls, cfg, in_channels, mask_classification):
        ret = super().from_config(cfg, in_channels, mask_classification)
        ret["num_object_classes"] = cfg.PART_DISTILLATION.NUM_OBJECT_CLASSES
        ret["num_part_classes"] = cfg.PART_DISTILLATION.NUM_PART_CLASSES
        return ret

    def _prepare_extra(self, mask):
        """`mask` carries the targets (reference :148-149): rows of class_embed used per image."""
        targets = mask
        K = self.num_part_classes
        dev = self.class_embed.weight.device
        cls = upload_small([int(t["gt_object_class"]) for t in targets], torch.long, dev)      # no blocking pageable copy
        rows = cls[:, None] * K + torch.arange(K, device=dev)[None, :]
        last = torch.full((len(targets), 1), self.class_embed.weight.shape[0] - 1, dtype=torch.long, device=dev)
        rows = torch.cat([rows, last], dim=1)                                   # [B, K+1]
        self.class_embed.weight._pd_rows = self.class_embed.bias._pd_rows = rows.reshape(-1)
        return rows

    def _class_logits(self, decoder_output, rows):
        w = self.class_embed.weight[rows]                                       # [B,K+1,C] float64
        b = self.class_embed.bias[rows]                                         # [B,K+1]
        with torch.autocast(device_type=decoder_output.device.type, enabled=False):
            return torch.baddbmm(b.unsqueeze(1), decoder_output.double(), w.transpose(1, 2))

    def _class_logits_stacked(self, d, rows):
        Lp, B, Q, C = d.shape
        out = self._class_logits(d.transpose(0, 1).reshape(B, Lp * Q, C), rows)                # [B, (L+1)*Q, K+1]
        return out.view(B, Lp, Q, -1).transpose(0, 1)

    def apply_gradient_mask(self, outputs, targets):
        """reference :215-230, for callers that hold full-width logits."""
        K = self.num_part_classes
        sel = [outputs[i][:, int(t["gt_object_class"]) * K:(int(t["gt_object_class"]) + 1) * K]
               for i, t in enumerate(targets)]
        return torch.cat([torch.stack(sel, 0), outputs[:, :, -1:]], dim=-1) + outputs.sum() * 0

    def _finish(self, out, output):
        out["query_feats"] = output.permute(1, 0, 2)
