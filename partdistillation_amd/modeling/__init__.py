"""Importing this package registers every class of the hot path under the
reference's registry names (META_ARCH / BACKBONE / SEM_SEG_HEADS /
TRANSFORMER_DECODER)."""
from .backbone.resnet import build_resnet_backbone  # noqa: F401
from .backbone.swin import D2SwinTransformer  # noqa: F401
from .criterion import SetCriterion  # noqa: F401
from .matcher import HungarianMatcher  # noqa: F401
from .meta_arch.mask_former_head import MaskFormerHead  # noqa: F401
from .pixel_decoder.msdeformattn import MSDeformAttnPixelDecoder  # noqa: F401
from .transformer_decoder.mask2former_transformer_decoder import MultiScaleMaskedTransformerDecoder  # noqa: F401
from .transformer_decoder.part_distillation_transformer_decoder import PartDistillationTransformerDecoder  # noqa: F401
