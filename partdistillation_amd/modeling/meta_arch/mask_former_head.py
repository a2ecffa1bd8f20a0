"""``MaskFormerHead`` (reference meta_arch/mask_former_head.py:22-143): owns the
pixel decoder and the transformer predictor and routes features between them."""
import logging
from typing import Dict

from torch import nn

from ...compat import SEM_SEG_HEADS_REGISTRY, ShapeSpec, configurable
from ..pixel_decoder.fpn import build_pixel_decoder
from ..transformer_decoder.maskformer_transformer_decoder import build_transformer_decoder

logger = logging.getLogger(__name__)


@SEM_SEG_HEADS_REGISTRY.register()
class MaskFormerHead(nn.Module):
    _version = 2

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        old, new = "sem_seg_head.pixel_decoder.pixel_decoder", "sem_seg_head.pixel_decoder"
        for k in list(state_dict.keys()):
            if old in k:
                state_dict[k.replace(old, new)] = state_dict.pop(k)
                logger.warning(f"{k} ==> {k.replace(old, new)}")
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    @configurable
    def __init__(self, input_shape: Dict[str, ShapeSpec], *, num_classes: int, pixel_decoder: nn.Module,
                 loss_weight: float = 1.0, ignore_value: int = -1, transformer_predictor: nn.Module,
                 transformer_in_feature: str):
        super().__init__()
        self.in_features = [k for k, _ in sorted(input_shape.items(), key=lambda kv: kv[1].stride)]
        self.ignore_value, self.common_stride, self.loss_weight = ignore_value, 4, loss_weight
        self.pixel_decoder = pixel_decoder
        self.predictor = transformer_predictor
        self.transformer_in_feature = transformer_in_feature
        self.num_classes = num_classes

    @classmethod
    def from_config(cls, cfg, input_shape: Dict[str, ShapeSpec]):
        feat = cfg.MODEL.MASK_FORMER.TRANSFORMER_IN_FEATURE
        if feat in ("transformer_encoder", "multi_scale_pixel_decoder"):
            in_ch = cfg.MODEL.SEM_SEG_HEAD.CONVS_DIM
        elif feat == "pixel_embedding":
            in_ch = cfg.MODEL.SEM_SEG_HEAD.MASK_DIM
        else:
            in_ch = input_shape[feat].channels
        return dict(input_shape={k: v for k, v in input_shape.items() if k in cfg.MODEL.SEM_SEG_HEAD.IN_FEATURES},
                    ignore_value=cfg.MODEL.SEM_SEG_HEAD.IGNORE_VALUE, num_classes=cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES,
                    pixel_decoder=build_pixel_decoder(cfg, input_shape), loss_weight=cfg.MODEL.SEM_SEG_HEAD.LOSS_WEIGHT,
                    transformer_in_feature=feat,
                    transformer_predictor=build_transformer_decoder(cfg, in_ch, mask_classification=True))

    def forward(self, features, mask=None):
        return self.layers(features, mask)

    def layers(self, features, mask=None):
        mask_features, encoder_features, multi_scale = self.pixel_decoder.forward_features(features)
        feat = self.transformer_in_feature
        if feat == "multi_scale_pixel_decoder":
            return self.predictor(multi_scale, mask_features, mask)
        if feat == "transformer_encoder":
            assert encoder_features is not None, "Please use the TransformerEncoderPixelDecoder."
            return self.predictor(encoder_features, mask_features, mask)
        if feat == "pixel_embedding":
            return self.predictor(mask_features, mask_features, mask)
        return self.predictor(features[feat], mask_features, mask)
