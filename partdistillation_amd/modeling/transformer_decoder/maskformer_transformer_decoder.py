"""``TRANSFORMER_DECODER_REGISTRY`` + ``build_transformer_decoder`` (reference
transformer_decoder/maskformer_transformer_decoder.py:19-30).  The MaskFormer-v1
``StandardTransformerDecoder`` is not on the hot path (no shipped YAML selects
it) and is not built."""
from ...compat import TRANSFORMER_DECODER_REGISTRY  # noqa: F401


def build_transformer_decoder(cfg, in_channels, mask_classification=True):
    name = cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME
    return TRANSFORMER_DECODER_REGISTRY.get(name)(cfg, in_channels, mask_classification)
