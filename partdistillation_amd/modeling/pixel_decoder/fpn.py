"""``build_pixel_decoder`` (reference pixel_decoder/fpn.py:25-37)."""
from ...compat import SEM_SEG_HEADS_REGISTRY


def build_pixel_decoder(cfg, input_shape):
    name = cfg.MODEL.SEM_SEG_HEAD.PIXEL_DECODER_NAME
    model = SEM_SEG_HEADS_REGISTRY.get(name)(cfg, input_shape)
    if not callable(getattr(model, "forward_features", None)):
        raise ValueError("Only SEM_SEG_HEADS with forward_features method can be used as pixel decoder. "
                         f"Please implement forward_features for {name} to only return mask features.")
    return model
