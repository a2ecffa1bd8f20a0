from .config import CfgNode, CN, configurable, get_cfg  # noqa: F401
from .registry import (BACKBONE_REGISTRY, META_ARCH_REGISTRY, SEM_SEG_HEADS_REGISTRY, TRANSFORMER_DECODER_REGISTRY,  # noqa: F401
                       Registry, build_backbone, build_model, build_sem_seg_head)
from .structures import BitMasks, ImageList, Instances, ShapeSpec  # noqa: F401
