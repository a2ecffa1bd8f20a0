"""Registry surface of detectron2 that the hot path is plugged into.

Same registry names and ``.register() / .get()`` API as
``detectron2.utils.registry.Registry``; when the real detectron2 is importable
the classes are additionally registered into ITS registries so
``detectron2.modeling.build_model(cfg)`` finds them (neither box has it today).
"""
from typing import Any, Dict, Iterator, Tuple


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._obj_map: Dict[str, Any] = {}
        self._mirror = None          # a real detectron2 registry, when present

    def _do_register(self, name, obj):
        if name in self._obj_map:
            raise AssertionError(f"An object named '{name}' was already registered in '{self._name}' registry!")
        self._obj_map[name] = obj
        if self._mirror is not None and name not in self._mirror:
            self._mirror.register(obj)

    def register(self, obj=None):
        if obj is None:
            def deco(func_or_class):
                self._do_register(func_or_class.__name__, func_or_class)
                return func_or_class
            return deco
        self._do_register(obj.__name__, obj)
        return obj

    def get(self, name: str):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return ret

    def __contains__(self, name):
        return name in self._obj_map

    def __iter__(self) -> Iterator[Tuple[str, Any]]:
        return iter(self._obj_map.items())

    def __repr__(self):
        return f"Registry of {self._name}: {sorted(self._obj_map)}"


META_ARCH_REGISTRY = Registry("META_ARCH")
BACKBONE_REGISTRY = Registry("BACKBONE")
SEM_SEG_HEADS_REGISTRY = Registry("SEM_SEG_HEADS")
TRANSFORMER_DECODER_REGISTRY = Registry("TRANSFORMER_MODULE")   # reference maskformer_transformer_decoder.py:19

try:                                                            # bridge to a real detectron2, if any
    from detectron2.modeling import (BACKBONE_REGISTRY as _B, META_ARCH_REGISTRY as _M,
                                     SEM_SEG_HEADS_REGISTRY as _S)
    META_ARCH_REGISTRY._mirror, BACKBONE_REGISTRY._mirror, SEM_SEG_HEADS_REGISTRY._mirror = _M, _B, _S
except Exception:                                               # noqa: BLE001 - absent offline
    pass


def build_model(cfg):
    """detectron2.modeling.build_model: META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)."""
    import torch
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model


def build_backbone(cfg, input_shape=None):
    from .structures import ShapeSpec
    if input_shape is None:
        input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    return BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)


def build_sem_seg_head(cfg, input_shape):
    return SEM_SEG_HEADS_REGISTRY.get(cfg.MODEL.SEM_SEG_HEAD.NAME)(cfg, input_shape)
