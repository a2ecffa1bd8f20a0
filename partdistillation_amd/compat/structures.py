"""The slice of detectron2.structures / detectron2.layers the hot path touches
(SURVEY Appendix D): ShapeSpec, ImageList.from_tensors, BitMasks, Instances."""
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

import torch


@dataclass
class ShapeSpec:
    channels: Optional[int] = None
    height: Optional[int] = None
    width: Optional[int] = None
    stride: Optional[int] = None


class ImageList:
    """Batch of images padded bottom/right to a common, divisible size."""

    def __init__(self, tensor: torch.Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    @staticmethod
    def from_tensors(tensors, size_divisibility: int = 0, pad_value: float = 0.0) -> "ImageList":
        sizes = [(int(t.shape[-2]), int(t.shape[-1])) for t in tensors]
        h = max(s[0] for s in sizes)
        w = max(s[1] for s in sizes)
        if size_divisibility > 1:
            d = size_divisibility
            h, w = (h + d - 1) // d * d, (w + d - 1) // d * d
        if len(tensors) == 1 and sizes[0] == (h, w):
            return ImageList(tensors[0].unsqueeze(0), sizes)
        out = tensors[0].new_full((len(tensors),) + tuple(tensors[0].shape[:-2]) + (h, w), pad_value)
        for i, t in enumerate(tensors):
            out[i, ..., : sizes[i][0], : sizes[i][1]].copy_(t)
        return ImageList(out, sizes)


class BitMasks:
    def __init__(self, tensor: torch.Tensor):
        self.tensor = tensor.to(torch.bool)

    def to(self, *a, **k):
        return BitMasks(self.tensor.to(*a, **k))

    def __len__(self):
        return self.tensor.shape[0]


class Instances:
    """Field bag with image_size; fields move together with ``.to``."""

    def __init__(self, image_size: Tuple[int, int], **fields: Any):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for k, v in fields.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            object.__setattr__(self, name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        fields = object.__getattribute__(self, "_fields")
        if name in fields:
            return fields[name]
        raise AttributeError(f"Cannot find field '{name}' in the given Instances!")

    def set(self, name, value):
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def get(self, name):
        return self._fields[name]

    def get_fields(self) -> Dict[str, Any]:
        return self._fields

    def to(self, *a, **k):
        ret = Instances(self._image_size)
        for key, v in self._fields.items():
            ret.set(key, v.to(*a, **k) if hasattr(v, "to") else v)
        return ret

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        return 0
