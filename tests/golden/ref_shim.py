"""Import shim that lets the reference's Python modules run in THIS container.

Only used by tests/golden/make_golden.py (build container; /root/reference is
not present on the GPU box).  Nothing in here is product code and nothing here
copies reference source: it provides *stand-ins* for the third-party packages
the reference imports but which are absent offline (detectron2 0.6, fvcore,
timm, torchvision) following SURVEY.md Appendix D/E, then imports the
reference's leaf modules by path, skipping its package __init__ files (which
drag in wandb / pydensecrf / pycocotools).

The stand-ins below are the documented contract for the detectron2 surface the
hot path touches ("parity unpinned by the reference's own tests").
"""
import importlib
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class Registry(dict):
    def __init__(self, name):
        super().__init__()
        self._name = name

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self[o.__name__] = o
                return o
            return deco
        self[obj.__name__] = obj
        return obj

    def get(self, name):
        return self[name]


class ShapeSpec:
    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride


class Conv2d(nn.Conv2d):
    """detectron2.layers.Conv2d: conv -> norm -> activation."""

    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def get_norm(norm, out_channels):
    if norm is None or norm == "":
        return None
    if norm == "GN":
        return nn.GroupNorm(32, out_channels)
    raise ValueError(norm)


def c2_xavier_fill(module):
    nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def point_sample(input, point_coords, **kwargs):
    add_dim = False
    if point_coords.dim() == 3:
        add_dim = True
        point_coords = point_coords.unsqueeze(2)
    output = F.grid_sample(input, 2.0 * point_coords - 1.0, **kwargs)
    if add_dim:
        output = output.squeeze(3)
    return output


def get_uncertain_point_coords_with_randomness(coarse_logits, uncertainty_func, num_points,
                                               oversample_ratio, importance_sample_ratio):
    assert oversample_ratio >= 1
    assert 0 <= importance_sample_ratio <= 1
    num_boxes = coarse_logits.shape[0]
    num_sampled = int(num_points * oversample_ratio)
    point_coords = torch.rand(num_boxes, num_sampled, 2, device=coarse_logits.device)
    point_logits = point_sample(coarse_logits, point_coords, align_corners=False)
    point_uncertainties = uncertainty_func(point_logits)
    num_uncertain_points = int(importance_sample_ratio * num_points)
    num_random_points = num_points - num_uncertain_points
    idx = torch.topk(point_uncertainties[:, 0, :], k=num_uncertain_points, dim=1)[1]
    shift = num_sampled * torch.arange(num_boxes, dtype=torch.long, device=coarse_logits.device)
    idx += shift[:, None]
    point_coords = point_coords.view(-1, 2)[idx.view(-1), :].view(num_boxes, num_uncertain_points, 2)
    if num_random_points > 0:
        point_coords = torch.cat(
            [point_coords, torch.rand(num_boxes, num_random_points, 2, device=coarse_logits.device)], dim=1)
    return point_coords


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        mask = x.new_empty(shape).bernoulli_(keep)
        return x * mask / keep


def sem_seg_postprocess(result, img_size, output_height, output_width):
    """detectron2.modeling.postprocessing.sem_seg_postprocess (0.6): crop the padding, bilinear resize."""
    result = result[:, : img_size[0], : img_size[1]].expand(1, -1, -1, -1)
    return F.interpolate(result, size=(output_height, output_width), mode="bilinear", align_corners=False)[0]


class _Instances:
    """detectron2.structures.Instances, as far as the inference branches use it (attribute bag)."""

    def __init__(self, image_size, **kw):
        self.__dict__["_image_size"] = image_size
        self.__dict__["_fields"] = dict(kw)

    def __setattr__(self, k, v):
        self._fields[k] = v

    def __getattr__(self, k):
        try:
            return self.__dict__["_fields"][k]
        except KeyError:
            raise AttributeError(k)

    def has(self, k):
        return k in self._fields

    def to(self, *a, **k):
        return self


class _ImageList:
    """detectron2.structures.ImageList.from_tensors: zero-pad (bottom / right) to the largest size, rounded up to the
    divisibility."""

    def __init__(self, tensor, image_sizes):
        self.tensor, self.image_sizes = tensor, image_sizes

    @staticmethod
    def from_tensors(tensors, size_divisibility=0, pad_value=0.0):
        sizes = [tuple(t.shape[-2:]) for t in tensors]
        H, W = max(s[0] for s in sizes), max(s[1] for s in sizes)
        if size_divisibility > 1:
            H = (H + size_divisibility - 1) // size_divisibility * size_divisibility
            W = (W + size_divisibility - 1) // size_divisibility * size_divisibility
        out = tensors[0].new_full((len(tensors), tensors[0].shape[0], H, W), pad_value)
        for i, t in enumerate(tensors):
            out[i, :, : t.shape[-2], : t.shape[-1]] = t
        return _ImageList(out, sizes)


def install_inference_standins():
    """stand-ins needed only to IMPORT the eval-side meta-architectures (proposal generation): logging / visualisation /
    serialisation packages that the functions exercised by the goldens never call."""
    comm = sys.modules["detectron2.utils.comm"]
    comm.is_main_process = lambda: False
    comm.synchronize = lambda: None
    _mod("wandb")
    _mod("detectron2.data", MetadataCatalog=types.SimpleNamespace(get=lambda name: types.SimpleNamespace(save_path="/tmp/pd_ref_save", class_codes=[])))
    m = sys.modules["detectron2.modeling"]
    m.build_backbone = lambda cfg: None
    m.build_sem_seg_head = lambda cfg, shape: None
    _mod("detectron2.data.detection_utils", read_image=None)
    _mod("detectron2.modeling.backbone", Backbone=nn.Module)
    _mod("detectron2.modeling.postprocessing", sem_seg_postprocess=sem_seg_postprocess)
    _mod("detectron2.structures", ImageList=_ImageList, Instances=_Instances, BitMasks=None)
    _mod("detectron2.utils.memory", retry_if_cuda_oom=lambda f: f)
    _mod("detectron2.utils.visualizer", ColorMode=None, Visualizer=object, GenericMask=None, _create_text_labels=None)
    _mod("pycocotools")
    # mask IoU of pycocotools (maskApi.c rleIou, iscrowd = 0): |a & b| / |a | b| in double precision.  `encode` keeps the
    # dense mask; the reference only feeds its result back into `iou` (utils/utils.py:35-42).
    import numpy as _np

    def _iou(pr, gt, iscrowd):
        out = _np.zeros((len(pr), len(gt)), dtype=_np.float64)
        for i, a in enumerate(pr):
            for j, b in enumerate(gt):
                a_, b_ = _np.asarray(a).astype(bool), _np.asarray(b).astype(bool)
                inter, union = float((a_ & b_).sum()), float((a_ | b_).sum())
                out[i, j] = inter / union if union > 0 else 0.0
        return out
    _mod("pycocotools.mask", encode=lambda m: _np.asarray(m), iou=_iou)
    _mod("pydensecrf")
    _mod("pydensecrf.densecrf")
    _mod("pydensecrf.utils")
    if "detectron2.utils.visualizer" in sys.modules:
        sys.modules["detectron2.utils.visualizer"].Visualizer = type("Visualizer", (), {})
    return importlib.import_module("part_distillation.proposal_generation_model")


def load_clustering_module():
    """the reference's evaluation/clustering_module.py loaded as a FILE (its package __init__ pulls evaluators that need
    the COCO / LVIS APIs); stand-ins: a no-op DatasetEvaluator base and single-process comm.all_gather.  sklearn is real."""
    import importlib.util
    comm = sys.modules["detectron2.utils.comm"]
    comm.is_main_process = lambda: True
    comm.synchronize = lambda: None
    comm.all_gather = lambda x: [x]
    _mod("detectron2.evaluation", DatasetEvaluator=object)
    spec = importlib.util.spec_from_file_location("pd_ref_clustering_module",
                                                  "/root/reference/part_distillation/evaluation/clustering_module.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def install():
    """Install stand-ins into sys.modules and return the reference leaf modules."""
    if "part_distillation" in sys.modules and getattr(sys.modules["part_distillation"], "_shimmed", False):
        return sys.modules["part_distillation"]._leafs
    world = {"size": 1}
    _mod("detectron2")
    _mod("detectron2.config", configurable=lambda f: f)
    _mod("detectron2.layers", Conv2d=Conv2d, ShapeSpec=ShapeSpec, get_norm=get_norm, DeformConv=None)
    _mod("detectron2.modeling", SEM_SEG_HEADS_REGISTRY=Registry("SEM_SEG_HEADS"),
         BACKBONE_REGISTRY=Registry("BACKBONE"), META_ARCH_REGISTRY=Registry("META_ARCH"),
         Backbone=nn.Module, ShapeSpec=ShapeSpec)
    _mod("detectron2.utils")
    _mod("detectron2.utils.registry", Registry=Registry)
    _mod("detectron2.utils.comm", get_world_size=lambda: world["size"])
    _mod("detectron2.projects")
    _mod("detectron2.projects.point_rend")
    _mod("detectron2.projects.point_rend.point_features", point_sample=point_sample,
         get_uncertain_point_coords_with_randomness=get_uncertain_point_coords_with_randomness)
    _mod("fvcore")
    _mod("fvcore.nn")
    wi = _mod("fvcore.nn.weight_init", c2_xavier_fill=c2_xavier_fill)
    sys.modules["fvcore.nn"].weight_init = wi
    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.layers", DropPath=DropPath, to_2tuple=lambda x: (x, x) if not isinstance(x, tuple) else x,
         trunc_normal_=nn.init.trunc_normal_)
    _mod("torchvision", _is_tracing=lambda: False)
    _mod("MultiScaleDeformableAttention")

    base = REF_ROOT + "/part_distillation"
    pkgs = {
        "part_distillation": base,
        "part_distillation.utils": base + "/utils",
        "part_distillation.modeling": base + "/modeling",
        "part_distillation.modeling.backbone": base + "/modeling/backbone",
        "part_distillation.modeling.meta_arch": base + "/modeling/meta_arch",
        "part_distillation.modeling.pixel_decoder": base + "/modeling/pixel_decoder",
        "part_distillation.modeling.pixel_decoder.ops": base + "/modeling/pixel_decoder/ops",
        "part_distillation.modeling.pixel_decoder.ops.functions": base + "/modeling/pixel_decoder/ops/functions",
        "part_distillation.modeling.pixel_decoder.ops.modules": base + "/modeling/pixel_decoder/ops/modules",
        "part_distillation.modeling.transformer_decoder": base + "/modeling/transformer_decoder",
    }
    for name, path in pkgs.items():
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    imp = importlib.import_module
    leafs = types.SimpleNamespace(
        msda_func=imp("part_distillation.modeling.pixel_decoder.ops.functions.ms_deform_attn_func"),
        msda_mod=imp("part_distillation.modeling.pixel_decoder.ops.modules.ms_deform_attn"),
    )
    # the `.ops.modules` package __init__ normally re-exports MSDeformAttn
    sys.modules["part_distillation.modeling.pixel_decoder.ops.modules"].MSDeformAttn = leafs.msda_mod.MSDeformAttn
    leafs.posenc = imp("part_distillation.modeling.transformer_decoder.position_encoding")
    leafs.pixdec = imp("part_distillation.modeling.pixel_decoder.msdeformattn")
    leafs.fpn = imp("part_distillation.modeling.pixel_decoder.fpn")
    leafs.m2f_dec = imp("part_distillation.modeling.transformer_decoder.mask2former_transformer_decoder")
    leafs.pd_dec = imp("part_distillation.modeling.transformer_decoder.part_distillation_transformer_decoder")
    leafs.matcher = imp("part_distillation.modeling.matcher")
    leafs.criterion = imp("part_distillation.modeling.criterion")
    leafs.head = imp("part_distillation.modeling.meta_arch.mask_former_head")
    leafs.swin = imp("part_distillation.modeling.backbone.swin")
    leafs.misc = imp("part_distillation.utils.misc")
    leafs.world = world
    leafs.ShapeSpec = ShapeSpec
    sys.modules["part_distillation"]._shimmed = True
    sys.modules["part_distillation"]._leafs = leafs
    return leafs


if __name__ == "__main__":
    L = install()
    print("reference leaf modules imported:", [k for k in vars(L)])
