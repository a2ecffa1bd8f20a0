"""Minimal yacs/detectron2-compatible config: ``CfgNode`` (attribute access,
YAML ``_BASE_`` inheritance, ``merge_from_file`` / ``merge_from_list``),
``get_cfg()`` with the detectron2 0.6 defaults the hot path reads (SURVEY
Appendix F) and the ``@configurable`` decorator."""
import ast
import copy
import functools
import inspect
import os

import yaml

BASE_KEY = "_BASE_"


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError(f"Attempted to set {name} on a frozen CfgNode")
        self[name] = value

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def _set_frozen(self, flag):
        object.__setattr__(self, "_frozen", flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            out[k] = copy.deepcopy(v, memo)
        return out

    # ------------------------------------------------------------------ merging
    @staticmethod
    def load_yaml_with_base(filename):
        with open(filename, "r") as f:
            cfg = yaml.safe_load(f) or {}
        if BASE_KEY in cfg:
            base = cfg.pop(BASE_KEY)
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(filename), base)
            merged = CfgNode.load_yaml_with_base(base)
            _merge_dict(cfg, merged)
            return merged
        return cfg

    def merge_from_file(self, filename, allow_unsafe=False):
        self._merge(CfgNode.load_yaml_with_base(filename), [])

    def merge_from_other_cfg(self, other):
        self._merge(other, [])

    def _merge(self, src, path):
        for k, v in src.items():
            if k not in self:
                raise KeyError("Non-existent config key: " + ".".join(path + [k]))
            if isinstance(v, dict) and isinstance(self[k], CfgNode):
                self[k]._merge(v, path + [k])
            else:
                self[k] = _coerce(v, self[k], ".".join(path + [k]))

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0, "override list has odd length"
        for full_key, v in zip(opts[0::2], opts[1::2]):
            node = self
            keys = full_key.split(".")
            for sub in keys[:-1]:
                if sub not in node:
                    raise KeyError(f"Non-existent config key: {full_key}")
                node = node[sub]
            if keys[-1] not in node:
                raise KeyError(f"Non-existent config key: {full_key}")
            node[keys[-1]] = _coerce(_decode(v), node[keys[-1]], full_key)

    def dump(self):
        return yaml.safe_dump(_to_plain(self))


def _to_plain(x):
    if isinstance(x, dict):
        return {k: _to_plain(v) for k, v in x.items()}
    if isinstance(x, tuple):
        return list(x)
    return x


def _merge_dict(src, dst):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge_dict(v, dst[k])
        else:
            dst[k] = v


def _decode(v):
    if not isinstance(v, str):
        return v
    try:
        return ast.literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _coerce(new, old, key):
    if old is None or new is None or type(new) is type(old):
        return new
    if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
        return float(new)
    if isinstance(old, (list, tuple)) and isinstance(new, (list, tuple)):
        return type(old)(new)
    if isinstance(old, CfgNode) and isinstance(new, dict):
        return CfgNode(new)
    if isinstance(old, str) and not isinstance(new, str):
        raise ValueError(f"Type mismatch for {key}: {type(old)} vs {type(new)}")
    if isinstance(old, bool) != isinstance(new, bool) or isinstance(old, (int, float)) != isinstance(new, (int, float)):
        raise ValueError(f"Type mismatch ({type(old)} vs. {type(new)}) for config key: {key}")
    return new


CN = CfgNode


def get_cfg() -> CfgNode:
    """detectron2 0.6 defaults, restricted to what the hot path and its YAMLs touch."""
    C = CN()
    C.VERSION = 2
    C.SEED = -1
    C.OUTPUT_DIR = "./output"
    C.CUDNN_BENCHMARK = False
    C.VIS_PERIOD = 0
    C.MODEL = CN(dict(
        DEVICE="cuda", META_ARCHITECTURE="GeneralizedRCNN", WEIGHTS="", MASK_ON=False, KEYPOINT_ON=False,
        LOAD_PROPOSALS=False, PIXEL_MEAN=[103.530, 116.280, 123.675], PIXEL_STD=[1.0, 1.0, 1.0],
        BACKBONE=dict(NAME="build_resnet_backbone", FREEZE_AT=2),
        RESNETS=dict(DEPTH=50, OUT_FEATURES=["res4"], NUM_GROUPS=1, NORM="FrozenBN", WIDTH_PER_GROUP=64,
                     STRIDE_IN_1X1=True, RES5_DILATION=1, RES2_OUT_CHANNELS=256, STEM_OUT_CHANNELS=64,
                     DEFORM_ON_PER_STAGE=[False, False, False, False], DEFORM_MODULATED=False,
                     DEFORM_NUM_GROUPS=1, STEM_TYPE="basic", RES4_DILATION=1, RES5_MULTI_GRID=[1, 2, 4]),
        SEM_SEG_HEAD=dict(NAME="SemSegFPNHead", IN_FEATURES=["p2", "p3", "p4", "p5"], IGNORE_VALUE=255,
                          NUM_CLASSES=54, CONVS_DIM=128, COMMON_STRIDE=4, NORM="GN", LOSS_WEIGHT=1.0,
                          LOSS_TYPE="hard_pixel_mining", PROJECT_FEATURES=["res2"], PROJECT_CHANNELS=[48],
                          ASPP_CHANNELS=256, ASPP_DILATIONS=[6, 12, 18], ASPP_DROPOUT=0.1,
                          USE_DEPTHWISE_SEPARABLE_CONV=False),
    ))
    C.INPUT = CN(dict(MIN_SIZE_TRAIN=(800,), MIN_SIZE_TRAIN_SAMPLING="choice", MAX_SIZE_TRAIN=1333, MIN_SIZE_TEST=800,
                      MAX_SIZE_TEST=1333, RANDOM_FLIP="horizontal", FORMAT="BGR", MASK_FORMAT="polygon",
                      CROP=dict(ENABLED=False, TYPE="relative_range", SIZE=[0.9, 0.9])))
    C.DATASETS = CN(dict(TRAIN=(), TEST=(), PROPOSAL_FILES_TRAIN=(), PROPOSAL_FILES_TEST=()))
    C.DATALOADER = CN(dict(NUM_WORKERS=4, ASPECT_RATIO_GROUPING=True, SAMPLER_TRAIN="TrainingSampler",
                           REPEAT_THRESHOLD=0.0, FILTER_EMPTY_ANNOTATIONS=True))
    C.SOLVER = CN(dict(
        LR_SCHEDULER_NAME="WarmupMultiStepLR", MAX_ITER=40000, BASE_LR=0.001, MOMENTUM=0.9, NESTEROV=False,
        WEIGHT_DECAY=0.0001, WEIGHT_DECAY_NORM=0.0, GAMMA=0.1, STEPS=(30000,), WARMUP_FACTOR=1.0 / 1000,
        WARMUP_ITERS=1000, WARMUP_METHOD="linear", CHECKPOINT_PERIOD=5000, IMS_PER_BATCH=16,
        REFERENCE_WORLD_SIZE=0, BIAS_LR_FACTOR=1.0, WEIGHT_DECAY_BIAS=None, POLY_LR_POWER=0.9,
        POLY_LR_CONSTANT_ENDING=0.0,
        CLIP_GRADIENTS=dict(ENABLED=False, CLIP_TYPE="value", CLIP_VALUE=1.0, NORM_TYPE=2.0),
        AMP=dict(ENABLED=False)))
    C.TEST = CN(dict(EXPECTED_RESULTS=[], EVAL_PERIOD=0, DETECTIONS_PER_IMAGE=100,
                     AUG=dict(ENABLED=False, MIN_SIZES=(400, 500, 600, 700, 800, 900, 1000, 1100, 1200),
                              MAX_SIZE=4000, FLIP=True)))
    return C


# ---------------------------------------------------------------------- configurable
def configurable(init_func=None, *, from_config=None):
    """detectron2.config.configurable: ``Cls(cfg, ...)`` routes through
    ``Cls.from_config(cfg, ...)``; explicit keyword construction is untouched."""
    if init_func is not None:
        assert inspect.isfunction(init_func) and init_func.__name__ == "__init__"

        @functools.wraps(init_func)
        def wrapped(self, *args, **kwargs):
            fc = getattr(type(self), "from_config", None)
            if fc is None or not _called_with_cfg(*args, **kwargs):
                return init_func(self, *args, **kwargs)
            explicit = _get_args_from_config(fc, *args, **kwargs)
            return init_func(self, **explicit)
        return wrapped

    def wrapper(orig_func):
        @functools.wraps(orig_func)
        def wrapped(*args, **kwargs):
            if _called_with_cfg(*args, **kwargs):
                return orig_func(**_get_args_from_config(from_config, *args, **kwargs))
            return orig_func(*args, **kwargs)
        return wrapped
    return wrapper


def _called_with_cfg(*args, **kwargs):
    if len(args) and isinstance(args[0], CfgNode):
        return True
    return isinstance(kwargs.get("cfg"), CfgNode)


def _get_args_from_config(from_config_func, *args, **kwargs):
    sig = inspect.signature(from_config_func)
    if list(sig.parameters)[0] != "cfg":
        raise TypeError("from_config's first argument must be 'cfg'")
    var = any(p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD) for p in sig.parameters.values())
    if var:
        return from_config_func(*args, **kwargs)
    supported = set(sig.parameters)
    extra = {k: kwargs.pop(k) for k in list(kwargs) if k not in supported}
    ret = from_config_func(*args, **kwargs)
    ret.update(extra)
    return ret
