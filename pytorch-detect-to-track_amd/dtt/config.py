"""Global `cfg` with the reference's key set and defaults (lib/model/utils/config.py:11-302), plus
cfg_from_file / cfg_from_list with the same strict key + type checks (config.py:337-399), so the
reference's cfgs/*.yml load unchanged.  Only the keys marked (*) are read by the hot path."""
import ast
import os

import numpy as np
import yaml


class AttrDict(dict):
    """dict with attribute access (stands in for easydict.EasyDict, which the reference imports)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))

cfg = AttrDict({
    "TRAIN": {
        "LEARNING_RATE": 0.001, "MOMENTUM": 0.9, "WEIGHT_DECAY": 0.0005, "GAMMA": 0.1, "STEPSIZE": [30000],
        "DISPLAY": 10, "DOUBLE_BIAS": True, "TRUNCATED": False, "BIAS_DECAY": False, "USE_GT": False,
        "ASPECT_GROUPING": False, "SNAPSHOT_KEPT": 3, "SUMMARY_INTERVAL": 180, "SCALES": (600,), "MAX_SIZE": 1000,
        "TRIM_HEIGHT": 600, "TRIM_WIDTH": 600, "IMS_PER_BATCH": 1,
        "BATCH_SIZE": 128,                       # (*) RoIs sampled per image by the proposal-target layer
        "FG_FRACTION": 0.25, "FG_THRESH": 0.5, "BG_THRESH_HI": 0.5, "BG_THRESH_LO": 0.1, "USE_FLIPPED": True,
        "BBOX_REG": True, "BBOX_THRESH": 0.5, "SNAPSHOT_ITERS": 5000, "SNAPSHOT_PREFIX": "res101_faster_rcnn",
        "BBOX_NORMALIZE_TARGETS": True, "BBOX_INSIDE_WEIGHTS": (1.0, 1.0, 1.0, 1.0),
        "BBOX_NORMALIZE_TARGETS_PRECOMPUTED": True, "BBOX_NORMALIZE_MEANS": (0.0, 0.0, 0.0, 0.0),
        "BBOX_NORMALIZE_STDS": (0.1, 0.1, 0.2, 0.2), "PROPOSAL_METHOD": "gt", "HAS_RPN": True,
        "RPN_POSITIVE_OVERLAP": 0.7,             # (*)
        "RPN_NEGATIVE_OVERLAP": 0.3,             # (*)
        "RPN_CLOBBER_POSITIVES": False,          # (*)
        "RPN_FG_FRACTION": 0.5,                  # (*)
        "RPN_BATCHSIZE": 256,                    # (*)
        "RPN_NMS_THRESH": 0.7,                   # (*)
        "RPN_PRE_NMS_TOP_N": 12000,              # (*)
        "RPN_POST_NMS_TOP_N": 2000,              # (*)
        "RPN_MIN_SIZE": 8,
        "RPN_BBOX_INSIDE_WEIGHTS": (1.0, 1.0, 1.0, 1.0),  # (*)
        "RPN_POSITIVE_WEIGHT": -1.0,             # (*)
        # not in the reference: how the RoI sampler consumes numpy's generator (dtt/targets.py): "device" = counts stay on
        # the GPU, "reference" = the reference's exact draw order (one 8-byte host read per image)
        "SAMPLER_RNG": "device",
        "USE_ALL_GT": True, "BN_TRAIN": False,
    },
    "TEST": {
        "SCALES": (600,), "MAX_SIZE": 1000,
        "NMS": 0.3,                              # (*) per-class NMS in the test driver
        "SVM": False, "BBOX_REG": True, "HAS_RPN": False, "PROPOSAL_METHOD": "gt",
        "RPN_NMS_THRESH": 0.7,                   # (*)
        "RPN_PRE_NMS_TOP_N": 6000,               # (*)
        "RPN_POST_NMS_TOP_N": 300,               # (*)
        "RPN_MIN_SIZE": 16, "MODE": "nms", "RPN_TOP_N": 5000,
    },
    "RESNET": {"MAX_POOL": False, "FIXED_BLOCKS": 1},
    "MOBILENET": {"REGU_DEPTH": False, "FIXED_LAYERS": 5, "WEIGHT_DECAY": 0.00004, "DEPTH_MULTIPLIER": 1.0},
    "DEDUP_BOXES": 1.0 / 16.0,
    "PIXEL_MEANS": np.array([[[102.9801, 115.9465, 122.7717]]]),
    "RNG_SEED": 3,                               # (*) seeds numpy for the anchor / RoI subsampling
    "EPS": 1e-14,
    "ROOT_DIR": _ROOT,
    "DATA_DIR": os.environ.get("DTT_DATA_DIR") or os.path.join(_ROOT, "data"),   # devkits live in DATA_DIR/ILSVRC
    "MATLAB": "matlab",
    "EXP_DIR": "default",
    "USE_GPU_NMS": True,
    "GPU_ID": 0,
    "POOLING_MODE": "crop",
    "POOLING_SIZE": 7,                           # (*)
    "MAX_NUM_GT_BOXES": 20,                      # (*) drivers raise it to 30
    "ANCHOR_SCALES": [8, 16, 32],                # (*) drivers set [4, 8, 16, 32]
    "ANCHOR_RATIOS": [0.5, 1, 2],                # (*)
    "FEAT_STRIDE": [16],                         # (*)
    "CUDA": False,
    "CROP_RESIZE_WITH_MAX_POOL": True,
    # not in the reference's config: its _RFCN hard-codes max_displacement = 8 (rfcn.py:58-60).  BASELINE.json's
    # config 5 asks for d = 16; set via `--set CORR_MAX_DISPLACEMENT 16` (changes corr_bbox_net's input width, so
    # checkpoints are only interchangeable at the default)
    "CORR_MAX_DISPLACEMENT": 8,
    # not in the reference's config either: its _RFCN ignores POOLING_MODE and always pools with PSRoI (rfcn.py:40-44,
    # SURVEY appendix A #4).  BASELINE.json's config 5 asks for "RoI-Align path + correlation d=16": with
    # RFCN_ROI_FEATURES = "align" | "pool" | "crop" the detector ALSO pools the 512-channel `top` map with the op the
    # legacy head selects by POOLING_MODE (faster_rcnn.py:72-83) for every RoI it scores and exposes the result as
    # `model.roi_feat` (the 10-tuple of rfcn.py:249-250 is unchanged).  "" = reference behaviour.
    "RFCN_ROI_FEATURES": "",
})


import copy as _copy

_DEFAULTS = _copy.deepcopy(cfg)


def reset_cfg():
    """Put the process-wide cfg back to its import-time values, in place (every module holds a reference to the same
    object).  For processes that run several independent configurations one after the other -- the test suite."""
    def put(live, saved):
        for k in [k for k in live if k not in saved]:
            del live[k]
        for k, v in saved.items():
            if isinstance(v, AttrDict) and isinstance(live.get(k), AttrDict):
                put(live[k], v)
            else:
                live[k] = _copy.deepcopy(v)
    put(cfg, _DEFAULTS)
    return cfg


def _merge(src, dst, path=""):
    for k, v in src.items():
        if k not in dst:
            raise KeyError("{} is not a valid config key".format(path + k))
        old = dst[k]
        if isinstance(old, AttrDict):
            if not isinstance(v, dict):
                raise ValueError("config key {} must be a mapping".format(path + k))
            _merge(v, old, path + k + ".")
            continue
        if type(old) is not type(v):
            if isinstance(old, np.ndarray):
                v = np.array(v, dtype=old.dtype)
            elif isinstance(old, tuple) and isinstance(v, list):
                v = tuple(v)  # YAML has no tuples; the reference's EasyDict turns lists into tuples
            elif isinstance(old, list) and isinstance(v, tuple):
                v = list(v)
            else:
                raise ValueError("Type mismatch ({} vs. {}) for config key: {}".format(type(old), type(v), path + k))
        dst[k] = v


def cfg_from_file(filename):
    """Merge a YAML file into cfg (config.py:370-376; SafeLoader instead of the bare yaml.load)."""
    with open(filename, "r") as f:
        data = yaml.safe_load(f) or {}
    _merge(data, cfg)


def cfg_from_list(cfg_list):
    """['TRAIN.RPN_BATCHSIZE', '128', ...] overrides, values via literal_eval (config.py:379-399)."""
    if len(cfg_list) % 2 != 0:
        raise ValueError("cfg_from_list expects KEY VALUE pairs")
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        d = cfg
        parts = k.split(".")
        for sub in parts[:-1]:
            if sub not in d:
                raise KeyError(k)
            d = d[sub]
        if parts[-1] not in d:
            raise KeyError(k)
        try:
            value = ast.literal_eval(v) if isinstance(v, str) else v
        except (ValueError, SyntaxError):
            value = v
        old = d[parts[-1]]
        if isinstance(old, tuple) and isinstance(value, list):
            value = tuple(value)
        if type(value) is not type(old):
            raise TypeError("type {} does not match original type {} for {}".format(type(value), type(old), k))
        d[parts[-1]] = value


def apply_dataset_defaults(dataset="imagenet_vid"):
    """The per-dataset `set_cfgs` the drivers hard-code (trainval_net.py:162-170, test_net.py:100-118)."""
    if dataset in ("imagenet_vid", "imagenet_vid+imagenet_det", "imagenet_vid_det"):
        cfg_from_list(["ANCHOR_SCALES", "[4, 8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "30"])
    else:
        cfg_from_list(["ANCHOR_SCALES", "[8, 16, 32]", "ANCHOR_RATIOS", "[0.5,1,2]", "MAX_NUM_GT_BOXES", "20"])
