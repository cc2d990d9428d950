"""Config tree of the hot path: same keys, defaults, yaml ``BASE`` includes, ``--opts`` and flag overrides as the
reference's ``mvuld/config.py`` (defaults :5-322, ``_update_config_from_file`` :324-336, ``update_config`` :339-390,
``get_config`` :393-400) -- without yacs (absent here): ``CfgNode`` below implements the subset the drivers use
(``clone / defrost / freeze / merge_from_file / merge_from_list / dump``, attribute access, immutability).

Only sub-trees read on the MVulD path are kept (``MODEL.SWIN`` / ``SWIN_MOE`` / ``SWIN_MLP`` belong to model types
no MVulD yaml selects).  Where the reference assigns a key twice, the later value is the default here (:140-148).
"""
import copy
import os

import yaml


class CfgNode(dict):
    """Attribute-style nested dict with yacs' freeze semantics."""

    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError(f"Attempted to set {k} to {v}, but CfgNode is immutable")
        self[k] = v

    def _set_frozen(self, flag):
        object.__setattr__(self, "_frozen", flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def is_frozen(self):
        return object.__getattribute__(self, "_frozen")

    def clone(self):
        c = CfgNode(copy.deepcopy(self.to_dict()))
        return c

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else v) for k, v in self.items()}

    def dump(self, **kw):
        return yaml.safe_dump(_plain(self.to_dict()), **kw)

    def _merge(self, other, path=""):
        for k, v in other.items():
            if k not in self:
                raise KeyError(f"Non-existent config key: {path}{k}")
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError(f"{path}{k}: expected a mapping")
                self[k]._merge(v, f"{path}{k}.")
            else:
                self[k] = _coerce(v, self[k], f"{path}{k}")

    def merge_from_file(self, cfg_filename):
        with open(cfg_filename, "r") as f:
            y = yaml.safe_load(f) or {}
        self._merge(y)

    def merge_from_other_cfg(self, other):
        self._merge(other.to_dict() if isinstance(other, CfgNode) else other)

    def merge_from_list(self, cfg_list):
        if len(cfg_list) % 2 != 0:
            raise ValueError(f"Override list has odd length: {cfg_list}; it must be a list of pairs")
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            node = self
            parts = full_key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError(f"Non-existent key: {full_key}")
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"Non-existent key: {full_key}")
            if isinstance(v, str):
                try:
                    v = yaml.safe_load(v)
                except yaml.YAMLError:
                    pass
            node[parts[-1]] = _coerce(v, node[parts[-1]], full_key)


def _plain(o):
    if isinstance(o, dict):
        return {k: _plain(v) for k, v in o.items()}
    if isinstance(o, tuple):
        return [_plain(v) for v in o]
    if isinstance(o, list):
        return [_plain(v) for v in o]
    return o


def _coerce(new, old, key):
    """yacs-style type check: same type, or a few safe casts (int->float, list<->tuple, anything over None)."""
    if old is None or new is None or type(new) is type(old):
        return new
    if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
        return float(new)
    if isinstance(old, float) and isinstance(new, str):
        return float(new)                       # yaml parses 1e-8 as a string
    if isinstance(old, tuple) and isinstance(new, list):
        return tuple(new)
    if isinstance(old, list) and isinstance(new, tuple):
        return list(new)
    raise ValueError(f"Type mismatch ({type(old)} vs. {type(new)}) with values ({old} vs. {new}) for config key: {key}")


_DEFAULTS = {
    "BASE": [""],
    "DATA": {
        "BATCH_SIZE": 128, "DATA_PATH": "datasets", "DATASET": "imagenet", "IMG_SIZE": 384, "INTERPOLATION": "bicubic",
        "ZIP_MODE": False, "CACHE_MODE": "part", "PIN_MEMORY": False, "NUM_WORKERS": 8,
    },
    "MODEL": {
        "TYPE": "swin2", "NAME": "swin_base_patch4_window7_224", "PRETRAINED": "", "RESUME": "", "NUM_CLASSES": 2,
        "DROP_RATE": 0.0, "DROP_PATH_RATE": 0.1, "LABEL_SMOOTHING": 0.1,
        "SWINV2": {
            "PATCH_SIZE": 4, "IN_CHANS": 3, "EMBED_DIM": 96, "DEPTHS": [2, 2, 6, 2], "NUM_HEADS": [3, 6, 12, 24],
            "WINDOW_SIZE": 7, "MLP_RATIO": 4.0, "QKV_BIAS": True, "APE": False, "PATCH_NORM": True,
            "PRETRAINED_WINDOW_SIZES": [0, 0, 0, 0],
        },
        "MULTI": {"RESUME": ""},
    },
    "TRAIN": {
        "START_EPOCH": 0, "EPOCHS": 500, "WARMUP_EPOCHS": 20,
        "WEIGHT_DECAY": 0.005, "BASE_LR": 5e-5, "WARMUP_LR": 5e-7, "MIN_LR": 5e-6,
        "CLIP_GRAD": 5.0, "AUTO_RESUME": False, "BEST_RESUME": True, "ACCUMULATION_STEPS": 1, "USE_CHECKPOINT": False,
        "LR_SCHEDULER": {"NAME": "cosine", "DECAY_EPOCHS": 30, "DECAY_RATE": 0.1},
        "OPTIMIZER": {"NAME": "adamw", "EPS": 1e-8, "BETAS": (0.9, 0.999), "MOMENTUM": 0.9},
        "MOE": {"SAVE_MASTER": False},
        "DATA_PATH": "datasets/total/train_balanced.txt",
    },
    "AUG": {
        "COLOR_JITTER": 0.4, "AUTO_AUGMENT": "rand-m9-mstd0.5-inc1", "REPROB": 0.25, "REMODE": "pixel", "RECOUNT": 1,
        "MIXUP": 0.8, "CUTMIX": 1.0, "CUTMIX_MINMAX": None, "MIXUP_PROB": 1.0, "MIXUP_SWITCH_PROB": 0.5, "MIXUP_MODE": "batch",
    },
    "TEST": {"CROP": False, "SEQUENTIAL": False, "SHUFFLE": False, "DATA_PATH": "datasets/total/test.txt"},
    "VAL": {"DATA_PATH": "datasets/total/valid.txt"},
    "AMP_ENABLE": True, "AMP_OPT_LEVEL": "", "OUTPUT": "output",
    "MULTI_OUTPUT": "myoutput/Multi_DefectModel_new_GCN/3",
    "TAG": "default", "SAVE_FREQ": 1, "PRINT_FREQ": 50, "SEED": 0, "EVAL_MODE": False, "THROUGHPUT_MODE": False,
    "LOCAL_RANK": 0,
    # ---- additions of this implementation (absent in the reference; all optional) ----
    "FUSED": {
        "ENABLE": True,            # train the three encoders in one step (north_star); False = reference-faithful head-only step
        "DTYPE": "bf16",           # activation storage: bf16 | fp32 | fp8 (= bf16 + forward encoder GEMMs in e4m3)
        "HEAD": "Multi_DefectModel_new_GCN",   # any head class of models/{GraphModel,new_model,MotivationModel}.py (main_bigvul.py:124-129)
        "SYNTHETIC": True,         # synthetic Big-Vul-shaped data (there is no dataset on the box)
        "DATA_ROOT": "",           # a corpus directory in the reference's file formats (data/bigvul_dataset.py: BigVulFiles) instead
        "SYNTH_TRAIN": 256, "SYNTH_VAL": 64, "SYNTH_TEST": 64,
        "SEQ_LEN": 512, "NODES_LO": 150, "NODES_HI": 250,
        "LINE_LEN": 64,            # BigVulFiles: per-line token ids of every function are padded (pad id 1) / truncated to this width
        # HIDDEN_DROPOUT / ATTN_DROPOUT: HF RobertaConfig defaults (unixcoder.py:107-110 builds the model from that config); active
        # whenever the text encoder is in train() mode, as it is in the fused step
        "TEXT": {"VOCAB": 51416, "HIDDEN": 768, "LAYERS": 12, "HEADS": 12, "INTERMEDIATE": 3072, "MAX_POS": 1026,
                 "HIDDEN_DROPOUT": 0.1, "ATTN_DROPOUT": 0.1},
    },
}

_C = CfgNode(_DEFAULTS)


def _update_config_from_file(config, cfg_file):
    config.defrost()
    with open(cfg_file, "r") as f:
        yaml_cfg = yaml.safe_load(f) or {}
    for cfg in yaml_cfg.setdefault("BASE", [""]):
        if cfg:
            _update_config_from_file(config, os.path.join(os.path.dirname(cfg_file), cfg))
    print("=> merge config from {}".format(cfg_file))
    config._merge(yaml_cfg)
    config.freeze()


def update_config(config, args):
    _update_config_from_file(config, args.cfg)
    config.defrost()
    if getattr(args, "opts", None):
        config.merge_from_list(args.opts)

    def has(name):
        return getattr(args, name, None)

    if has("batch_size"):
        config.DATA.BATCH_SIZE = args.batch_size
    if has("data_path"):
        config.DATA.DATA_PATH = args.data_path
    if has("test_data_path"):
        config.TEST.DATA_PATH = args.test_data_path
    if has("zip"):
        config.DATA.ZIP_MODE = True
    if has("cache_mode"):
        config.DATA.CACHE_MODE = args.cache_mode
    if has("pretrained"):
        config.MODEL.PRETRAINED = args.pretrained
    if has("resume"):
        config.MODEL.RESUME = args.resume
    if has("myresume"):
        config.MODEL.MULTI.RESUME = args.myresume
    if has("accumulation_steps"):
        config.TRAIN.ACCUMULATION_STEPS = args.accumulation_steps
    if has("use_checkpoint"):
        config.TRAIN.USE_CHECKPOINT = True
    if has("amp_opt_level"):
        print("[warning] Apex amp has been deprecated, please use pytorch amp instead!")
        if args.amp_opt_level == "O0":
            config.AMP_ENABLE = False
    if has("disable_amp"):
        config.AMP_ENABLE = False
    if has("output"):
        config.OUTPUT = args.output
    if has("tag"):
        config.TAG = args.tag
    if has("eval"):
        config.EVAL_MODE = True
    if has("throughput"):
        config.THROUGHPUT_MODE = True
    lr = getattr(args, "local_rank", None)
    config.LOCAL_RANK = int(os.environ.get("LOCAL_RANK", 0)) if lr is None else lr
    config.OUTPUT = os.path.join(config.OUTPUT, config.MODEL.NAME, config.TAG)
    config.MULTI_OUTPUT = os.path.join(config.MULTI_OUTPUT, config.MODEL.NAME, config.TAG)
    config.freeze()


def get_config(args):
    """A clone of the defaults with the yaml, ``--opts`` and flag overrides applied (reference :393-400)."""
    config = _C.clone()
    update_config(config, args)
    return config
