"""``build_model(config)``: the reference factory (mvuld/models/build.py:7-102) for the one model type the hot
path selects (``MODEL.TYPE == 'swinv2'``, :26-43).  swin / swin_moe / swin_mlp are upstream carry-overs that no
MVulD config uses and are out of scope."""
import torch

from .swin_transformer_v2 import SwinTransformerV2


def build_model(config, act_dtype=torch.bfloat16):
    model_type = config.MODEL.TYPE
    if model_type == 'swinv2':
        return SwinTransformerV2(img_size=config.DATA.IMG_SIZE,
                                 patch_size=config.MODEL.SWINV2.PATCH_SIZE,
                                 in_chans=config.MODEL.SWINV2.IN_CHANS,
                                 num_classes=config.MODEL.NUM_CLASSES,
                                 embed_dim=config.MODEL.SWINV2.EMBED_DIM,
                                 depths=config.MODEL.SWINV2.DEPTHS,
                                 num_heads=config.MODEL.SWINV2.NUM_HEADS,
                                 window_size=config.MODEL.SWINV2.WINDOW_SIZE,
                                 mlp_ratio=config.MODEL.SWINV2.MLP_RATIO,
                                 qkv_bias=config.MODEL.SWINV2.QKV_BIAS,
                                 drop_rate=config.MODEL.DROP_RATE,
                                 drop_path_rate=config.MODEL.DROP_PATH_RATE,
                                 ape=config.MODEL.SWINV2.APE,
                                 patch_norm=config.MODEL.SWINV2.PATCH_NORM,
                                 use_checkpoint=config.TRAIN.USE_CHECKPOINT,
                                 pretrained_window_sizes=config.MODEL.SWINV2.PRETRAINED_WINDOW_SIZES,
                                 act_dtype=act_dtype)
    raise NotImplementedError(f"Unkown or out-of-scope model: {model_type} (the MVulD hot path uses 'swinv2')")
