"""The fused multimodal model the north_star names: SwinV2 image encoder + UniXcoder text encoder + graph/fusion
head in ONE forward/backward (the reference runs the encoders offline and caches their outputs:
mvuld/data/data_list.py:179-211,265-317).  Composition of the reference module
boundaries: ``img_embedding = swin.forward_features(images)``, ``func_text_embedding = unixcoder.get_xcode_vec(ids)[1]``,
``logits = head(g, img_embedding, func_text_embedding)``.  State-dict prefixes: ``swin.``, ``unixcoder.``, ``head.``."""
import torch
import torch.nn as nn

from .GraphModel import Multi_DefectModel_new_GCN, cross_entropy
from .swin_transformer_v2 import SwinTransformerV2
from .unixcoder import MyUniXcoder, RobertaConfigLite, RobertaModel


class FusedMVulD(nn.Module):
    def __init__(self, config, roberta_config=None, act_dtype=torch.bfloat16, swin=None):
        super().__init__()
        from .build import build_model
        self.act_dtype = act_dtype
        self.swin = swin if swin is not None else build_model(config, act_dtype)
        rc = roberta_config or RobertaConfigLite()
        self.unixcoder = MyUniXcoder(RobertaModel(rc, act_dtype), rc)
        for p in self.unixcoder.classifier.parameters():      # not on the fused path
            p.requires_grad_(False)
        for p in self.swin.head.parameters():
            p.requires_grad_(False)
        self.head = Multi_DefectModel_new_GCN(config, act_dtype=act_dtype)
        for n, p in self.head.named_parameters():
            if n.startswith(self.head.unused_parameter_prefixes):
                p.requires_grad_(False)

    def no_weight_decay(self):
        return {"swin." + k for k in self.swin.no_weight_decay()}

    def no_weight_decay_keywords(self):
        return self.swin.no_weight_decay_keywords()

    def forward(self, g, images, source_ids):
        img = self.swin.forward_features(images)                       # [B,1024]
        _, txt = self.unixcoder.get_xcode_vec(source_ids)              # [B,768]
        g.ndata["_FUNC_EMB"] = txt                                     # (per-node repeat is dead in the reference head)
        return self.head(g, img, txt)
