"""Oracle: the fused multimodal forward (+ loss) the north_star names.  TEST INFRASTRUCTURE.

Composition of the three reference module boundaries with one set of weights
(SURVEY.md section 0.2): SwinV2.forward_features (swin_transformer_v2.py:623-635)
-> img_embedding; MyUniXcoder.get_xcode_vec (unixcoder.py:33-38) -> func_text_embedding;
Multi_DefectModel_new_GCN.forward (GraphModel.py:150-211) -> logits;
CrossEntropyLoss / softmax as main_bigvul.py:298,328-333.

State-dict prefixes: ``swin.`` ``unixcoder.`` (then the reference's own
``encoder.`` prefix) and ``head.``.
"""
import torch.nn.functional as F

from .swin_ref import SwinCfg, swin_forward_features, swin_param_shapes
from .roberta_ref import RobertaCfg, unixcoder_sentence, roberta_param_shapes
from .head_ref import head_forward, head_param_shapes


def fused_forward(sd, images, ids, src, dst, batch_num_nodes, node_emb, pos_emb,
                  swin_cfg: SwinCfg, rob_cfg: RobertaCfg, training=False):
    img = swin_forward_features(sd, images, swin_cfg, prefix="swin.")
    _, txt = unixcoder_sentence(sd, ids, rob_cfg, prefix="unixcoder.encoder.")
    logits = head_forward(sd, src, dst, batch_num_nodes, node_emb, pos_emb, img, txt,
                          training=training, prefix="head.")
    return logits, img, txt


def fused_loss(sd, images, ids, src, dst, batch_num_nodes, node_emb, pos_emb, targets,
               swin_cfg, rob_cfg, training=True):
    logits, _, _ = fused_forward(sd, images, ids, src, dst, batch_num_nodes, node_emb, pos_emb,
                                 swin_cfg, rob_cfg, training)
    return F.cross_entropy(logits, targets), logits


def fused_param_shapes(swin_cfg: SwinCfg, rob_cfg: RobertaCfg, num_classes=2):
    P = {}
    P.update(swin_param_shapes(swin_cfg, "swin."))
    P.update(roberta_param_shapes(rob_cfg, "unixcoder.encoder."))
    P["unixcoder.classifier.weight"] = (2, rob_cfg.hidden_size)
    P["unixcoder.classifier.bias"] = (2,)
    P.update(head_param_shapes(num_classes, "head."))
    return P
