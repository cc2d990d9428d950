"""The two modality-ablation heads of the reference's ``models/new_model.py`` on the kernels of the full head
(same constructor, ``forward(g, img_embedding, func_text_embedding)`` and state-dict keys; unused parameters kept):
  Multi_DefectModel_noGlobalImage (new_model.py:81-199)  graph branch x text branch (elementwise product), no image feature
  Multi_DefectModel_noFunc        (new_model.py:202-326) image branch ++ graph branch, no function-level text feature
The graph branch is the full head's (GATConv x2 -> fc -> 8 hidden -> pad to 100 -> bn/fc + positions -> 8 x Rs_GCN -> l2norm -> mean)."""
import torch
import torch.nn as nn

from .. import hip, ops
from .GraphModel import (_ConcatColsFn, _MulFn, _add_gat, _add_gcn, _bn_classifier, _cast_in, _feature_branch, _gat_node_features,
                         _gcn_readout, _head_common, _hidden_stack, _padded_gcn_input, cast_to)


class _GraphBranchHead(nn.Module):
    FINAL_IN = 512

    def __init__(self, config, pretrained=True, attention=True, act_dtype=torch.bfloat16):
        super().__init__()
        _head_common(self, config, act_dtype, 0.2)
        _add_gat(self)
        self.fconly = nn.Linear(768, 512)
        self.hidden = nn.ModuleList([nn.Linear(512, 512) for _ in range(8)])
        _add_gcn(self)
        self.bn_text = nn.BatchNorm1d(768)
        self.ln_text = nn.LayerNorm(768)
        self.fc_text = nn.Linear(768, 512)
        self.bn_gat = nn.BatchNorm1d(self.max_node)
        self.fc_gat = nn.Linear(512, 480)
        self.bn_bbox = nn.BatchNorm1d(self.max_node)
        self.fc_bbox = nn.Linear(4, 32)
        self.swinbn = nn.BatchNorm1d(1024)
        self.swinfc = nn.Linear(1024, 512)
        self.hbn = nn.BatchNorm1d(512)
        self.hln = nn.LayerNorm(512)
        self.hfc = nn.Linear(512, 512)
        self.final_fc_bn = nn.BatchNorm1d(self.FINAL_IN)
        self.final_fc = nn.Linear(self.FINAL_IN, self.num_classes)

    def graph_feature(self, g):
        """[B, 512] fp32: new_model.py:160-194 / 284-318."""
        ad, tr, B = self.act_dtype, self.training, g.batch_size
        ops.USE_SPLIT3[0] = ad == torch.bfloat16
        h = g.ndata["_UNIX_NODE_EMB"]
        hip.require_gpu(h)
        h = _hidden_stack(self.hidden, _gat_node_features(self, g, _cast_in(h, ad), tr), self.p_hidden, tr)
        g.ndata['HGATOUTPUT'] = h
        g.ndata['HFGATOUTPUT'] = g.ndata["pos_emb"]
        return _gcn_readout(self, _padded_gcn_input(self, g, h, _cast_in(g.ndata["pos_emb"], ad), B), B)


class Multi_DefectModel_noGlobalImage(_GraphBranchHead):
    FINAL_IN = 512

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.unused_parameter_prefixes = ("fconly.", "ln_text.", "swinbn.", "swinfc.", "hbn.", "hln.", "hfc.")

    def forward(self, g, img_embedding, func_text_embedding):
        hip.require_gpu(func_text_embedding)
        t = _feature_branch(func_text_embedding, self.bn_text, self.fc_text, self.act_dtype)
        return _bn_classifier(self, _MulFn.apply(cast_to(t, torch.float32), self.graph_feature(g)))          # :196


class Multi_DefectModel_noFunc(_GraphBranchHead):
    FINAL_IN = 1024

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.unused_parameter_prefixes = ("fconly.", "bn_text.", "ln_text.", "fc_text.", "hbn.", "hln.", "hfc.")

    def forward(self, g, img_embedding, func_text_embedding):
        hip.require_gpu(img_embedding)
        x = _feature_branch(img_embedding, self.swinbn, self.swinfc, self.act_dtype)
        return _bn_classifier(self, _ConcatColsFn.apply(cast_to(x, torch.float32), self.graph_feature(g)))    # :320
