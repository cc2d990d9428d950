"""The single-modality "motivation" heads of the reference's ``models/MotivationModel.py`` on the kernels of the full head
(same constructor, ``forward(g, img_embedding, func_text_embedding)`` and state-dict keys; the parameters the reference constructs
and never uses are kept):
  Multi_DefectModel_Image    (:83-107)   final_fc on the Swin feature alone
  Multi_DefectModel_FuncText (:110-145)  final_fc on the function-level text feature alone
  Multi_DefectModel_Graph    (:148-256)  the full head's graph branch (GAT + positions + Rs_GCN), final_fc without its BatchNorm
  Multi_DefectModel_Graph1   (:259-348)  node MLP -> pad -> bn_gat / fc_gat(512->512) / ELU -> Rs_GCN x8 -> l2norm -> mean -> final_fc
  Multi_DefectModel_Graph2   (:351-426)  GAT x2 -> fc -> 8 hidden -> dgl.mean_nodes -> hbn / hfc / ELU -> final_fc"""
import torch
import torch.nn as nn

from .. import hip, ops
from .GraphModel import (_MeanNodesFn, _SegmentPadFn, _add_gat, _add_gcn, _cast_in, _gat_node_features, _gcn_readout, _head_common,
                         _hidden_stack, _padded_gcn_input, batch_norm, cast_to, linear_act)


def _classify(m, feats):
    return linear_act(feats, m.final_fc.weight, m.final_fc.bias, None, torch.float32)


class Multi_DefectModel_Image(nn.Module):
    def __init__(self, config, pretrained=True, attention=True, act_dtype=torch.bfloat16):
        super().__init__()
        _head_common(self, config, act_dtype, 0.2)
        self.swinbn = nn.BatchNorm1d(1024)
        self.swinfc = nn.Linear(1024, 512)
        self.final_fc = nn.Linear(1024, self.num_classes)
        self.final_fc_bn = nn.BatchNorm1d(1024)
        self.unused_parameter_prefixes = ("swinbn.", "swinfc.", "final_fc_bn.")

    def forward(self, g, img_embedding, func_text_embedding):
        hip.require_gpu(img_embedding)
        ops.USE_SPLIT3[0] = False
        return _classify(self, _cast_in(img_embedding, self.act_dtype))                                   # :106


class Multi_DefectModel_FuncText(nn.Module):
    def __init__(self, config, pretrained=True, attention=True, act_dtype=torch.bfloat16):
        super().__init__()
        _head_common(self, config, act_dtype, 0.2)
        self.fconly = nn.Linear(768, 512)
        self.hidden = nn.ModuleList([nn.Linear(512, 512) for _ in range(8)])
        self.bn_text = nn.BatchNorm1d(768)
        self.ln_text = nn.LayerNorm(768)
        self.fc_text = nn.Linear(768, 512)
        self.final_fc = nn.Linear(768, self.num_classes)
        self.final_fc_bn = nn.BatchNorm1d(768)
        self.unused_parameter_prefixes = ("fconly.", "hidden.", "bn_text.", "ln_text.", "fc_text.", "final_fc_bn.")

    def forward(self, g, img_embedding, func_text_embedding):
        hip.require_gpu(func_text_embedding)
        ops.USE_SPLIT3[0] = False
        return _classify(self, _cast_in(func_text_embedding, self.act_dtype))                             # :144


class Multi_DefectModel_Graph(nn.Module):
    def __init__(self, config, pretrained=True, attention=True, act_dtype=torch.bfloat16):
        super().__init__()
        _head_common(self, config, act_dtype, 0.2)
        _add_gat(self)
        self.fconly = nn.Linear(768, 512)
        self.hidden = nn.ModuleList([nn.Linear(512, 512) for _ in range(8)])
        _add_gcn(self)
        self.bn_gat = nn.BatchNorm1d(self.max_node)
        self.fc_gat = nn.Linear(512, 480)
        self.bn_bbox = nn.BatchNorm1d(self.max_node)
        self.fc_bbox = nn.Linear(4, 32)
        self.hbn = nn.BatchNorm1d(512)
        self.hln = nn.LayerNorm(512)
        self.hfc = nn.Linear(512, 512)
        self.final_fc = nn.Linear(512, self.num_classes)
        self.final_fc_bn = nn.BatchNorm1d(512)
        self.unused_parameter_prefixes = ("fconly.", "hbn.", "hln.", "hfc.", "final_fc_bn.")

    def forward(self, g, img_embedding, func_text_embedding):
        ad, tr, B = self.act_dtype, self.training, g.batch_size
        ops.USE_SPLIT3[0] = ad == torch.bfloat16
        h = g.ndata["_UNIX_NODE_EMB"]
        hip.require_gpu(h)
        h = _hidden_stack(self.hidden, _gat_node_features(self, g, _cast_in(h, ad), tr), self.p_hidden, tr)
        g.ndata['HGATOUTPUT'] = h
        g.ndata['HFGATOUTPUT'] = g.ndata["pos_emb"]
        return _classify(self, _gcn_readout(self, _padded_gcn_input(self, g, h, _cast_in(g.ndata["pos_emb"], ad), B), B))   # :254


class Multi_DefectModel_Graph1(nn.Module):
    def __init__(self, config, pretrained=True, attention=True, act_dtype=torch.bfloat16):
        super().__init__()
        _head_common(self, config, act_dtype, 0.2)
        self.fconly = nn.Linear(768, 512)
        self.hidden = nn.ModuleList([nn.Linear(512, 512) for _ in range(8)])
        _add_gcn(self)
        self.fc_gat = nn.Linear(512, 512)
        self.bn_gat = nn.BatchNorm1d(self.max_node)
        self.bn_text = nn.BatchNorm1d(768)
        self.ln_text = nn.LayerNorm(768)
        self.fc_text = nn.Linear(768, 512)
        self.hbn = nn.BatchNorm1d(512)
        self.hln = nn.LayerNorm(512)
        self.hfc = nn.Linear(512, 512)
        self.final_fc = nn.Linear(512, self.num_classes)
        self.final_fc_bn = nn.BatchNorm1d(512 * 3)
        self.unused_parameter_prefixes = ("bn_text.", "ln_text.", "fc_text.", "hbn.", "hln.", "hfc.", "final_fc_bn.")

    def forward(self, g, img_embedding, func_text_embedding):
        ad, tr, B = self.act_dtype, self.training, g.batch_size
        ops.USE_SPLIT3[0] = ad == torch.bfloat16
        h = g.ndata["_UNIX_NODE_EMB"]
        hip.require_gpu(h)
        h = linear_act(_cast_in(h, ad), self.fconly.weight, self.fconly.bias, "elu", None, self.p_mlp, tr)
        h = _hidden_stack(self.hidden, h, self.p_hidden, tr)
        g.ndata['HGATOUTPUT'] = h
        g.ndata['HFGATOUTPUT'] = g.ndata["pos_emb"]
        h_i = batch_norm(_SegmentPadFn.apply(h, g.index()["node_offsets"], B, self.max_node), self.bn_gat)
        rows = linear_act(h_i, self.fc_gat.weight, self.fc_gat.bias, "elu").view(B * self.max_node, 512)
        return _classify(self, _gcn_readout(self, rows, B))                                               # :346


class Multi_DefectModel_Graph2(nn.Module):
    def __init__(self, config, pretrained=True, attention=True, act_dtype=torch.bfloat16):
        super().__init__()
        _head_common(self, config, act_dtype, 0.1)
        _add_gat(self)
        self.fconly = nn.Linear(768, 512)
        self.hidden = nn.ModuleList([nn.Linear(512, 512) for _ in range(8)])
        self.hbn = nn.BatchNorm1d(512)
        self.hfc = nn.Linear(512, 512)
        self.final_fc = nn.Linear(512, self.num_classes)
        self.unused_parameter_prefixes = ("fconly.",)

    def forward(self, g, img_embedding, func_text_embedding):
        ad, tr = self.act_dtype, self.training
        ops.USE_SPLIT3[0] = False
        h = g.ndata["_UNIX_NODE_EMB"]
        hip.require_gpu(h)
        h = _hidden_stack(self.hidden, _gat_node_features(self, g, _cast_in(h, ad), tr), self.p_hidden, tr)
        hmean = cast_to(_MeanNodesFn.apply(h, g.index()["node_offsets"], g.batch_size), torch.float32)
        return _classify(self, linear_act(batch_norm(hmean, self.hbn), self.hfc.weight, self.hfc.bias, "elu"))   # :420-424
