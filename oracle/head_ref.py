"""Oracle: graph/fusion head + GATConv + Rs_GCN, functional over a state_dict.  TEST INFRASTRUCTURE.

Follows /root/reference/mvuld/models/GraphModel.py:
  unbatch_features            :30-54   (pad with zero rows / truncate to max_node)
  l2norm                      :74-79   (over dim=1, no eps)
  Multi_DefectModel_new_GCN.forward :150-211  (constructor :83-148 for shapes)
and /root/reference/mvuld/models/Rs_GCN.py:52-73.

``gat_conv`` restates dgl-cu102==0.8.1 ``dgl.nn.pytorch.GATConv`` (third party,
absent from /root/reference; call sites GraphModel.py:99-105,167-170) from its
documented algorithm -- **parity unpinned**:
    ft = fc(x).view(N,H,O); el = (ft*attn_l).sum(-1); er = (ft*attn_r).sum(-1)
    e  = leaky_relu(el[src] + er[dst], 0.2)
    a  = softmax of e over the incoming edges of each dst (per head; every
         multi-edge and self-loop is its own term)
    out[dst] = sum_e a * ft[src] + bias
The dead ``h_func`` branch (:172,:177) never reaches the logits and is omitted.
Dropout is identity (eval or p = 0).
"""
import torch
import torch.nn.functional as F

MAX_NODE = 100


def gat_conv(sd, p, x, src, dst, heads=4, out=512, slope=0.2):
    N = x.shape[0]
    ft = F.linear(x, sd[p + "fc.weight"]).view(N, heads, out)
    el = (ft * sd[p + "attn_l"]).sum(-1)                      # [N,H]
    er = (ft * sd[p + "attn_r"]).sum(-1)
    e = F.leaky_relu(el[src] + er[dst], slope)                # [E,H]
    emax = torch.full((N, heads), float("-inf"), dtype=e.dtype)
    emax = emax.scatter_reduce(0, dst[:, None].expand(-1, heads), e, "amax", include_self=True)
    ex = torch.exp(e - emax[dst])
    den = torch.zeros(N, heads, dtype=e.dtype).index_add_(0, dst, ex)
    a = ex / den[dst]                                         # [E,H]
    msg = ft[src] * a[..., None]                              # [E,H,O]
    outp = torch.zeros(N, heads, out, dtype=x.dtype).index_add_(0, dst, msg)
    return outp + sd[p + "bias"].view(1, heads, out)


def unbatch_pad(h, batch_num_nodes, max_node=MAX_NODE):
    outs, o = [], 0
    for n in batch_num_nodes:
        n = int(n)
        seg = h[o:o + n]
        o += n
        if n < max_node:
            seg = torch.cat([seg, torch.zeros(max_node - n, *h.shape[1:], dtype=h.dtype)], 0)
        else:
            seg = seg[:max_node]
        outs.append(seg)
    return torch.stack(outs)


def _bn(sd, p, x, training, eps=1e-5, momentum=0.1):
    """BatchNorm1d over dim 1 of [B,C] or [B,C,L]."""
    rm, rv = sd[p + "running_mean"].clone(), sd[p + "running_var"].clone()
    return F.batch_norm(x, rm, rv, sd[p + "weight"], sd[p + "bias"], training, momentum, eps)


def rs_gcn(sd, p, v, training):
    """v: [B, D, N] -> [B, D, N]   (Rs_GCN.py:52-73)."""
    g_v = F.conv1d(v, sd[p + "g.weight"], sd[p + "g.bias"]).permute(0, 2, 1)
    th = F.conv1d(v, sd[p + "theta.weight"], sd[p + "theta.bias"]).permute(0, 2, 1)
    ph = F.conv1d(v, sd[p + "phi.weight"], sd[p + "phi.bias"])
    R = torch.matmul(th, ph)
    R = R / R.size(-1)
    y = torch.matmul(R, g_v).permute(0, 2, 1).contiguous()
    wy = F.conv1d(y, sd[p + "W.0.weight"], sd[p + "W.0.bias"])
    wy = _bn(sd, p + "W.1.", wy, training)
    return wy + v, R


def head_forward(sd, src, dst, batch_num_nodes, node_emb, pos_emb, img_embedding, func_text_embedding,
                 training=False, prefix="", return_parts=False):
    """Logits [B, num_classes] as Multi_DefectModel_new_GCN.forward (GraphModel.py:150-211)."""
    P = prefix
    x = F.elu(F.linear(_bn(sd, P + "swinbn.", img_embedding, training), sd[P + "swinfc.weight"], sd[P + "swinfc.bias"]))
    t = _bn(sd, P + "bn_text.", func_text_embedding, training)
    t = F.elu(F.linear(t, sd[P + "fc_text.weight"], sd[P + "fc_text.bias"]))

    h = gat_conv(sd, P + "gat.", node_emb, src, dst)
    h = h.reshape(h.shape[0], -1)
    h = gat_conv(sd, P + "gat2.", h, src, dst)
    h = h.reshape(h.shape[0], -1)
    h = F.elu(F.linear(h, sd[P + "fc.weight"], sd[P + "fc.bias"]))
    for i in range(8):
        h = F.elu(F.linear(h, sd[P + f"hidden.{i}.weight"], sd[P + f"hidden.{i}.bias"]))
    hgat = h
    h_i = unbatch_pad(h, batch_num_nodes)                      # [B,100,512]
    pos_i = unbatch_pad(pos_emb, batch_num_nodes)              # [B,100,4]
    h_i = F.elu(F.linear(_bn(sd, P + "bn_gat.", h_i, training), sd[P + "fc_gat.weight"], sd[P + "fc_gat.bias"]))
    pos_i = F.elu(F.linear(_bn(sd, P + "bn_bbox.", pos_i, training), sd[P + "fc_bbox.weight"], sd[P + "fc_bbox.bias"]))
    g = torch.cat([h_i, pos_i], dim=2).permute(0, 2, 1)        # [B,512,100]
    for i in range(1, 9):
        g, R = rs_gcn(sd, P + f"Rs_GCN_{i}.", g, training)
    g = g.permute(0, 2, 1)                                     # [B,100,512]
    g = g / g.pow(2).sum(dim=1, keepdim=True).sqrt()
    hf = g.mean(dim=1)
    allf = torch.cat([x, hf, t], dim=1)
    logits = F.linear(_bn(sd, P + "final_fc_bn.", allf, training), sd[P + "final_fc.weight"], sd[P + "final_fc.bias"])
    if return_parts:
        return logits, {"hgat": hgat, "hf": hf, "x": x, "t": t}
    return logits


def head_param_shapes(num_classes=2, prefix=""):
    hf, emb, H = 512, 768, 4
    P = {}
    for n, fin in (("gat", emb), ("gat2", hf * H)):
        P[f"{n}.fc.weight"] = (H * hf, fin)
        P[f"{n}.attn_l"] = (1, H, hf); P[f"{n}.attn_r"] = (1, H, hf)
        P[f"{n}.bias"] = (H * hf,)

    def lin(n, o, i):
        P[n + ".weight"] = (o, i); P[n + ".bias"] = (o,)

    def bn(n, c):
        P[n + ".weight"] = (c,); P[n + ".bias"] = (c,)
        P[n + ".running_mean"] = (c,); P[n + ".running_var"] = (c,)
        P[n + ".num_batches_tracked"] = ()

    def ln(n, c):
        P[n + ".weight"] = (c,); P[n + ".bias"] = (c,)

    lin("fc", hf, hf * H); lin("fconly", hf, emb)
    for i in range(8):
        lin(f"hidden.{i}", hf, hf)
    for i in range(1, 9):
        b = f"Rs_GCN_{i}."
        for n in ("g", "theta", "phi"):
            P[b + n + ".weight"] = (hf, hf, 1); P[b + n + ".bias"] = (hf,)
        P[b + "W.0.weight"] = (hf, hf, 1); P[b + "W.0.bias"] = (hf,)
        bn(b + "W.1", hf)
    bn("bn_text", emb); ln("ln_text", emb); lin("fc_text", hf, emb)
    bn("bn_gat", MAX_NODE); lin("fc_gat", 480, 512)
    bn("bn_bbox", MAX_NODE); lin("fc_bbox", 32, 4)
    bn("swinbn", 1024); lin("swinfc", hf, 1024)
    bn("hbn", hf); ln("hln", hf); lin("hfc", hf, hf)
    lin("final_fc", num_classes, hf * 3); bn("final_fc_bn", hf * 3)
    return {prefix + k: v for k, v in P.items()}


# ------------------------------------------------------------------------------------------------ ablation heads
def head_gat_mean_forward(sd, src, dst, batch_num_nodes, node_emb, img_embedding, func_text_embedding, training=False, prefix=""):
    """Multi_DefectModel.forward (GraphModel.py:261-303, the pre-Rs_GCN head): GAT x2 -> MLP -> dgl.mean_nodes -> BN + Linear + ELU,
    concatenated with the image and text branches.  (The h_func branch of :289,:296 feeds nothing and is not restated.)
    Dropouts are identity (eval or rate 0)."""
    P = prefix
    x = F.elu(F.linear(_bn(sd, P + "swinbn.", img_embedding, training), sd[P + "swinfc.weight"], sd[P + "swinfc.bias"]))
    t = F.elu(F.linear(_bn(sd, P + "bn_text.", func_text_embedding, training), sd[P + "fc_text.weight"], sd[P + "fc_text.bias"]))
    h = gat_conv(sd, P + "gat.", node_emb, src, dst)
    h = gat_conv(sd, P + "gat2.", h.reshape(h.shape[0], -1), src, dst)
    h = F.elu(F.linear(h.reshape(h.shape[0], -1), sd[P + "fc.weight"], sd[P + "fc.bias"]))
    for i in range(8):
        h = F.elu(F.linear(h, sd[P + f"hidden.{i}.weight"], sd[P + f"hidden.{i}.bias"]))
    off = [0] + torch.cumsum(torch.as_tensor(batch_num_nodes), 0).tolist()
    hmean = torch.stack([h[off[b]:off[b + 1]].mean(0) for b in range(len(off) - 1)])            # dgl.mean_nodes
    hf = F.elu(F.linear(_bn(sd, P + "hbn.", hmean, training), sd[P + "hfc.weight"], sd[P + "hfc.bias"]))
    allf = torch.cat([x, hf, t], dim=1)
    return F.linear(_bn(sd, P + "final_fc_bn.", allf, training), sd[P + "final_fc.weight"], sd[P + "final_fc.bias"])


def head_nograph_forward(sd, img_embedding, func_text_embedding, training=False, prefix=""):
    """Multi_DefectModel_noGraph.forward (GraphModel.py:345-359): image and text branches only."""
    P = prefix
    x = F.elu(F.linear(_bn(sd, P + "swinbn.", img_embedding, training), sd[P + "swinfc.weight"], sd[P + "swinfc.bias"]))
    t = F.elu(F.linear(_bn(sd, P + "bn_text.", func_text_embedding, training), sd[P + "fc_text.weight"], sd[P + "fc_text.bias"]))
    allf = torch.cat([x, t], dim=1)
    return F.linear(_bn(sd, P + "final_fc_bn.", allf, training), sd[P + "final_fc.weight"], sd[P + "final_fc.bias"])


def head_rq3_forward(sd, pos, gat, gcn, src, dst, batch_num_nodes, node_emb, pos_emb, img_embedding, func_text_embedding, training=False,
                     prefix=""):
    """The RQ3 ablation heads Multi_DefectModel_{000,001,100,110,011}.forward (GraphModel.py:401-430, 486-531, 577-615, 669-718,
    894-947) as one function of the (pos, gat, gcn) switches.  Dropouts are identity."""
    P = prefix
    x = F.elu(F.linear(_bn(sd, P + "swinbn.", img_embedding, training), sd[P + "swinfc.weight"], sd[P + "swinfc.bias"]))
    t = F.elu(F.linear(_bn(sd, P + "bn_text.", func_text_embedding, training), sd[P + "fc_text.weight"], sd[P + "fc_text.bias"]))
    if gat:
        h = gat_conv(sd, P + "gat.", node_emb, src, dst)
        h = gat_conv(sd, P + "gat2.", h.reshape(h.shape[0], -1), src, dst)
        h = F.elu(F.linear(h.reshape(h.shape[0], -1), sd[P + "fc.weight"], sd[P + "fc.bias"]))
        for i in range(8):
            h = F.elu(F.linear(h, sd[P + f"hidden.{i}.weight"], sd[P + f"hidden.{i}.bias"]))
    else:
        h = F.elu(F.linear(node_emb, sd[P + "fconly.weight"], sd[P + "fconly.bias"]))
    if not (pos or gcn):
        off = [0] + torch.cumsum(torch.as_tensor(batch_num_nodes), 0).tolist()
        hmean = torch.stack([h[off[b]:off[b + 1]].mean(0) for b in range(len(off) - 1)])
        hf = F.elu(F.linear(_bn(sd, P + "hbn.", hmean, training), sd[P + "hfc.weight"], sd[P + "hfc.bias"]))
    else:
        h_i = _bn(sd, P + "bn_gat.", unbatch_pad(h, batch_num_nodes), training)
        h_i = F.elu(h_i) if (gcn and gat) else F.elu(F.linear(h_i, sd[P + "fc_gat.weight"], sd[P + "fc_gat.bias"]))
        if pos:
            pos_i = unbatch_pad(pos_emb, batch_num_nodes)
            pos_i = F.elu(F.linear(_bn(sd, P + "bn_bbox.", pos_i, training), sd[P + "fc_bbox.weight"], sd[P + "fc_bbox.bias"]))
            hf = torch.cat([h_i, pos_i], dim=2).mean(dim=1)
        else:
            gg = h_i.permute(0, 2, 1)
            for i in range(1, 9):
                gg, _ = rs_gcn(sd, P + f"Rs_GCN_{i}.", gg, training)
            gg = gg.permute(0, 2, 1)
            gg = gg / gg.pow(2).sum(dim=1, keepdim=True).sqrt()
            hf = gg.mean(dim=1)
    allf = torch.cat([x, hf, t], dim=1)
    return F.linear(_bn(sd, P + "final_fc_bn.", allf, training), sd[P + "final_fc.weight"], sd[P + "final_fc.bias"])


def head_nogat_forward(sd, batch_num_nodes, node_emb, pos_emb, img_embedding, func_text_embedding, training=False, prefix=""):
    """Multi_DefectModel_NOGAT.forward (GraphModel.py:1003-1050): raw node embeddings + positions -> Rs_GCN x8 -> l2norm -> mean."""
    P = prefix
    x = F.elu(F.linear(_bn(sd, P + "swinbn.", img_embedding, training), sd[P + "swinfc.weight"], sd[P + "swinfc.bias"]))
    t = F.elu(F.linear(_bn(sd, P + "bn_text.", func_text_embedding, training), sd[P + "fc_text.weight"], sd[P + "fc_text.bias"]))
    h_i = unbatch_pad(node_emb, batch_num_nodes)
    pos_i = unbatch_pad(pos_emb, batch_num_nodes)
    h_i = F.elu(F.linear(_bn(sd, P + "bn_gat.", h_i, training), sd[P + "fc_gat.weight"], sd[P + "fc_gat.bias"]))
    pos_i = F.elu(F.linear(_bn(sd, P + "bn_bbox.", pos_i, training), sd[P + "fc_bbox.weight"], sd[P + "fc_bbox.bias"]))
    gg = torch.cat([h_i, pos_i], dim=2).permute(0, 2, 1)
    for i in range(1, 9):
        gg, _ = rs_gcn(sd, P + f"Rs_GCN_{i}.", gg, training)
    gg = gg.permute(0, 2, 1)
    gg = gg / gg.pow(2).sum(dim=1, keepdim=True).sqrt()
    allf = torch.cat([x, gg.mean(dim=1), t], dim=1)
    return F.linear(_bn(sd, P + "final_fc_bn.", allf, training), sd[P + "final_fc.weight"], sd[P + "final_fc.bias"])


# ------------------------------------------------------------------------------------------------ the remaining ablation heads
def _img_branch(sd, P, img, training):
    return F.elu(F.linear(_bn(sd, P + "swinbn.", img, training), sd[P + "swinfc.weight"], sd[P + "swinfc.bias"]))


def _text_branch(sd, P, txt, training):
    return F.elu(F.linear(_bn(sd, P + "bn_text.", txt, training), sd[P + "fc_text.weight"], sd[P + "fc_text.bias"]))


def _lin_elu(sd, p, x):
    return F.elu(F.linear(x, sd[p + "weight"], sd[p + "bias"]))


def _gat_mlp(sd, P, x, src, dst):
    """GATConv x2 -> fc + ELU -> 8 hidden + ELU (GraphModel.py:167-177)."""
    h = gat_conv(sd, P + "gat.", x, src, dst)
    h = gat_conv(sd, P + "gat2.", h.reshape(h.shape[0], -1), src, dst)
    h = _lin_elu(sd, P + "fc.", h.reshape(h.shape[0], -1))
    for i in range(8):
        h = _lin_elu(sd, P + f"hidden.{i}.", h)
    return h


def _gcn_l2_mean(sd, P, rows, training):
    """[B, 100, 512] -> 8 x Rs_GCN -> l2norm over the node axis -> mean over nodes (GraphModel.py:189-204)."""
    gg = rows.permute(0, 2, 1)
    for i in range(1, 9):
        gg, _ = rs_gcn(sd, P + f"Rs_GCN_{i}.", gg, training)
    gg = gg.permute(0, 2, 1)
    gg = gg / gg.pow(2).sum(dim=1, keepdim=True).sqrt()
    return gg.mean(dim=1)


def _pad_bn_fc(sd, P, h, pos, bnn, training, fc_pos="fc_bbox."):
    h_i = _lin_elu(sd, P + "fc_gat.", _bn(sd, P + "bn_gat.", unbatch_pad(h, bnn), training))
    pos_i = _lin_elu(sd, P + fc_pos, _bn(sd, P + "bn_bbox.", unbatch_pad(pos, bnn), training))
    return torch.cat([h_i, pos_i], dim=2)


def _final_bn_fc(sd, P, feats, training):
    return F.linear(_bn(sd, P + "final_fc_bn.", feats, training), sd[P + "final_fc.weight"], sd[P + "final_fc.bias"])


def head_gatpos_forward(sd, src, dst, batch_num_nodes, node_emb, pos_emb, img, txt, training=False, prefix=""):
    """Multi_DefectModel_GATPOS.forward (GraphModel.py:773-827)."""
    P = prefix
    x, t = _img_branch(sd, P, img, training), _text_branch(sd, P, txt, training)
    h = torch.cat([_lin_elu(sd, P + "fc_gat.", node_emb), _lin_elu(sd, P + "fc_bbox.", pos_emb)], dim=1)
    h = _gat_mlp(sd, P, h, src, dst)
    h_i = _lin_elu(sd, P + "hfc.", _bn(sd, P + "bn_gat.", unbatch_pad(h, batch_num_nodes), training))
    return _final_bn_fc(sd, P, torch.cat([x, h_i.mean(dim=1), t], dim=1), training)


def head_mlp_gcn_forward(sd, variant, batch_num_nodes, node_emb, pos_emb, img, txt, training=False, prefix=""):
    """Multi_DefectModel_NOGAT2 / NOGAT3 / NOGAT4 .forward (GraphModel.py:1334-1382, 1118-1170, 1226-1274); variant in {2, 3, 4}."""
    P = prefix
    x, t = _img_branch(sd, P, img, training), _text_branch(sd, P, txt, training)
    h = _lin_elu(sd, P + "fconly.", node_emb)
    pos = pos_emb
    if variant == 4:
        h = torch.cat([h, _lin_elu(sd, P + "fc_bbox.", pos_emb)], dim=1)
    for i in range(8):
        h = _lin_elu(sd, P + f"hidden.{i}.", h)
    if variant == 3:
        pos = _lin_elu(sd, P + "fc_bbox.", pos_emb)
        for i in range(8):
            pos = _lin_elu(sd, P + f"pos_hidden.{i}.", pos)
    if variant == 4:
        rows = _lin_elu(sd, P + "fc_gat.", _bn(sd, P + "bn_gat.", unbatch_pad(h, batch_num_nodes), training))
    else:
        rows = _pad_bn_fc(sd, P, h, pos, batch_num_nodes, training, "fc_bbox2." if variant == 3 else "fc_bbox.")
    return _final_bn_fc(sd, P, torch.cat([x, _gcn_l2_mean(sd, P, rows, training), t], dim=1), training)


def _full_graph_branch(sd, P, src, dst, bnn, node_emb, pos_emb, training):
    h = _gat_mlp(sd, P, node_emb, src, dst)
    return _gcn_l2_mean(sd, P, _pad_bn_fc(sd, P, h, pos_emb, bnn, training), training)


def head_noglobalimage_forward(sd, src, dst, batch_num_nodes, node_emb, pos_emb, txt, training=False, prefix=""):
    """Multi_DefectModel_noGlobalImage.forward (new_model.py:147-199): text feature x graph feature."""
    P = prefix
    hf = _full_graph_branch(sd, P, src, dst, batch_num_nodes, node_emb, pos_emb, training)
    return _final_bn_fc(sd, P, _text_branch(sd, P, txt, training) * hf, training)


def head_nofunc_forward(sd, src, dst, batch_num_nodes, node_emb, pos_emb, img, training=False, prefix=""):
    """Multi_DefectModel_noFunc.forward (new_model.py:269-326): image feature ++ graph feature."""
    P = prefix
    hf = _full_graph_branch(sd, P, src, dst, batch_num_nodes, node_emb, pos_emb, training)
    return _final_bn_fc(sd, P, torch.cat([_img_branch(sd, P, img, training), hf], dim=1), training)


def head_single_modality_forward(sd, feat, prefix=""):
    """Multi_DefectModel_Image / _FuncText .forward (MotivationModel.py:105-107, 143-145): final_fc on the raw encoder feature."""
    return F.linear(feat, sd[prefix + "final_fc.weight"], sd[prefix + "final_fc.bias"])


def head_graph_forward(sd, src, dst, batch_num_nodes, node_emb, pos_emb, training=False, prefix=""):
    """Multi_DefectModel_Graph.forward (MotivationModel.py:205-256): the full graph branch, final_fc without BatchNorm."""
    hf = _full_graph_branch(sd, prefix, src, dst, batch_num_nodes, node_emb, pos_emb, training)
    return F.linear(hf, sd[prefix + "final_fc.weight"], sd[prefix + "final_fc.bias"])


def head_graph1_forward(sd, batch_num_nodes, node_emb, training=False, prefix=""):
    """Multi_DefectModel_Graph1.forward (MotivationModel.py:306-348): node MLP -> pad -> bn_gat / fc_gat / ELU -> Rs_GCN x8."""
    P = prefix
    h = _lin_elu(sd, P + "fconly.", node_emb)
    for i in range(8):
        h = _lin_elu(sd, P + f"hidden.{i}.", h)
    rows = _lin_elu(sd, P + "fc_gat.", _bn(sd, P + "bn_gat.", unbatch_pad(h, batch_num_nodes), training))
    return F.linear(_gcn_l2_mean(sd, P, rows, training), sd[P + "final_fc.weight"], sd[P + "final_fc.bias"])


def head_graph2_forward(sd, src, dst, batch_num_nodes, node_emb, training=False, prefix=""):
    """Multi_DefectModel_Graph2.forward (MotivationModel.py:389-426): GAT + MLP -> dgl.mean_nodes -> hbn / hfc / ELU -> final_fc."""
    P = prefix
    h = _gat_mlp(sd, P, node_emb, src, dst)
    off = [0] + torch.cumsum(torch.as_tensor(batch_num_nodes), 0).tolist()
    hmean = torch.stack([h[off[b]:off[b + 1]].mean(0) for b in range(len(off) - 1)])
    hf = _lin_elu(sd, P + "hfc.", _bn(sd, P + "hbn.", hmean, training))
    return F.linear(hf, sd[P + "final_fc.weight"], sd[P + "final_fc.bias"])


def ablation_forward(name, sd, g_src, g_dst, bnn, node_emb, pos_emb, img, txt, training=False, prefix=""):
    """Dispatch by the reference class name (without the ``Multi_DefectModel`` prefix: "", "_noGraph", "_000", ..., "_Graph2")."""
    a = dict(training=training, prefix=prefix)
    rq3 = {"_000": (0, 0, 0), "_001": (0, 0, 1), "_100": (1, 0, 0), "_110": (1, 1, 0), "_011": (0, 1, 1)}
    if name == "":
        return head_gat_mean_forward(sd, g_src, g_dst, bnn, node_emb, img, txt, **a)
    if name == "_noGraph":
        return head_nograph_forward(sd, img, txt, **a)
    if name in rq3:
        return head_rq3_forward(sd, *map(bool, rq3[name]), g_src, g_dst, bnn, node_emb, pos_emb, img, txt, **a)
    if name == "_NOGAT":
        return head_nogat_forward(sd, bnn, node_emb, pos_emb, img, txt, **a)
    if name in ("_NOGAT2", "_NOGAT3", "_NOGAT4"):
        return head_mlp_gcn_forward(sd, int(name[-1]), bnn, node_emb, pos_emb, img, txt, **a)
    if name == "_GATPOS":
        return head_gatpos_forward(sd, g_src, g_dst, bnn, node_emb, pos_emb, img, txt, **a)
    if name == "_noGlobalImage":
        return head_noglobalimage_forward(sd, g_src, g_dst, bnn, node_emb, pos_emb, txt, **a)
    if name == "_noFunc":
        return head_nofunc_forward(sd, g_src, g_dst, bnn, node_emb, pos_emb, img, **a)
    if name == "_Image":
        return head_single_modality_forward(sd, img, prefix)
    if name == "_FuncText":
        return head_single_modality_forward(sd, txt, prefix)
    if name == "_Graph":
        return head_graph_forward(sd, g_src, g_dst, bnn, node_emb, pos_emb, **a)
    if name == "_Graph1":
        return head_graph1_forward(sd, bnn, node_emb, **a)
    if name == "_Graph2":
        return head_graph2_forward(sd, g_src, g_dst, bnn, node_emb, **a)
    raise KeyError(name)


ABLATION_HEADS = {   # reference module -> class-name suffixes
    "GraphModel": ["", "_noGraph", "_000", "_001", "_100", "_110", "_GATPOS", "_011", "_NOGAT", "_NOGAT3", "_NOGAT4", "_NOGAT2"],
    "new_model": ["_noGlobalImage", "_noFunc"],
    "MotivationModel": ["_Image", "_FuncText", "_Graph", "_Graph1", "_Graph2"],
}
