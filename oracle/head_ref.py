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
