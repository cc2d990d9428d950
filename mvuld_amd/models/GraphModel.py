"""Graph / fusion head on libmvuld_hip.so: ``Multi_DefectModel_new_GCN`` with the reference's constructor,
``forward(g, img_embedding, func_text_embedding)`` and state_dict keys (mvuld/models/GraphModel.py:81-211).

Execution differences (results identical up to float rounding):
* ``GATConv`` (dgl 0.8.1, third party) is this file's module over CSR-by-destination kernels;
* ``unbatch_features`` (:30-54, a Python loop over ``dgl.unbatch`` + ``torch.cat``) is one segmented pad kernel;
* the Rs_GCN chain runs on node rows ``[B*100, 512]`` -- no permutes (:190,:200);
* the dead ``h_func`` branch (:172,:177: never reaches the logits) is not computed; ``fconly``, ``ln_text``,
  ``hbn``, ``hln``, ``hfc`` stay as (unused) parameters exactly as in the reference.
``g`` is a ``mvuld_amd.graph.BatchedGraph`` (the dgl stand-in); ``g.ndata['HGATOUTPUT'/'HFGATOUTPUT']`` are set
as in the reference (:180-181).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F   # noqa: F401  (API parity; no torch compute is used on the path)

from .. import hip, ops
from ..hip import call, ptr, dt
from .Rs_GCN import Rs_GCN

_SEED = [0x5EED]


def _next_seed():
    _SEED[0] = (_SEED[0] * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
    return _SEED[0]


# ------------------------------------------------------------------------------------------------ functions
class _LinearActFn(torch.autograd.Function):
    """y = dropout(act(x W^T + b)); act in {None, 'elu'}."""

    @staticmethod
    def forward(ctx, x, w, b, act, out_dtype, p_drop, training):
        hip.require_gpu(x)
        ad = x.dtype
        x2 = x.reshape(-1, x.shape[-1])
        w2 = ops.weight(w, ad).reshape(w.shape[0], -1)
        y = ops.gemm_nt(x2, w2, bias=None if b is None else b.data, epi=hip.EPI_ELU if act == "elu" else hip.EPI_NONE,
                        out_dtype=out_dtype or ad)
        seed = 0
        out = y
        if training and p_drop > 0:
            seed = _next_seed()
            out = ops.dropout(y, p_drop, seed)
        ctx.save_for_backward(x2, y)
        ctx.meta = (w, b, act, p_drop if training else 0.0, seed, x.shape, ad)
        return out.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dout):
        x2, y = ctx.saved_tensors
        w, b, act, p, seed, xshape, ad = ctx.meta
        d = dout.reshape(-1, dout.shape[-1]).contiguous()
        if d.dtype != ad:
            d = ops.cast(d, ad)
        if p > 0:
            d = ops.dropout(d, p, seed)
        if act == "elu":
            d = ops.act_bwd(d, y, 1)
        ops.linear_wgrad(d, x2, w, b)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(d, ops.weight_t(w, ad)).view(xshape)
        return dx, None, None, None, None, None, None


def linear_act(x, w, b, act=None, out_dtype=None, p_drop=0.0, training=False):
    return _LinearActFn.apply(x, w, b, act, out_dtype, p_drop, training)


class _BatchNormFn(torch.autograd.Function):
    """BatchNorm1d over dim 1 of [B,C] or [B,C,F] (channel = dim 1), batch statistics when training."""

    @staticmethod
    def forward(ctx, x, bn, _w):
        # `_w` (= bn.weight) is passed only so that this node exists in the graph when x needs no gradient
        hip.require_gpu(x)
        x = x.contiguous()
        if x.dim() == 2:
            O, C, I = x.shape[0], x.shape[1], 1
            so, sc, si = C, 1, 1
        else:
            O, C, I = x.shape
            so, sc, si = C * I, I, 1
        training = bn.training
        y, sm, sr = ops.batchnorm_fwd(x, bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, O, C, I, so, sc, si,
                                      training, bn.eps, bn.momentum)
        if training:
            bn.num_batches_tracked += 1
        ctx.save_for_backward(x, sm, sr)
        ctx.meta = (bn, (O, C, I, so, sc, si), training)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, sm, sr = ctx.saved_tensors
        bn, lay, training = ctx.meta
        return ops.batchnorm_bwd(dy.contiguous(), x, bn.weight, bn.bias, sm, sr, *lay, training), None, None


class _GATConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, conv, index, training, _w):
        # `_w` (= conv.fc.weight) keeps this node in the graph when x (raw node features) needs no gradient
        hip.require_gpu(x)
        ad = x.dtype
        N = x.shape[0]
        H, O = conv._num_heads, conv._out_feats
        seed, p = 0, (conv.feat_drop_p if training else 0.0)
        xin = x
        if p > 0:
            seed = _next_seed()
            xin = ops.dropout(x, p, seed)
        ft = ops.gemm_nt(xin, ops.weight(conv.fc.weight, ad))                       # [N, H*O]
        el = torch.empty((N, H), dtype=torch.float32, device=x.device)
        er = torch.empty((N, H), dtype=torch.float32, device=x.device)
        call("gat_scores_fwd", ptr(ft), ptr(conv.attn_l), ptr(conv.attn_r), ptr(el), ptr(er), N, H, O, dt(ft))
        E = index["src_by_dst"].numel()
        alpha = torch.empty((max(E, 1), H), dtype=torch.float32, device=x.device)
        out = torch.empty((N, H, O), dtype=ad, device=x.device)
        call("gat_aggregate_fwd", ptr(ft), ptr(el), ptr(er), ptr(index["indptr_dst"]), ptr(index["src_by_dst"]),
             ptr(conv.bias), ptr(out), ptr(alpha), N, E, H, O, conv.negative_slope, dt(ft))
        ctx.save_for_backward(xin, ft, el, er, alpha)
        ctx.meta = (conv, index, p, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        xin, ft, el, er, alpha = ctx.saved_tensors
        conv, index, p, seed = ctx.meta
        ad = xin.dtype
        N = xin.shape[0]
        H, O = conv._num_heads, conv._out_feats
        E = index["src_by_dst"].numel()
        dout = dout.contiguous()
        ops.colsum_into(dout.view(N, H * O), ops.grad_of(conv.bias))
        dft = torch.empty_like(ft)
        dev = xin.device
        ws1 = torch.empty((max(E, 1), H), dtype=torch.float32, device=dev)
        ws2 = torch.empty((N, H), dtype=torch.float32, device=dev)
        ws3 = torch.empty((N, H), dtype=torch.float32, device=dev)
        call("gat_aggregate_bwd", ptr(dout), ptr(ft), ptr(el), ptr(er), ptr(alpha), ptr(conv.attn_l), ptr(conv.attn_r),
             ptr(index["indptr_dst"]), ptr(index["src_by_dst"]), ptr(index["indptr_src"]), ptr(index["dst_by_src"]),
             ptr(index["slot_by_src"]), ptr(dft), ptr(ops.grad_of(conv.attn_l)), ptr(ops.grad_of(conv.attn_r)),
             ptr(ws1), ptr(ws2), ptr(ws3), N, E, H, O, conv.negative_slope, dt(ft))
        ops.linear_wgrad(dft, xin, conv.fc.weight, None)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(dft, ops.weight_t(conv.fc.weight, ad))
            if p > 0:
                dx = ops.dropout(dx, p, seed)
        return dx, None, None, None, None


class _SegmentPadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, offsets, B, maxn):
        h = h.contiguous()
        F_ = h.shape[1]
        out = torch.empty((B, maxn, F_), dtype=h.dtype, device=h.device)
        call("segment_pad_fwd", ptr(h), ptr(offsets), ptr(out), B, maxn, F_, dt(h))
        ctx.save_for_backward(offsets)
        ctx.meta = (B, maxn, F_, h.shape[0])
        return out

    @staticmethod
    def backward(ctx, dout):
        (offsets,) = ctx.saved_tensors
        B, maxn, F_, ntot = ctx.meta
        dh = torch.empty((ntot, F_), dtype=dout.dtype, device=dout.device)
        dout = dout.contiguous()
        call("segment_pad_bwd", ptr(dout), ptr(offsets), ptr(dh), B, maxn, F_, ntot, dt(dh))
        return dh, None, None, None


class _L2NormMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, B):
        """g [B*Nn, C] node rows -> [B, C]."""
        C = g.shape[1]
        Nn = g.shape[0] // B
        hf = torch.empty((B, C), dtype=g.dtype, device=g.device)
        ssum = torch.empty((B, C), dtype=torch.float32, device=g.device)
        snrm = torch.empty((B, C), dtype=torch.float32, device=g.device)
        call("l2norm_mean_fwd", ptr(g), ptr(hf), ptr(ssum), ptr(snrm), B, Nn, C, dt(g))
        ctx.save_for_backward(g, ssum, snrm)
        ctx.meta = (B, Nn, C)
        return hf

    @staticmethod
    def backward(ctx, dhf):
        g, ssum, snrm = ctx.saved_tensors
        B, Nn, C = ctx.meta
        dg = torch.empty_like(g)
        dhf = dhf.contiguous()
        call("l2norm_mean_bwd", ptr(g), ptr(dhf), ptr(ssum), ptr(snrm), ptr(dg), B, Nn, C, dt(g))
        return dg, None


class _CastFn(torch.autograd.Function):
    """storage-dtype change that autograd can see (bf16 encoders -> fp32 tail and back)."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return ops.cast(x.contiguous(), dtype)

    @staticmethod
    def backward(ctx, d):
        return ops.cast(d.contiguous(), ctx.src), None


def cast_to(x, dtype):
    return x if x.dtype == dtype else _CastFn.apply(x, dtype)


class _ConcatColsFn(torch.autograd.Function):
    """cat along the last dim through device copies (memory movement only)."""

    @staticmethod
    def forward(ctx, *xs):
        widths = [x.shape[-1] for x in xs]
        out = torch.empty(xs[0].shape[:-1] + (sum(widths),), dtype=xs[0].dtype, device=xs[0].device)
        o = 0
        for x, w in zip(xs, widths):
            out[..., o:o + w].copy_(x)
            o += w
        ctx.widths = widths
        return out

    @staticmethod
    def backward(ctx, d):
        outs, o = [], 0
        for w in ctx.widths:
            outs.append(d[..., o:o + w].contiguous())
            o += w
        return tuple(outs)


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, loss_scale):
        hip.require_gpu(logits)
        assert logits.dtype == torch.float32
        B, K = logits.shape
        loss = torch.zeros((), dtype=torch.float32, device=logits.device)
        probs = torch.empty_like(logits)
        dlog = torch.empty_like(logits)
        logits, target = logits.contiguous(), target.contiguous()
        call("cross_entropy", ptr(logits), ptr(target), ptr(loss), ptr(probs), ptr(dlog), B, K, float(loss_scale))
        ctx.save_for_backward(dlog)
        ctx.mark_non_differentiable(probs)
        return loss, probs

    @staticmethod
    def backward(ctx, dloss, _dp):
        # dlog was formed in forward for an upstream factor of 1; whatever autograd hands down (1 for loss.backward(), k for
        # (k * loss).backward(), ...) is applied on the device: no host sync, no torch arithmetic
        (dlog,) = ctx.saved_tensors
        out = torch.empty_like(dlog)
        dloss = dloss.to(torch.float32).contiguous()
        call("scale_by_dev", ptr(dlog), ptr(dloss), ptr(out), dlog.numel())
        return out, None, None


class _SoftCrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, smoothing, loss_scale):
        hip.require_gpu(logits)
        assert logits.dtype == torch.float32
        B, K = logits.shape
        loss = torch.zeros((), dtype=torch.float32, device=logits.device)
        probs, dlog = torch.empty_like(logits), torch.empty_like(logits)
        logits, target = logits.contiguous(), target.contiguous()
        soft = target.dtype == torch.float32
        assert (soft and target.shape == logits.shape) or (not soft and target.dtype == torch.int64 and target.shape == (B,))
        call("cross_entropy_soft", ptr(logits), ptr(target) if soft else None, None if soft else ptr(target), float(smoothing), ptr(loss), ptr(probs),
             ptr(dlog), B, K, float(loss_scale))
        ctx.save_for_backward(dlog)
        ctx.mark_non_differentiable(probs)
        return loss, probs

    @staticmethod
    def backward(ctx, dloss, _dp):
        (dlog,) = ctx.saved_tensors
        out = torch.empty_like(dlog)
        call("scale_by_dev", ptr(dlog), ptr(dloss.to(torch.float32).contiguous()), ptr(out), dlog.numel())
        return out, None, None, None


def soft_target_cross_entropy(logits, target, loss_scale=1.0):
    """timm SoftTargetCrossEntropy (main.py:136-138): mean_b sum_k -target[b,k] * log_softmax(logits)[b,k]; target [B, K] fp32."""
    return _SoftCrossEntropyFn.apply(logits, target, 0.0, loss_scale)


def label_smoothing_cross_entropy(logits, target, smoothing=0.1, loss_scale=1.0):
    """timm LabelSmoothingCrossEntropy (main.py:139-140): (1 - smoothing) * nll + smoothing * mean_k(-log_softmax); target [B] int64."""
    return _SoftCrossEntropyFn.apply(logits, target, smoothing, loss_scale)


def cross_entropy(logits, target, loss_scale=1.0):
    """(mean CE * loss_scale, softmax probs): CrossEntropyLoss + F.softmax of main_bigvul.py:298,330-333."""
    return _CrossEntropyFn.apply(logits, target, loss_scale)


# ------------------------------------------------------------------------------------------------ modules
class GATConv(nn.Module):
    """dgl.nn.pytorch.GATConv(in_feats, out_feats, num_heads, feat_drop) with dgl's parameter names."""

    def __init__(self, in_feats, out_feats, num_heads, feat_drop=0.0, attn_drop=0.0, negative_slope=0.2, residual=False,
                 activation=None, allow_zero_in_degree=False, bias=True):
        super().__init__()
        assert attn_drop == 0.0 and not residual and activation is None and bias
        self._in_feats, self._out_feats, self._num_heads = in_feats, out_feats, num_heads
        self.negative_slope = negative_slope
        self.feat_drop_p = float(feat_drop)
        self.fc = nn.Linear(in_feats, out_feats * num_heads, bias=False)
        self.attn_l = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.attn_r = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.bias = nn.Parameter(torch.zeros(num_heads * out_feats))
        gain = nn.init.calculate_gain("relu")
        nn.init.xavier_normal_(self.fc.weight, gain=gain)
        nn.init.xavier_normal_(self.attn_l, gain=gain)
        nn.init.xavier_normal_(self.attn_r, gain=gain)

    def forward(self, graph, feat):
        return _GATConvFn.apply(feat, self, graph.index(), self.training, self.fc.weight)


def l2norm(X):
    """L2-normalise over dim 1 (reference helper, GraphModel.py:74-79); fused with the mean on the hot path."""
    raise RuntimeError("l2norm is fused into _L2NormMeanFn on the hot path")


def batch_norm(x, bn):
    return _BatchNormFn.apply(x, bn, bn.weight)


class Multi_DefectModel_new_GCN(nn.Module):
    '''best model (reference GraphModel.py:81-211)'''

    def __init__(self, config, pretrained=True, attention=True, act_dtype=torch.bfloat16):
        super().__init__()
        self.num_features = 1024
        self.config = config
        self.num_classes = config.MODEL.NUM_CLASSES
        self.act_dtype = act_dtype
        self.tail_fp32 = True           # eps-free l2norm over nodes, final BatchNorm + classifier: fp32 storage
        self.chain_fp32 = os.environ.get("MVULD_CHAIN_FP32", "1") == "1"   # the 8 Rs_GCN blocks
        hfeat, embfeat, numheads = 512, 768, 4
        self.p_gat, self.p_mlp, self.p_hidden = 0.2, 0.2, 0.2
        self.gat = GATConv(in_feats=embfeat, out_feats=hfeat, num_heads=numheads, feat_drop=self.p_gat)
        self.gat2 = GATConv(in_feats=hfeat * numheads, out_feats=hfeat, num_heads=numheads, feat_drop=self.p_gat)
        self.fc = nn.Linear(hfeat * numheads, hfeat)
        self.fconly = nn.Linear(embfeat, hfeat)
        self.hidden = nn.ModuleList([nn.Linear(hfeat, hfeat) for _ in range(8)])
        for i in range(1, 9):
            setattr(self, f"Rs_GCN_{i}", Rs_GCN(in_channels=512, inter_channels=512))
        self.bn_text = nn.BatchNorm1d(embfeat)
        self.ln_text = nn.LayerNorm(embfeat)
        self.fc_text = nn.Linear(embfeat, hfeat)
        self.max_node = 100
        self.bn_gat = nn.BatchNorm1d(self.max_node)
        self.fc_gat = nn.Linear(512, 480)
        self.bn_bbox = nn.BatchNorm1d(self.max_node)
        self.fc_bbox = nn.Linear(4, 32)
        self.swinbn = nn.BatchNorm1d(self.num_features)
        self.swinfc = nn.Linear(self.num_features, hfeat)
        self.hbn = nn.BatchNorm1d(hfeat)
        self.hln = nn.LayerNorm(hfeat)
        self.hfc = nn.Linear(hfeat, hfeat)
        self.final_fc = nn.Linear(hfeat * 3, self.num_classes)
        self.final_fc_bn = nn.BatchNorm1d(hfeat * 3)
        # parameters the reference constructs but never uses in forward (hence find_unused_parameters=True there)
        self.unused_parameter_prefixes = ("fconly.", "ln_text.", "hbn.", "hln.", "hfc.")

    def forward(self, g, img_embedding, func_text_embedding):
        """Reference signature (GraphModel.py:151-211).  The graph branch depends on neither encoder output, so the fused model
        may run it early on another stream: forward == forward_join(img, text, forward_graph(g))."""
        return self.forward_join(g, img_embedding, func_text_embedding, self.forward_graph(g))

    def forward_graph(self, g):
        """Graph branch (:163-204): GAT x2 -> MLP -> unbatch/pad -> Rs_GCN x8 -> l2norm over nodes + mean  => [B, 512]."""
        ad = self.act_dtype
        tr = self.training
        # bf16 mode: the fp32 tail's GEMMs run as 3-term bf16 splits on the matrix cores (error ~2^-16); the fp32 parity
        # mode keeps exact fp32 FMA GEMMs.  (Set here so that this step's backward sees the same choice.)
        ops.USE_SPLIT3[0] = (ad == torch.bfloat16)
        B = g.batch_size
        h = g.ndata["_UNIX_NODE_EMB"]
        bboxes = g.ndata["pos_emb"]
        hip.require_gpu(h, bboxes)
        h = ops.cast(h.contiguous(), ad) if h.dtype != ad else h
        bboxes = ops.cast(bboxes.contiguous(), ad) if bboxes.dtype != ad else bboxes
        h = self.gat(g, h).view(h.shape[0], -1)
        h = self.gat2(g, h).view(h.shape[0], -1)
        h = linear_act(h, self.fc.weight, self.fc.bias, "elu", None, self.p_mlp, tr)
        for hl in self.hidden:
            h = linear_act(h, hl.weight, hl.bias, "elu", None, self.p_hidden, tr)
        g.ndata['HGATOUTPUT'] = h
        g.ndata['HFGATOUTPUT'] = bboxes
        # unbatch + pad/truncate to 100 nodes  (:182-184)
        off = g.index()["node_offsets"]
        h_i = _SegmentPadFn.apply(h, off, B, self.max_node)                      # [B,100,512]
        pos_i = _SegmentPadFn.apply(bboxes, off, B, self.max_node)               # [B,100,4]
        h_i = linear_act(batch_norm(h_i, self.bn_gat), self.fc_gat.weight, self.fc_gat.bias, "elu")    # [B,100,480]
        pos_i = linear_act(batch_norm(pos_i, self.bn_bbox), self.fc_bbox.weight, self.fc_bbox.bias, "elu")  # [B,100,32]
        v = _ConcatColsFn.apply(h_i, pos_i).view(B * self.max_node, 512)          # node rows; no permute needed
        # The 8-block Rs_GCN chain, the eps-free l2norm over nodes and the final BatchNorm are <1 % of the FLOPs but
        # numerically the touchiest part of the model (a residual chain feeding a difference of near-equal terms):
        # they run with fp32 storage even when the encoders run in bf16.
        tail = torch.float32 if self.tail_fp32 else ad
        chain = torch.float32 if self.chain_fp32 else ad
        v = cast_to(v, chain)
        for i in range(1, 9):
            v, _ = getattr(self, f"Rs_GCN_{i}").forward_rows(v, B)
        v = cast_to(v, tail)
        return _L2NormMeanFn.apply(v, B)                                          # l2norm over nodes + mean (:201-204)

    def forward_join(self, g, img_embedding, func_text_embedding, h_feature):
        """Image branch (:153-154), text branch (:158-159), concat + final BatchNorm + classifier (:206-210)."""
        ad = self.act_dtype
        hip.require_gpu(img_embedding, func_text_embedding)
        ops.USE_SPLIT3[0] = (ad == torch.bfloat16)
        img_embedding = ops.cast(img_embedding.contiguous(), ad) if img_embedding.dtype != ad else img_embedding
        func_text_embedding = ops.cast(func_text_embedding.contiguous(), ad) if func_text_embedding.dtype != ad else func_text_embedding
        x = linear_act(batch_norm(img_embedding, self.swinbn), self.swinfc.weight, self.swinfc.bias, "elu")
        t = linear_act(batch_norm(func_text_embedding, self.bn_text), self.fc_text.weight, self.fc_text.bias, "elu")
        tail = torch.float32 if self.tail_fp32 else ad
        all_feats = _ConcatColsFn.apply(cast_to(x, tail), h_feature, cast_to(t, tail))
        return linear_act(batch_norm(all_feats, self.final_fc_bn), self.final_fc.weight, self.final_fc.bias,
                          None, torch.float32)


# ------------------------------------------------------------------------------------------------ ablation heads
class _MeanNodesFn(torch.autograd.Function):
    """dgl.mean_nodes(g, "h") (GraphModel.py:299): per-graph mean of the node rows, [sum N, C] -> [B, C]."""

    @staticmethod
    def forward(ctx, h, off, B):
        out = torch.empty((B, h.shape[1]), dtype=h.dtype, device=h.device)
        call("segment_mean_fwd", ptr(h), ptr(off), ptr(out), B, h.shape[1], dt(h))
        ctx.save_for_backward(off)
        ctx.dims = (h.shape[0], B, h.shape[1])
        return out

    @staticmethod
    def backward(ctx, dout):
        (off,) = ctx.saved_tensors
        T, B, C = ctx.dims
        dx = torch.empty((T, C), dtype=dout.dtype, device=dout.device)
        call("segment_mean_bwd", ptr(dout.contiguous()), ptr(off), ptr(dx), B, C, dt(dx))
        return dx, None, None


class Multi_DefectModel(nn.Module):
    """The pre-Rs_GCN head (reference GraphModel.py:214-303): GAT x2 -> Linear+ELU -> 8 hidden Linear+ELU -> dgl.mean_nodes ->
    BatchNorm + Linear + ELU, concatenated with the image and text branches.  Same constructor, forward signature and state-dict
    keys; the reference's dead h_func branch (:289,:296) is not computed."""

    def __init__(self, config, pretrained=True, attention=True, act_dtype=torch.bfloat16):
        super().__init__()
        self.num_features, self.config, self.num_classes, self.act_dtype = 1024, config, config.MODEL.NUM_CLASSES, act_dtype
        hfeat, embfeat, numheads = 512, 768, 4
        self.p_gat = self.p_mlp = self.p_hidden = 0.1
        self.gat = GATConv(in_feats=embfeat, out_feats=hfeat, num_heads=numheads, feat_drop=self.p_gat)
        self.gat2 = GATConv(in_feats=hfeat * numheads, out_feats=hfeat, num_heads=numheads, feat_drop=self.p_gat)
        self.fc = nn.Linear(hfeat * numheads, hfeat)
        self.fconly = nn.Linear(embfeat, hfeat)
        self.hidden = nn.ModuleList([nn.Linear(hfeat, hfeat) for _ in range(8)])
        self.bn_text = nn.BatchNorm1d(embfeat)
        self.fc_text = nn.Linear(embfeat, hfeat)
        self.swinbn = nn.BatchNorm1d(self.num_features)
        self.swinfc = nn.Linear(self.num_features, hfeat)
        self.hbn = nn.BatchNorm1d(hfeat)
        self.hfc = nn.Linear(hfeat, hfeat)
        self.final_fc = nn.Linear(hfeat * 3, self.num_classes)
        self.final_fc_bn = nn.BatchNorm1d(hfeat * 3)
        self.unused_parameter_prefixes = ("fconly.",)

    def forward(self, g, img_embedding, func_text_embedding):
        ad, tr = self.act_dtype, self.training
        ops.USE_SPLIT3[0] = False
        h = g.ndata["_UNIX_NODE_EMB"]
        hip.require_gpu(h, img_embedding, func_text_embedding)
        cast = lambda v: ops.cast(v.contiguous(), ad) if v.dtype != ad else v
        x = linear_act(batch_norm(cast(img_embedding), self.swinbn), self.swinfc.weight, self.swinfc.bias, "elu")
        t = linear_act(batch_norm(cast(func_text_embedding), self.bn_text), self.fc_text.weight, self.fc_text.bias, "elu")
        h = cast(h)
        h = self.gat(g, h).view(h.shape[0], -1)
        h = self.gat2(g, h).view(h.shape[0], -1)
        h = linear_act(h, self.fc.weight, self.fc.bias, "elu", None, self.p_mlp, tr)
        for hl in self.hidden:
            h = linear_act(h, hl.weight, hl.bias, "elu", None, self.p_hidden, tr)
        # per-graph tail in fp32 storage, like the main head's: train-mode BatchNorm over a handful of graphs divides by a tiny batch
        # deviation and would amplify bf16 rounding of its inputs
        hmean = cast_to(_MeanNodesFn.apply(h, g.index()["node_offsets"], g.batch_size), torch.float32)
        hf = linear_act(batch_norm(hmean, self.hbn), self.hfc.weight, self.hfc.bias, "elu")
        all_feats = _ConcatColsFn.apply(cast_to(x, torch.float32), hf, cast_to(t, torch.float32))
        return linear_act(batch_norm(all_feats, self.final_fc_bn), self.final_fc.weight, self.final_fc.bias, None, torch.float32)


class Multi_DefectModel_noGraph(nn.Module):
    """Image + text ablation (reference GraphModel.py:306-359): no graph branch at all.  The reference constructs fconly / hidden /
    ln_text / hbn / hln / hfc and never uses them; they are kept for state-dict compatibility."""

    def __init__(self, config, pretrained=True, attention=True, act_dtype=torch.bfloat16):
        super().__init__()
        self.num_features, self.config, self.num_classes, self.act_dtype = 1024, config, config.MODEL.NUM_CLASSES, act_dtype
        hfeat, embfeat = 512, 768
        self.fconly = nn.Linear(embfeat, hfeat)
        self.hidden = nn.ModuleList([nn.Linear(hfeat, hfeat) for _ in range(8)])
        self.bn_text = nn.BatchNorm1d(embfeat)
        self.ln_text = nn.LayerNorm(embfeat)
        self.fc_text = nn.Linear(embfeat, hfeat)
        self.swinbn = nn.BatchNorm1d(self.num_features)
        self.swinfc = nn.Linear(self.num_features, hfeat)
        self.hbn = nn.BatchNorm1d(hfeat)
        self.hln = nn.LayerNorm(hfeat)
        self.hfc = nn.Linear(hfeat, hfeat)
        self.final_fc = nn.Linear(hfeat * 2, self.num_classes)
        self.final_fc_bn = nn.BatchNorm1d(hfeat * 2)
        self.unused_parameter_prefixes = ("fconly.", "hidden.", "ln_text.", "hbn.", "hln.", "hfc.")

    def forward(self, g, img_embedding, func_text_embedding):
        ad = self.act_dtype
        ops.USE_SPLIT3[0] = False
        hip.require_gpu(img_embedding, func_text_embedding)
        cast = lambda v: ops.cast(v.contiguous(), ad) if v.dtype != ad else v
        x = linear_act(batch_norm(cast(img_embedding), self.swinbn), self.swinfc.weight, self.swinfc.bias, "elu")
        t = linear_act(batch_norm(cast(func_text_embedding), self.bn_text), self.fc_text.weight, self.fc_text.bias, "elu")
        return linear_act(batch_norm(_ConcatColsFn.apply(x, t), self.final_fc_bn), self.final_fc.weight, self.final_fc.bias, None,
                          torch.float32)


class _EluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.empty_like(x)
        call("elu_fwd", ptr(x), ptr(y), x.numel(), dt(x))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.act_bwd(dy.contiguous(), y, 1)


class _RQ3Head(nn.Module):
    """The reference's RQ3 ablation family (GraphModel.py:362-949), one class per (pos, gat, gcn) switch triple, all on the kernels of
    the full head (= 111, Multi_DefectModel_new_GCN):
      node features   gat: GATConv x2 -> fc + ELU -> 8 hidden Linear + ELU;  else: fconly + ELU
      read-out        neither pos nor gcn: dgl.mean_nodes -> hbn -> hfc -> ELU                                  (000)
                      pos, no gcn: pad to 100 nodes, bn_gat / fc_gat(512->480) / ELU and bn_bbox / fc_bbox(4->32) / ELU, concat,
                                   mean over the 100 slots                                                      (100, 110)
                      gcn, no pos: pad to 100 nodes, bn_gat (-> fc_gat 512->512 when there is no GAT) -> ELU, 8 x Rs_GCN,
                                   l2norm over nodes, mean                                                      (001, 011)
    Each subclass constructs exactly the parameters its reference class constructs (state-dict parity), used or not."""
    POS = GAT = GCN = False
    P_DROP = 0.2

    def __init__(self, config, pretrained=True, attention=True, act_dtype=torch.bfloat16):
        super().__init__()
        self.num_features, self.config, self.num_classes, self.act_dtype = 1024, config, config.MODEL.NUM_CLASSES, act_dtype
        hfeat, embfeat, numheads = 512, 768, 4
        self.p_gat = self.p_mlp = self.p_hidden = self.P_DROP
        self.chain_fp32 = os.environ.get("MVULD_CHAIN_FP32", "1") == "1"
        unused = ["hidden."] if not self.GAT else ["fconly."]
        if self.GAT:
            self.gat = GATConv(in_feats=embfeat, out_feats=hfeat, num_heads=numheads, feat_drop=self.p_gat)
            self.gat2 = GATConv(in_feats=hfeat * numheads, out_feats=hfeat, num_heads=numheads, feat_drop=self.p_gat)
            self.fc = nn.Linear(hfeat * numheads, hfeat)
        self.fconly = nn.Linear(embfeat, hfeat)
        self.hidden = nn.ModuleList([nn.Linear(hfeat, hfeat) for _ in range(8)])
        if self.GCN:
            for i in range(1, 9):
                setattr(self, f"Rs_GCN_{i}", Rs_GCN(in_channels=512, inter_channels=512))
        self.bn_text = nn.BatchNorm1d(embfeat)
        if not (self.POS and self.GAT):                      # 110 has no ln_text / hbn / hln / hfc
            self.ln_text = nn.LayerNorm(embfeat)
            unused.append("ln_text.")
        self.fc_text = nn.Linear(embfeat, hfeat)
        self.max_node = 100
        if self.POS or self.GCN:
            self.bn_gat = nn.BatchNorm1d(self.max_node)
            self.fc_gat = nn.Linear(512, 480 if self.POS else 512)
            if self.GCN and self.GAT:
                unused.append("fc_gat.")                     # 011 applies bn_gat + ELU only (:902)
        if self.POS:
            self.bn_bbox = nn.BatchNorm1d(self.max_node)
            self.fc_bbox = nn.Linear(4, 32)
        self.swinbn = nn.BatchNorm1d(self.num_features)
        self.swinfc = nn.Linear(self.num_features, hfeat)
        if not (self.POS and self.GAT):
            self.hbn = nn.BatchNorm1d(hfeat)
            self.hln = nn.LayerNorm(hfeat)
            self.hfc = nn.Linear(hfeat, hfeat)
            unused.append("hln.")
            if self.POS or self.GCN:
                unused += ["hbn.", "hfc."]
        self.final_fc = nn.Linear(hfeat * 3, self.num_classes)
        self.final_fc_bn = nn.BatchNorm1d(hfeat * 3)
        self.unused_parameter_prefixes = tuple(unused)

    def forward(self, g, img_embedding, func_text_embedding):
        ad, tr, B = self.act_dtype, self.training, g.batch_size
        ops.USE_SPLIT3[0] = self.GCN and ad == torch.bfloat16
        h = g.ndata["_UNIX_NODE_EMB"]
        hip.require_gpu(h, img_embedding, func_text_embedding)
        cast = lambda v: ops.cast(v.contiguous(), ad) if v.dtype != ad else v
        x = linear_act(batch_norm(cast(img_embedding), self.swinbn), self.swinfc.weight, self.swinfc.bias, "elu")
        t = linear_act(batch_norm(cast(func_text_embedding), self.bn_text), self.fc_text.weight, self.fc_text.bias, "elu")
        h = cast(h)
        if self.GAT:
            h = self.gat(g, h).view(h.shape[0], -1)
            h = self.gat2(g, h).view(h.shape[0], -1)
            h = linear_act(h, self.fc.weight, self.fc.bias, "elu", None, self.p_mlp, tr)
            for hl in self.hidden:
                h = linear_act(h, hl.weight, hl.bias, "elu", None, self.p_hidden, tr)
        else:
            h = linear_act(h, self.fconly.weight, self.fconly.bias, "elu", None, self.p_mlp, tr)
        off = g.index()["node_offsets"]
        if not (self.POS or self.GCN):
            hmean = cast_to(_MeanNodesFn.apply(h, off, B), torch.float32)
            hf = linear_act(batch_norm(hmean, self.hbn), self.hfc.weight, self.hfc.bias, "elu")
        else:
            g.ndata['HGATOUTPUT'] = h
            g.ndata['HFGATOUTPUT'] = g.ndata["pos_emb"]
            h_i = batch_norm(_SegmentPadFn.apply(h, off, B, self.max_node), self.bn_gat)                  # [B,100,512]
            if self.GCN and self.GAT:
                h_i = _EluFn.apply(h_i)
            else:
                h_i = linear_act(h_i, self.fc_gat.weight, self.fc_gat.bias, "elu")
            if self.POS:
                pos_i = _SegmentPadFn.apply(cast(g.ndata["pos_emb"]), off, B, self.max_node)
                pos_i = linear_act(batch_norm(pos_i, self.bn_bbox), self.fc_bbox.weight, self.fc_bbox.bias, "elu")
                v = _ConcatColsFn.apply(h_i, pos_i).view(B * self.max_node, 512)
                slots = (torch.arange(B + 1, dtype=torch.int32) * self.max_node).to(v.device)
                hf = cast_to(_MeanNodesFn.apply(v, slots, B), torch.float32)                               # torch.mean(dim=1)
            else:
                v = cast_to(h_i.reshape(B * self.max_node, 512), torch.float32 if self.chain_fp32 else ad)
                for i in range(1, 9):
                    v, _ = getattr(self, f"Rs_GCN_{i}").forward_rows(v, B)
                hf = _L2NormMeanFn.apply(cast_to(v, torch.float32), B)
        all_feats = _ConcatColsFn.apply(cast_to(x, torch.float32), hf, cast_to(t, torch.float32))
        return linear_act(batch_norm(all_feats, self.final_fc_bn), self.final_fc.weight, self.final_fc.bias, None, torch.float32)


class Multi_DefectModel_000(_RQ3Head):
    """MLP + mean_nodes (reference GraphModel.py:362-430)."""


class Multi_DefectModel_001(_RQ3Head):
    """GCN only (:433-531)."""
    GCN = True


class Multi_DefectModel_100(_RQ3Head):
    """Positional features only (:534-615)."""
    POS = True


class Multi_DefectModel_110(_RQ3Head):
    """Positional features + GAT (:618-718)."""
    POS = GAT = True
    P_DROP = 0.1


class Multi_DefectModel_011(_RQ3Head):
    """GAT + GCN (:830-947)."""
    GAT = GCN = True


class Multi_DefectModel_NOGAT(nn.Module):
    """Node embeddings straight into the Rs_GCN chain, no GAT / MLP (reference GraphModel.py:950-1050): pad to 100 nodes,
    bn_gat -> fc_gat(768 -> 480) -> ELU and bn_bbox -> fc_bbox(4 -> 32) -> ELU, concat, 8 x Rs_GCN, l2norm over nodes, mean."""

    def __init__(self, config, pretrained=True, attention=True, act_dtype=torch.bfloat16):
        super().__init__()
        self.num_features, self.config, self.num_classes, self.act_dtype = 1024, config, config.MODEL.NUM_CLASSES, act_dtype
        hfeat, embfeat = 512, 768
        self.chain_fp32 = os.environ.get("MVULD_CHAIN_FP32", "1") == "1"
        for i in range(1, 9):
            setattr(self, f"Rs_GCN_{i}", Rs_GCN(in_channels=512, inter_channels=512))
        self.bn_text = nn.BatchNorm1d(embfeat)
        self.ln_text = nn.LayerNorm(embfeat)
        self.fc_text = nn.Linear(embfeat, hfeat)
        self.max_node = 100
        self.bn_gat = nn.BatchNorm1d(self.max_node)
        self.fc_gat = nn.Linear(768, 480)
        self.bn_bbox = nn.BatchNorm1d(self.max_node)
        self.fc_bbox = nn.Linear(4, 32)
        self.swinbn = nn.BatchNorm1d(self.num_features)
        self.swinfc = nn.Linear(self.num_features, hfeat)
        self.hbn = nn.BatchNorm1d(hfeat)
        self.hln = nn.LayerNorm(hfeat)
        self.hfc = nn.Linear(hfeat, hfeat)
        self.final_fc = nn.Linear(hfeat * 3, self.num_classes)
        self.final_fc_bn = nn.BatchNorm1d(hfeat * 3)
        self.unused_parameter_prefixes = ("ln_text.", "hbn.", "hln.", "hfc.")

    def forward(self, g, img_embedding, func_text_embedding):
        ad, B = self.act_dtype, g.batch_size
        ops.USE_SPLIT3[0] = ad == torch.bfloat16
        h = g.ndata["_UNIX_NODE_EMB"]
        hip.require_gpu(h, img_embedding, func_text_embedding)
        cast = lambda v: ops.cast(v.contiguous(), ad) if v.dtype != ad else v
        x = linear_act(batch_norm(cast(img_embedding), self.swinbn), self.swinfc.weight, self.swinfc.bias, "elu")
        t = linear_act(batch_norm(cast(func_text_embedding), self.bn_text), self.fc_text.weight, self.fc_text.bias, "elu")
        g.ndata['HGATOUTPUT'] = h
        g.ndata['HFGATOUTPUT'] = g.ndata["pos_emb"]
        off = g.index()["node_offsets"]
        h_i = _SegmentPadFn.apply(cast(h), off, B, self.max_node)                                         # [B,100,768]
        pos_i = _SegmentPadFn.apply(cast(g.ndata["pos_emb"]), off, B, self.max_node)                      # [B,100,4]
        h_i = linear_act(batch_norm(h_i, self.bn_gat), self.fc_gat.weight, self.fc_gat.bias, "elu")       # [B,100,480]
        pos_i = linear_act(batch_norm(pos_i, self.bn_bbox), self.fc_bbox.weight, self.fc_bbox.bias, "elu")
        v = cast_to(_ConcatColsFn.apply(h_i, pos_i).view(B * self.max_node, 512), torch.float32 if self.chain_fp32 else ad)
        for i in range(1, 9):
            v, _ = getattr(self, f"Rs_GCN_{i}").forward_rows(v, B)
        hf = _L2NormMeanFn.apply(cast_to(v, torch.float32), B)
        all_feats = _ConcatColsFn.apply(cast_to(x, torch.float32), hf, cast_to(t, torch.float32))
        return linear_act(batch_norm(all_feats, self.final_fc_bn), self.final_fc.weight, self.final_fc.bias, None, torch.float32)


# ------------------------------------------------------------------------------------------------ building blocks of the remaining ablation heads
def _cast_in(v, ad):
    return ops.cast(v.contiguous(), ad) if v.dtype != ad else v


def _feature_branch(v, bn, fc, ad):
    """BatchNorm1d -> Linear -> ELU of an encoder feature [B, C] (GraphModel.py:153-159)."""
    return linear_act(batch_norm(_cast_in(v, ad), bn), fc.weight, fc.bias, "elu")


def _gat_node_features(m, g, h, tr):
    """GATConv x2 -> fc + ELU + dropout (GraphModel.py:167-171)."""
    h = m.gat(g, h).view(h.shape[0], -1)
    h = m.gat2(g, h).view(h.shape[0], -1)
    return linear_act(h, m.fc.weight, m.fc.bias, "elu", None, m.p_mlp, tr)


def _hidden_stack(layers, h, p, tr):
    for hl in layers:
        h = linear_act(h, hl.weight, hl.bias, "elu", None, p, tr)
    return h


def _gcn_readout(m, v, B):
    """8 x Rs_GCN on node rows [B*100, 512] -> l2norm over nodes -> mean over the 100 slots (GraphModel.py:189-204)."""
    v = cast_to(v, torch.float32 if m.chain_fp32 else m.act_dtype)
    for i in range(1, 9):
        v, _ = getattr(m, f"Rs_GCN_{i}").forward_rows(v, B)
    return _L2NormMeanFn.apply(cast_to(v, torch.float32), B)


def _padded_gcn_input(m, g, h, pos, B):
    """pad to 100 nodes; bn_gat -> fc_gat -> ELU and bn_bbox -> fc_bbox -> ELU; concat -> node rows (GraphModel.py:182-188)."""
    off = g.index()["node_offsets"]
    h_i = _SegmentPadFn.apply(h, off, B, m.max_node)
    pos_i = _SegmentPadFn.apply(pos, off, B, m.max_node)
    h_i = linear_act(batch_norm(h_i, m.bn_gat), m.fc_gat.weight, m.fc_gat.bias, "elu")
    fcb = m.fc_bbox2 if hasattr(m, "fc_bbox2") else m.fc_bbox
    pos_i = linear_act(batch_norm(pos_i, m.bn_bbox), fcb.weight, fcb.bias, "elu")
    return _ConcatColsFn.apply(h_i, pos_i).view(B * m.max_node, 512)


def _bn_classifier(m, feats):
    return linear_act(batch_norm(feats, m.final_fc_bn), m.final_fc.weight, m.final_fc.bias, None, torch.float32)


class _MulFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return ops.mul(a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        dy = dy.contiguous()
        return ops.mul(dy, b), ops.mul(dy, a)


def _head_common(m, config, act_dtype, p):
    m.num_features, m.config, m.num_classes, m.act_dtype = 1024, config, config.MODEL.NUM_CLASSES, act_dtype
    m.p_gat = m.p_mlp = m.p_hidden = p
    m.max_node = 100
    m.chain_fp32 = os.environ.get("MVULD_CHAIN_FP32", "1") == "1"


def _add_gat(m, hfeat=512, embfeat=768, numheads=4):
    m.gat = GATConv(in_feats=embfeat, out_feats=hfeat, num_heads=numheads, feat_drop=m.p_gat)
    m.gat2 = GATConv(in_feats=hfeat * numheads, out_feats=hfeat, num_heads=numheads, feat_drop=m.p_gat)
    m.fc = nn.Linear(hfeat * numheads, hfeat)


def _add_gcn(m):
    for i in range(1, 9):
        setattr(m, f"Rs_GCN_{i}", Rs_GCN(in_channels=512, inter_channels=512))


class Multi_DefectModel_GATPOS(nn.Module):
    """Positions joined to the node embeddings BEFORE the GAT (reference GraphModel.py:721-827): ELU(fc_gat 768->720) ++ ELU(fc_bbox 4->48)
    -> GATConv x2 -> fc -> 8 hidden -> pad to 100 -> bn_gat -> hfc -> ELU -> mean over the 100 slots."""

    def __init__(self, config, pretrained=True, attention=True, act_dtype=torch.bfloat16):
        super().__init__()
        _head_common(self, config, act_dtype, 0.1)
        _add_gat(self)
        self.fconly = nn.Linear(768, 512)
        self.hidden = nn.ModuleList([nn.Linear(512, 512) for _ in range(8)])
        self.bn_text = nn.BatchNorm1d(768)
        self.fc_text = nn.Linear(768, 512)
        self.swinbn = nn.BatchNorm1d(1024)
        self.swinfc = nn.Linear(1024, 512)
        self.hbn = nn.BatchNorm1d(512)
        self.hfc = nn.Linear(512, 512)
        self.bn_gat = nn.BatchNorm1d(self.max_node)
        self.fc_gat = nn.Linear(768, 720)
        self.bn_bbox = nn.BatchNorm1d(self.max_node)
        self.fc_bbox = nn.Linear(4, 48)
        self.final_fc = nn.Linear(512 * 3, self.num_classes)
        self.final_fc_bn = nn.BatchNorm1d(512 * 3)
        self.unused_parameter_prefixes = ("fconly.", "hbn.", "bn_bbox.")

    def forward(self, g, img_embedding, func_text_embedding):
        ad, tr, B = self.act_dtype, self.training, g.batch_size
        ops.USE_SPLIT3[0] = False
        h = g.ndata["_UNIX_NODE_EMB"]
        hip.require_gpu(h, img_embedding, func_text_embedding)
        x = _feature_branch(img_embedding, self.swinbn, self.swinfc, ad)
        t = _feature_branch(func_text_embedding, self.bn_text, self.fc_text, ad)
        h = linear_act(_cast_in(h, ad), self.fc_gat.weight, self.fc_gat.bias, "elu")                      # [N,720]
        pos = linear_act(_cast_in(g.ndata["pos_emb"], ad), self.fc_bbox.weight, self.fc_bbox.bias, "elu")  # [N,48]
        h = _ConcatColsFn.apply(h, pos)
        h = _hidden_stack(self.hidden, _gat_node_features(self, g, h, tr), self.p_hidden, tr)
        g.ndata['HGATOUTPUT'] = h
        g.ndata['HFGATOUTPUT'] = pos
        h_i = batch_norm(_SegmentPadFn.apply(h, g.index()["node_offsets"], B, self.max_node), self.bn_gat)
        h_i = linear_act(h_i, self.hfc.weight, self.hfc.bias, "elu").view(B * self.max_node, 512)
        slots = (torch.arange(B + 1, dtype=torch.int32) * self.max_node).to(h_i.device)
        hf = cast_to(_MeanNodesFn.apply(h_i, slots, B), torch.float32)
        return _bn_classifier(self, _ConcatColsFn.apply(cast_to(x, torch.float32), hf, cast_to(t, torch.float32)))


class _MlpGcnHead(nn.Module):
    """The reference's NOGAT2 / NOGAT3 / NOGAT4 heads (GraphModel.py:1053-1382): node MLP instead of the GAT in front of the Rs_GCN chain.
      NOGAT2: fconly(768->512) -> 8 hidden; raw positions padded beside them            (the full head with the GAT swapped for fconly)
      NOGAT3: as NOGAT2, positions through fc_bbox(4->128) + 8 pos_hidden(128) per node, fc_bbox2(128->32) after the padding
      NOGAT4: fconly(768->480) ++ ELU(fc_bbox 4->32) per node -> 8 hidden(512); after padding bn_gat -> fc_gat(512->512) -> ELU only"""
    VARIANT = 2

    def __init__(self, config, pretrained=True, attention=True, act_dtype=torch.bfloat16):
        super().__init__()
        _head_common(self, config, act_dtype, 0.2)
        v = self.VARIANT
        self.fconly = nn.Linear(768, 480 if v == 4 else 512)
        self.hidden = nn.ModuleList([nn.Linear(512, 512) for _ in range(8)])
        if v == 3:
            self.pos_hidden = nn.ModuleList([nn.Linear(128, 128) for _ in range(8)])
        _add_gcn(self)
        self.bn_text = nn.BatchNorm1d(768)
        self.ln_text = nn.LayerNorm(768)
        self.fc_text = nn.Linear(768, 512)
        self.bn_gat = nn.BatchNorm1d(self.max_node)
        self.fc_gat = nn.Linear(512, 512 if v == 4 else 480)
        if v != 4:
            self.bn_bbox = nn.BatchNorm1d(self.max_node)
        self.fc_bbox = nn.Linear(4, 128 if v == 3 else 32)
        if v == 3:
            self.fc_bbox2 = nn.Linear(128, 32)
        self.swinbn = nn.BatchNorm1d(1024)
        self.swinfc = nn.Linear(1024, 512)
        self.hbn = nn.BatchNorm1d(512)
        self.hln = nn.LayerNorm(512)
        self.hfc = nn.Linear(512, 512)
        self.final_fc = nn.Linear(512 * 3, self.num_classes)
        self.final_fc_bn = nn.BatchNorm1d(512 * 3)
        self.unused_parameter_prefixes = ("ln_text.", "hbn.", "hln.", "hfc.")

    def forward(self, g, img_embedding, func_text_embedding):
        ad, tr, B, v = self.act_dtype, self.training, g.batch_size, self.VARIANT
        ops.USE_SPLIT3[0] = ad == torch.bfloat16
        h = g.ndata["_UNIX_NODE_EMB"]
        hip.require_gpu(h, img_embedding, func_text_embedding)
        x = _feature_branch(img_embedding, self.swinbn, self.swinfc, ad)
        t = _feature_branch(func_text_embedding, self.bn_text, self.fc_text, ad)
        h = linear_act(_cast_in(h, ad), self.fconly.weight, self.fconly.bias, "elu", None, self.p_mlp, tr)
        pos = _cast_in(g.ndata["pos_emb"], ad)
        hfg = g.ndata["pos_emb"]
        if v == 4:
            h = _ConcatColsFn.apply(h, linear_act(pos, self.fc_bbox.weight, self.fc_bbox.bias, "elu"))
        h = _hidden_stack(self.hidden, h, self.p_hidden, tr)
        if v == 3:
            pos = linear_act(pos, self.fc_bbox.weight, self.fc_bbox.bias, "elu")
            pos = hfg = _hidden_stack(self.pos_hidden, pos, self.p_hidden, tr)
        g.ndata['HGATOUTPUT'] = h
        g.ndata['HFGATOUTPUT'] = hfg
        if v == 4:
            h_i = batch_norm(_SegmentPadFn.apply(h, g.index()["node_offsets"], B, self.max_node), self.bn_gat)
            rows = linear_act(h_i, self.fc_gat.weight, self.fc_gat.bias, "elu").view(B * self.max_node, 512)
        else:
            rows = _padded_gcn_input(self, g, h, pos, B)
        hf = _gcn_readout(self, rows, B)
        return _bn_classifier(self, _ConcatColsFn.apply(cast_to(x, torch.float32), hf, cast_to(t, torch.float32)))


class Multi_DefectModel_NOGAT2(_MlpGcnHead):
    """(reference GraphModel.py:1277-1382)"""
    VARIANT = 2


class Multi_DefectModel_NOGAT3(_MlpGcnHead):
    """(reference GraphModel.py:1053-1170)"""
    VARIANT = 3


class Multi_DefectModel_NOGAT4(_MlpGcnHead):
    """(reference GraphModel.py:1173-1274)"""
    VARIANT = 4


def head_class(name):
    """The head class of that name from GraphModel / new_model / MotivationModel (the reference picks one by editing
    main_bigvul.py:124-129; here it is the FUSED.HEAD / --opts key)."""
    import importlib
    for mod in ("GraphModel", "new_model", "MotivationModel"):
        m = importlib.import_module("mvuld_amd.models." + mod)
        if hasattr(m, name) and name.startswith("Multi_DefectModel"):
            return getattr(m, name)
    raise KeyError(f"unknown head {name!r}")
