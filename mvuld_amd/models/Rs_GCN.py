"""Rs_GCN non-local block on libmvuld_hip.so (reference: mvuld/models/Rs_GCN.py:7-73).

Same parameters (``g``, ``theta``, ``phi`` Conv1d(k=1); ``W = Sequential(Conv1d, BatchNorm1d)``, BN scale/shift
zero-initialised, :27-34).  The head keeps node features as rows ``[B*N, D]`` so the three 1x1 convs are plain
GEMMs and the two permutes of the reference (:55-56,:66) disappear; ``forward`` accepts the reference's
``[B, D, N]`` layout and returns ``(v_star, R_div_C)`` for API compatibility.
"""
import torch
import torch.nn as nn

from .. import hip, ops
from ..hip import call, ptr, dt


def _w2(conv):
    return conv.weight


def _cat3(mod, ad):
    """theta / phi / g as ONE [3 Di, D] operand: (W, b) pseudo-parameters over the ParamStore's flat buffers when the three convolutions'
    weights (and biases) sit back to back there -- which they do since `Rs_GCN.__init__` registers them consecutively -- else None.
    The three 1x1 convolutions read the same rows, so one product v . [W_theta; W_phi; W_g]^T gives [theta | phi | g] as column blocks
    (leading dimension 3 Di); backward forms the three weight gradients, the three bias gradients and the three input-gradient
    products as one launch each.  Per Rs_GCN block and step: 2 + 2 + 4 launches fewer (8 blocks in the head)."""
    c = mod.__dict__.get("_cat3")
    key = (ad, mod.theta.weight.data.data_ptr())          # (a re-flattened / moved model gets new views)
    if c is not None and c[0] == key:
        return c[1]
    out = None
    ws = (mod.theta.weight, mod.phi.weight, mod.g.weight)
    bs = (mod.theta.bias, mod.phi.bias, mod.g.bias)
    st = getattr(ws[0], "_mv_store", None)
    if st is not None and all(getattr(p, "_mv_store", None) is st for p in ws + bs) and ad == torch.float32:
        n, nb = ws[0].numel(), bs[0].numel()
        ow = [(p.data.data_ptr() - st.flat.data_ptr()) // 4 for p in ws]
        ob = [(p.data.data_ptr() - st.flat.data_ptr()) // 4 for p in bs]
        if ow[1] == ow[0] + n and ow[2] == ow[1] + n and ob[1] == ob[0] + nb and ob[2] == ob[1] + nb:
            Di, D = mod.inter_channels, mod.in_channels
            W = torch.nn.Parameter(st.flat[ow[0]:ow[0] + 3 * n].view(3 * Di, D), requires_grad=False)
            W.grad = st.grad[ow[0]:ow[0] + 3 * n].view(3 * Di, D)
            b = torch.nn.Parameter(st.flat[ob[0]:ob[0] + 3 * nb], requires_grad=False)
            b.grad = st.grad[ob[0]:ob[0] + 3 * nb]
            W._mv_store = b._mv_store = st
            out = (W, b)
    mod.__dict__["_cat3"] = (key, out)
    return out


CAT3 = [__import__("os").environ.get("MVULD_RSGCN_CAT3", "1") != "0"]      # theta / phi / g as one product (A/B, tests)


class _RsGCNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, mod, B):
        """v: [B*N, D] node rows."""
        D = mod.in_channels
        Di = mod.inter_channels
        N = v.shape[0] // B
        ad = v.dtype
        cat = _cat3(mod, ad) if (CAT3[0] and ops.SPLIT3_TRANS[0] and ops.f32x3_ok(ad, N, Di)) else None
        if cat is not None:
            tpg = ops.gemm_nt(v, cat[0].data, bias=cat[1].data)                     # [B*N, 3 Di] = [theta | phi | g]
            th, ph, gv = tpg[:, :Di], tpg[:, Di:2 * Di], tpg[:, 2 * Di:]
        else:
            tpg = None
            th = ops.gemm_nt(v, ops.weight(mod.theta.weight, ad).view(Di, D), bias=mod.theta.bias.data)
            ph = ops.gemm_nt(v, ops.weight(mod.phi.weight, ad).view(Di, D), bias=mod.phi.bias.data)
            gv = ops.gemm_nt(v, ops.weight(mod.g.weight, ad).view(Di, D), bias=mod.g.bias.data)
        ld = th.stride(0)                                                           # Di, or 3 Di for the column blocks of one product
        # R = theta^T phi / N   per graph: [N, N]
        R = ops.gemm_nt(th, ph, M=N, N=N, K=Di, lda=ld, ldb=ld, batch=B, sa=N * ld, sb=N * ld, alpha=1.0 / N)
        y = torch.empty((B * N, Di), dtype=ad, device=v.device)
        if ops.SPLIT3_TRANS[0] and ops.f32x3_ok(ad, N, Di):
            # y = R gv with gv read as it is stored ([N, Di] = the product's [K, N] operand): no transpose pass
            ops.gemm_nt(R, gv, out=y, M=N, N=Di, K=N, lda=N, ldb=ld, ldc=Di, batch=B, sa=N * N, sb=N * ld, sc=N * Di, tb=True)
        else:
            gvT = ops.transpose(gv, R=N, C=Di, batch=B)                   # [B, Di, N]
            ops.gemm_nt(R, gvT, out=y, M=N, N=Di, K=N, lda=N, ldb=N, ldc=Di, batch=B, sa=N * N, sb=Di * N, sc=N * Di)
        conv, bn = mod.W[0], mod.W[1]
        wy = ops.gemm_nt(y, ops.weight(conv.weight, ad).view(D, Di), bias=conv.bias.data)
        training = mod.training
        wyn, sm, sr = ops.batchnorm_fwd(wy, bn.weight.data, bn.bias.data, bn.running_mean, bn.running_var, B * N, D, 1, D, 1, 1,
                                        training, bn.eps, bn.momentum)
        if training:
            bn.num_batches_tracked += 1
        out = ops.add(wyn, v)
        ctx.save_for_backward(v, th, ph, gv, R, y, wy, sm, sr)
        ctx.mod, ctx.B, ctx.training, ctx.cat = mod, B, training, cat
        ctx.mark_non_differentiable(R)
        return out, R

    @staticmethod
    def backward(ctx, dout, _dR):
        v, th, ph, gv, R, y, wy, sm, sr = ctx.saved_tensors
        mod, B, training = ctx.mod, ctx.B, ctx.training
        D, Di = mod.in_channels, mod.inter_channels
        N = v.shape[0] // B
        ad = v.dtype
        conv, bn = mod.W[0], mod.W[1]
        dout = dout.contiguous()
        # the conv bias gradient (column sums of dwy, ~0 under batch statistics) comes from the BatchNorm kernel in fp32
        dwy = ops.batchnorm_bwd(dout, wy, bn.weight, bn.bias, sm, sr, B * N, D, 1, D, 1, 1, training, dxsum=ops.grad_of(conv.bias))
        ops.linear_wgrad(dwy, y, conv.weight, None)
        dy = ops.gemm_nt(dwy, ops.weight_t(conv.weight, ad))                                         # [B*N, Di]
        # y = R gv
        cat = ctx.cat
        ld = th.stride(0)
        dR = ops.gemm_nt(dy, gv, M=N, N=N, K=Di, lda=Di, ldb=ld, batch=B, sa=N * Di, sb=N * ld)     # [B,N,N]
        if cat is not None:
            dcat = torch.empty((B * N, 3 * Di), dtype=ad, device=v.device)          # [dtheta | dphi | dg], written in place by the three products
            dth, dph, dgv = dcat[:, :Di], dcat[:, Di:2 * Di], dcat[:, 2 * Di:]
        else:
            dgv = torch.empty_like(gv)
            dth = torch.empty_like(th)
            dph = torch.empty_like(ph)
        if ops.SPLIT3_TRANS[0] and ops.f32x3_ok(ad, N, Di):
            # dgv = R^T dy, dth = dR ph / N, dph = dR^T th / N: the [K, M] / [K, N] operands are read as they are stored (five transposes less)
            bs = dict(M=N, N=Di, K=N, ldc=ld, batch=B, sc=N * ld)
            ops.gemm_nt(R, dy, out=dgv, lda=N, ldb=Di, sa=N * N, sb=N * Di, ta=True, tb=True, **bs)
            ops.gemm_nt(dR, ph, out=dth, lda=N, ldb=ld, sa=N * N, sb=N * ld, tb=True, alpha=1.0 / N, **bs)
            ops.gemm_nt(dR, th, out=dph, lda=N, ldb=ld, sa=N * N, sb=N * ld, ta=True, tb=True, alpha=1.0 / N, **bs)
        else:
            RT = ops.transpose(R, R=N, C=N, batch=B)
            dyT = ops.transpose(dy, R=N, C=Di, batch=B)                                              # [B,Di,N]
            ops.gemm_nt(RT, dyT, out=dgv, M=N, N=Di, K=N, lda=N, ldb=N, ldc=Di, batch=B, sa=N * N, sb=Di * N, sc=N * Di)
            # R = th ph^T / N
            phT = ops.transpose(ph, R=N, C=Di, batch=B)
            thT = ops.transpose(th, R=N, C=Di, batch=B)
            dRT = ops.transpose(dR, R=N, C=N, batch=B)
            ops.gemm_nt(dR, phT, out=dth, M=N, N=Di, K=N, lda=N, ldb=N, ldc=Di, batch=B, sa=N * N, sb=Di * N, sc=N * Di, alpha=1.0 / N)
            ops.gemm_nt(dRT, thT, out=dph, M=N, N=Di, K=N, lda=N, ldb=N, ldc=Di, batch=B, sa=N * N, sb=Di * N, sc=N * Di, alpha=1.0 / N)
        if cat is not None:
            # one weight-gradient product, one column sum, one input-gradient product for the three convolutions (W_cat read as it is
            # stored: [3 Di, D] = the product's [K, N] operand)
            ops.linear_wgrad(dcat, v, cat[0], cat[1])
            dv = torch.empty_like(v)
            ops.gemm_nt(dcat, cat[0].data, out=dv, M=B * N, N=D, K=3 * Di, lda=3 * Di, ldb=D, ldc=D, tb=True, epi=hip.EPI_ADD_AUX, aux=dout)
            return dv, None, None
        # the transposed input is only an operand of the fp32 (parity-mode) weight-gradient route
        vT = None if (ops.USE_SPLIT3[0] and ops.USE_TN_WGRAD[0] and not ops.FORCE_SIMPLE_GEMM[0]) else ops.transpose(v)
        ops.linear_wgrad(dth, v, mod.theta.weight, mod.theta.bias, xT=vT)
        ops.linear_wgrad(dph, v, mod.phi.weight, mod.phi.bias, xT=vT)
        ops.linear_wgrad(dgv, v, mod.g.weight, mod.g.bias, xT=vT)
        dv = ops.gemm_nt(dth, ops.weight_t(mod.theta.weight, ad), epi=hip.EPI_ADD_AUX, aux=dout)
        dv = ops.gemm_nt(dph, ops.weight_t(mod.phi.weight, ad), epi=hip.EPI_ADD_AUX, aux=dv)
        dv = ops.gemm_nt(dgv, ops.weight_t(mod.g.weight, ad), epi=hip.EPI_ADD_AUX, aux=dv)
        return dv, None, None


class Rs_GCN(nn.Module):
    def __init__(self, in_channels, inter_channels, bn_layer=True):
        super().__init__()
        self.in_channels = in_channels
        self.inter_channels = inter_channels
        if self.inter_channels is None:
            self.inter_channels = max(in_channels // 2, 1)
        assert bn_layer, "the hot path uses bn_layer=True"
        # theta, phi, g registered back to back (the reference registers g, W, theta, phi: same names, another order): a ParamStore then
        # holds the three weights -- and the three biases -- contiguously, and the block runs them as one product (_cat3)
        self.theta = nn.Conv1d(self.in_channels, self.inter_channels, kernel_size=1)
        self.phi = nn.Conv1d(self.in_channels, self.inter_channels, kernel_size=1)
        self.g = nn.Conv1d(self.in_channels, self.inter_channels, kernel_size=1)
        self.W = nn.Sequential(nn.Conv1d(self.inter_channels, self.in_channels, kernel_size=1), nn.BatchNorm1d(self.in_channels))
        nn.init.constant_(self.W[1].weight, 0)
        nn.init.constant_(self.W[1].bias, 0)

    def forward_rows(self, v_rows, B):
        """[B*N, D] -> [B*N, D] (fast path used by the head)."""
        out, R = _RsGCNFn.apply(v_rows, self, B)
        return out, R

    def forward(self, v):
        """Reference layout: v [B, D, N] -> (v_star [B, D, N], R_div_C [B, N, N])."""
        hip.require_gpu(v)
        B, D, N = v.shape
        rows = ops.transpose(v.contiguous(), R=D, C=N, batch=B).view(B * N, D)
        out, R = self.forward_rows(rows, B)
        return ops.transpose(out.view(B, N, D), R=N, C=D, batch=B), R
