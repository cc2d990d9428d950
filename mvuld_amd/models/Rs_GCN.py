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


class _RsGCNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, mod, B):
        """v: [B*N, D] node rows."""
        D = mod.in_channels
        Di = mod.inter_channels
        N = v.shape[0] // B
        ad = v.dtype
        th = ops.gemm_nt(v, ops.weight(mod.theta.weight, ad).view(Di, D), bias=mod.theta.bias.data)
        ph = ops.gemm_nt(v, ops.weight(mod.phi.weight, ad).view(Di, D), bias=mod.phi.bias.data)
        gv = ops.gemm_nt(v, ops.weight(mod.g.weight, ad).view(Di, D), bias=mod.g.bias.data)
        # R = theta^T phi / N   per graph: [N, N]
        R = ops.gemm_nt(th, ph, M=N, N=N, K=Di, lda=Di, ldb=Di, batch=B, sa=N * Di, sb=N * Di, alpha=1.0 / N)
        y = torch.empty((B * N, Di), dtype=ad, device=v.device)
        if ops.SPLIT3_TRANS[0] and ops.f32x3_ok(ad, N, Di):
            # y = R gv with gv read as it is stored ([N, Di] = the product's [K, N] operand): no transpose pass
            ops.gemm_nt(R, gv, out=y, M=N, N=Di, K=N, lda=N, ldb=Di, ldc=Di, batch=B, sa=N * N, sb=N * Di, sc=N * Di, tb=True)
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
        ctx.mod, ctx.B, ctx.training = mod, B, training
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
        dR = ops.gemm_nt(dy, gv, M=N, N=N, K=Di, lda=Di, ldb=Di, batch=B, sa=N * Di, sb=N * Di)     # [B,N,N]
        dgv = torch.empty_like(gv)
        dth = torch.empty_like(th)
        dph = torch.empty_like(ph)
        if ops.SPLIT3_TRANS[0] and ops.f32x3_ok(ad, N, Di):
            # dgv = R^T dy, dth = dR ph / N, dph = dR^T th / N: the [K, M] / [K, N] operands are read as they are stored (five transposes less)
            bs = dict(M=N, N=Di, K=N, ldc=Di, batch=B, sc=N * Di)
            ops.gemm_nt(R, dy, out=dgv, lda=N, ldb=Di, sa=N * N, sb=N * Di, ta=True, tb=True, **bs)
            ops.gemm_nt(dR, ph, out=dth, lda=N, ldb=Di, sa=N * N, sb=N * Di, tb=True, alpha=1.0 / N, **bs)
            ops.gemm_nt(dR, th, out=dph, lda=N, ldb=Di, sa=N * N, sb=N * Di, ta=True, tb=True, alpha=1.0 / N, **bs)
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
        self.g = nn.Conv1d(self.in_channels, self.inter_channels, kernel_size=1)
        self.W = nn.Sequential(nn.Conv1d(self.inter_channels, self.in_channels, kernel_size=1), nn.BatchNorm1d(self.in_channels))
        nn.init.constant_(self.W[1].weight, 0)
        nn.init.constant_(self.W[1].bias, 0)
        self.theta = nn.Conv1d(self.in_channels, self.inter_channels, kernel_size=1)
        self.phi = nn.Conv1d(self.in_channels, self.inter_channels, kernel_size=1)

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
